"""Whole-batch file dataset, the device-side int16 normalisation and the prefetcher (data/netcdf_dataset.py) — the input
step in front of the hot path (reference semantics predict_pv_yield/netcdf_dataset.py:55-119)."""
import os

import numpy as np
import pytest
import torch

from predict_pv_yield_amd.data import netcdf_dataset as nd
from predict_pv_yield_amd.data.fake import FakeDataConfiguration, make_fake_batch


def _write_split(root, split, n, raw_int16=True, ext="npz", seed=0):
    cfg = FakeDataConfiguration(batch_size=2, history_minutes=10, forecast_minutes=10, satellite_image_size_pixels=8,
                                nwp_image_size_pixels=2)
    rng = np.random.default_rng(seed)
    written = []
    for i in range(n):
        sections = make_fake_batch(cfg, torch.Generator().manual_seed(seed + i), as_dict=True)
        arrays = {s: {k: v.numpy() for k, v in d.items()} for s, d in sections.items()}
        if raw_int16:
            arrays["satellite"]["data"] = rng.integers(0, 1024, arrays["satellite"]["data"].shape).astype(np.int16)
        nd.write_batch_file(os.path.join(root, split, nd.get_netcdf_filename(i, ext)), arrays)
        written.append(arrays)
    return written


@pytest.mark.parametrize("ext", ["npz", "nc"])
def test_batch_files_round_trip(tmp_path, ext):
    written = _write_split(str(tmp_path), "train", 3, ext=ext)
    ds = nd.NetCDFDataset(3, str(tmp_path / "train"), str(tmp_path / "tmp"))
    assert len(ds) == 3
    for i, ref in enumerate(written):
        got = ds[i]
        assert set(got) == set(ref)
        for s in ref:
            for k, v in ref[s].items():
                g = got[s][k].numpy()
                assert g.shape == v.shape, (s, k)
                if v.dtype == np.int64 and ext == "nc":
                    assert np.array_equal(g.astype(np.int64), v)      # classic NetCDF stores them as float64
                else:
                    assert g.dtype == v.dtype and np.array_equal(g, v), (s, k)
    assert got["satellite"]["data"].dtype == torch.int16                # raw counts stay int16 until they are on the device
    with pytest.raises(IndexError):
        ds[3]
    assert nd.get_netcdf_filename(7) == "000007.npz"


def test_datamodule_reads_prepared_batches(tmp_path):
    from predict_pv_yield_amd.data.dataloader import NetCDFDataModule
    _write_split(str(tmp_path), "train", 2)
    _write_split(str(tmp_path), "test", 1, seed=50)
    dm = NetCDFDataModule(temp_path=str(tmp_path / "tmp"), n_train_data=2, n_val_data=1, data_path=str(tmp_path),
                          fake_data=False, num_workers=0)
    train = list(dm.train_dataloader())
    assert len(train) == 2 and train[0]["satellite"]["data"].shape == (2, 11, 5, 8, 8)
    assert len(list(dm.val_dataloader())) == 1 and len(list(dm.test_dataloader())) == 1


def test_normalise_on_cpu_fails_loudly():
    batch = {"satellite": {"data": torch.zeros(1, 11, 2, 4, 4, dtype=torch.int16)}}
    with pytest.raises(RuntimeError, match="MI355X"):
        nd.normalise_satellite_on_device(batch)
    f32 = {"satellite": {"data": torch.zeros(1, 11, 2, 4, 4)}}
    assert nd.normalise_satellite_on_device(f32) is f32                # already normalised: untouched


@pytest.mark.gpu
def test_device_normalisation_and_prefetcher(tmp_path, device):
    """int16 counts cross to the device and are normalised there: bit-identical to the host formula of the reference."""
    written = _write_split(str(tmp_path), "train", 4)
    ds = nd.NetCDFDataset(4, str(tmp_path / "train"))
    loader = torch.utils.data.DataLoader(ds, batch_size=None)
    got = list(nd.DeviceBatchPrefetcher(loader, device, depth=2))
    assert len(got) == 4
    for ref, batch in zip(written, got):
        raw = ref["satellite"]["data"]
        host = (raw.astype(np.float32) - nd.SAT_MEAN[1:12, None, None, None]) / nd.SAT_STD[1:12, None, None, None]
        sat = batch.satellite.data
        assert sat.is_cuda and sat.dtype == torch.float32
        assert np.array_equal(sat.cpu().numpy(), host)
        assert torch.equal(batch.pv.pv_yield.cpu(), torch.from_numpy(ref["pv"]["pv_yield"]))
    # the Trainer's own move path does the same for loaders that are not wrapped in the prefetcher
    from predict_pv_yield_amd.lightning import _move
    moved = _move(ds[0], device)
    assert moved["satellite"]["data"].dtype == torch.float32
    assert torch.equal(moved["satellite"]["data"], got[0].satellite.data)


@pytest.mark.gpu
def test_trainer_fits_from_batch_files(tmp_path, device):
    from predict_pv_yield_amd import lightning as pl
    from predict_pv_yield_amd.data.dataloader import NetCDFDataModule
    from predict_pv_yield_amd.models.conv3d.model import Model
    _write_split(str(tmp_path), "train", 2)
    _write_split(str(tmp_path), "test", 1, seed=9)
    dm = NetCDFDataModule(temp_path=str(tmp_path / "tmp"), n_train_data=2, n_val_data=1, data_path=str(tmp_path),
                          fake_data=False, num_workers=0)
    model = Model(include_pv_yield=False, include_nwp=False, forecast_minutes=10, history_minutes=10,
                  number_of_conv3d_layers=1, conv3d_channels=32, image_size_pixels=8, number_sat_channels=11,
                  fc1_output_features=8, fc2_output_features=8, fc3_output_features=8)
    trainer = pl.Trainer(gpus=1, max_epochs=1)
    trainer.fit(model, datamodule=dm)
    assert all(torch.isfinite(p).all() for p in model.parameters())
