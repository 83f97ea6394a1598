"""NetCDFDataModule — same constructor surface as predict_pv_yield/data/dataloader.py:38-131
(temp_path, n_train_data, n_val_data, cloud, num_workers, pin_memory, data_path, fake_data [, shuffle_train]).
`fake_data=True` generates seeded synthetic batches; otherwise whole-batch files (`000000.npz` / `.nc`, see
data/netcdf_dataset.py) are read from `<data_path>/train` and `<data_path>/test` like the reference does
(data/dataloader.py:93-131); cloud download and NetCDF-4 decoding are outside the hot path.
Under data-parallel training each rank takes every world-th batch index, padded by wrapping to equal counts
(DistributedSampler as Lightning's replace_sampler_ddp installs it, configs/trainer/all_params.yaml:43): samples are
independent, no collective."""
import logging
import os
from typing import Optional

import torch
import yaml

from ..distributed import shard_indices
from ..lightning import LightningDataModule
from .fake import FakeDataConfiguration, FakeDataset

_LOG = logging.getLogger(__name__)


def _identity_collate(x):
    return x


class _Shard(torch.utils.data.Dataset):
    """The items of `base` at `indices` (this rank's share: distributed.shard_indices), shifted by `offset`."""

    def __init__(self, base, indices, offset=0):
        self.base, self.indices, self.offset = base, list(indices), offset

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, i):
        if i >= len(self):
            raise IndexError(i)
        return self.base[self.offset + self.indices[i]]


class NetCDFDataModule(LightningDataModule):
    def __init__(self, temp_path: str = ".", n_train_data: int = 24900, n_val_data: int = 1000, cloud: str = "aws",
                 num_workers: int = 8, pin_memory: bool = True, data_path: str = "prepared_ML_training_data/v4/",
                 fake_data: bool = False, shuffle_train: bool = True, batch_size: Optional[int] = None,
                 configuration: Optional[FakeDataConfiguration] = None):
        super().__init__()
        self.temp_path, self.cloud, self.data_path = temp_path, cloud, data_path
        self.n_train_data, self.n_val_data = n_train_data, n_val_data
        self.num_workers, self.pin_memory = num_workers, pin_memory
        self.fake_data, self.shuffle_train = fake_data, shuffle_train
        cfg = configuration or self._configuration_from_path(data_path)
        if batch_size is not None:
            cfg.batch_size = batch_size
        self.configuration = cfg

    @staticmethod
    def _configuration_from_path(data_path: str) -> FakeDataConfiguration:
        """Reads `<data_path>/configuration.yaml` (nowcasting_dataset layout, tests/configs/dataset/configuration.yaml)
        if present; only the fields the fake data needs are used."""
        cfg = FakeDataConfiguration()
        f = os.path.join(data_path, "configuration.yaml")
        if os.path.exists(f):
            with open(f) as fh:
                raw = yaml.safe_load(fh) or {}
            inp = raw.get("input_data", {})
            cfg.history_minutes = inp.get("default_history_minutes", cfg.history_minutes)
            cfg.forecast_minutes = inp.get("default_forecast_minutes", cfg.forecast_minutes)
            sat = inp.get("satellite", {})
            cfg.satellite_image_size_pixels = sat.get("satellite_image_size_pixels", cfg.satellite_image_size_pixels)
            if "satellite_channels" in sat:
                cfg.number_sat_channels = len(sat["satellite_channels"])
            nwp = inp.get("nwp", {})
            cfg.nwp_image_size_pixels = nwp.get("nwp_image_size_pixels", cfg.nwp_image_size_pixels)
            if "nwp_channels" in nwp:
                cfg.number_nwp_channels = len(nwp["nwp_channels"])
            cfg.batch_size = raw.get("process", {}).get("batch_size", cfg.batch_size)
            cfg.seed = raw.get("process", {}).get("seed", cfg.seed)
        else:
            _LOG.warning(f"no configuration.yaml under data_path={data_path!r}: using the default shapes "
                         f"({cfg.history_minutes}/{cfg.forecast_minutes} minutes, {cfg.satellite_image_size_pixels} px, "
                         f"batch {cfg.batch_size})")
        return cfg

    def _loader(self, n_batches: int, offset: int, split: str = "train"):
        mine = shard_indices(n_batches)
        if self.fake_data:
            base = FakeDataset(self.configuration, length=offset + n_batches)
            ds = _Shard(base, mine, offset)
        else:
            from .netcdf_dataset import NetCDFDataset
            base = NetCDFDataset(n_batches, os.path.join(self.data_path, split), os.path.join(self.temp_path, split),
                                 configuration=self.configuration)
            ds = _Shard(base, mine)
        # every item is a whole batch: batch_size=None (data/dataloader.py:82-91)
        return torch.utils.data.DataLoader(ds, batch_size=None, num_workers=0, pin_memory=False)

    def train_dataloader(self):
        return self._loader(self.n_train_data if not self.fake_data else min(self.n_train_data, 10), 0, "train")

    def val_dataloader(self):
        return self._loader(self.n_val_data if not self.fake_data else min(self.n_val_data, 10), 1000, "test")

    def test_dataloader(self):
        # the reference reads the "test" folder for both (data/dataloader.py:120-129)
        return self._loader(self.n_val_data if not self.fake_data else min(self.n_val_data, 10), 2000, "test")
