"""GPU parity of the sat+nwp Conv3D model (mirror of predict_pv_yield/models/conv3d/model_sat_nwp.py) against the
golden vectors produced by the reference's own module source and against the torch-CPU oracle; the tests at the
bottom mirror the reference's tests/models/conv3d/test_conv3d_model_sat_nwp.py."""
import os

import numpy as np
import pytest
import torch

from oracle import conv3d_oracle as co
from tests.test_oracle_conv import GOLD, SN, SN_PV, sat_nwp_case

pytestmark = pytest.mark.gpu


def _hip_model(oracle, kw, precision, device):
    from predict_pv_yield_amd.models.conv3d.model_sat_nwp import Model
    model = Model(**kw, precision=precision)
    assert list(model.state_dict().keys()) == list(oracle.state_dict().keys())
    model.load_state_dict(oracle.state_dict())
    return model.to(device)


def _batch(t, device):
    return {"satellite": {"data": t["sat"].to(device)},
            "pv": {"pv_yield": t["pv"].to(device), "pv_system_row_number": t["row"].to(device)},
            "gsp": {"gsp_yield": t["gsp"].to(device), "gsp_id": t["gsp_id"].to(device)},
            "nwp": {"data": t["nwp"].to(device)}}


def test_embedding_kernels(device):
    from predict_pv_yield_amd import functional as Fn
    g = torch.Generator().manual_seed(0)
    table = torch.randn(940, 16, generator=g)
    ids = torch.tensor([3, 939, 3, 0, 17, 3, 939])
    ref_t = table.clone().requires_grad_(True)
    ref = torch.nn.functional.embedding(ids, ref_t)
    dout = torch.randn(ref.shape, generator=g)
    ref.backward(dout)
    hip_t = table.to(device).requires_grad_(True)
    out = Fn.embedding(hip_t, ids.to(device))
    assert torch.equal(out.detach().cpu(), ref.detach())
    out.backward(dout.to(device))
    torch.testing.assert_close(hip_t.grad.cpu(), ref_t.grad, rtol=1e-6, atol=1e-7)
    assert torch.equal(Fn.embedding(hip_t, torch.tensor([940, -1], device=device)).detach().cpu(), torch.zeros(2, 16))


@pytest.mark.parametrize("tag,kw", [("sat_nwp", SN), ("sat_nwp_pv", SN_PV)])
def test_fp32_against_reference_golden(tag, kw, device):
    """HIP fp32 path vs the vectors the REFERENCE module produced: forward, first-step gradients, 3 Adam steps."""
    g = np.load(GOLD)
    oracle, args, yld, t = sat_nwp_case(g, tag, kw)
    model = _hip_model(oracle, kw, "fp32", device)
    batch = _batch(t, device)
    y = model(batch)
    np.testing.assert_allclose(y.detach().cpu().numpy(), g[f"{tag}/y_hat"], rtol=1e-4, atol=1e-5)
    opt = model.configure_optimizers()
    losses = []
    for step in range(3):
        opt.zero_grad()
        loss = model.training_step(batch, 0)
        loss.backward()
        if step == 0:
            for k, p in model.named_parameters():
                ref = g[f"{tag}/grad/{k}"]                      # [sum, abs-sum, 64 samples]
                f = p.grad.detach().cpu().flatten()
                idx = torch.linspace(0, f.numel() - 1, min(64, f.numel())).long()
                scale = np.abs(ref[2:]).max() + 1e-12
                assert np.abs(f[idx].double().numpy() - ref[2:]).max() <= 1e-3 * scale + 1e-7, k
                assert abs(f.double().abs().sum().item() - ref[1]) <= 1e-3 * ref[1] + 1e-6, k
        opt.step()
        losses.append(float(loss))
    np.testing.assert_allclose(losses, g[f"{tag}/losses"], rtol=1e-4)


def test_bf16_against_emulating_oracle(device):
    g = np.load(GOLD)
    oracle, args, yld, t = sat_nwp_case(g, "sat_nwp", SN, emulate_bf16=True)
    model = _hip_model(oracle, SN, "bf16", device)
    batch = _batch(t, device)
    y_ref = oracle(*args)
    _, nmae, _, _ = co.forecast_losses(y_ref, co.select_target(yld, oracle.forecast_len))
    nmae.backward()
    loss = model.training_step(batch, 0)
    loss.backward()
    assert abs(float(loss) - float(nmae)) <= 2e-3 * abs(float(nmae))
    for (k, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
        rel = (p.grad.cpu() - q.grad).norm().item() / (q.grad.norm().item() + 1e-12)
        assert rel <= 3e-2, (k, rel)
    # and the plain f32 reference output within bf16 noise
    np.testing.assert_allclose(model(batch).detach().cpu().numpy(), g["sat_nwp/y_hat"], rtol=3e-2, atol=3e-3)


# ---- mirrors of the reference's tests/models/conv3d/test_conv3d_model_sat_nwp.py -------------------------------
def _config():
    from predict_pv_yield_amd.utils import load_config
    return load_config(os.path.join(os.path.dirname(__file__), "configs", "model", "conv3d_sat_nwp.yaml"))


def _fake_loader(nwp_px=16):
    # reference fixture `configuration_conv3d` (tests/conftest.py) with nwp_image_size_pixels = 16: batches of 2,
    # 16 px satellite, 60/60 minutes
    from predict_pv_yield_amd.data.fake import FakeDataConfiguration, FakeDataset
    cfg = FakeDataConfiguration(batch_size=2, history_minutes=60, forecast_minutes=60, satellite_image_size_pixels=16,
                                nwp_image_size_pixels=nwp_px)
    return torch.utils.data.DataLoader(FakeDataset(cfg, length=2), batch_size=None)


def test_init():
    from predict_pv_yield_amd.models.conv3d.model_sat_nwp import Model
    _ = Model(**_config())


@pytest.mark.parametrize("future", [True, False])
def test_model_forward(future, device):
    from predict_pv_yield_amd.data.batch import BatchML
    from predict_pv_yield_amd.models.conv3d.model_sat_nwp import Model
    config = _config()
    config["include_future_satellite"] = future
    model = Model(**config).to(device)
    x = next(iter(_fake_loader()))
    y = model(BatchML(**x).to(device))
    assert len(y.shape) == 2 and y.shape[0] == 2 and y.shape[1] == model.forecast_len_30


def test_train(device):
    from predict_pv_yield_amd import lightning as pl
    from predict_pv_yield_amd.models.conv3d.model_sat_nwp import Model
    model = Model(**_config())
    loader = _fake_loader()
    trainer = pl.Trainer(gpus=1, max_epochs=1)
    trainer.fit(model, loader)
    _ = trainer.predict(model, loader)
