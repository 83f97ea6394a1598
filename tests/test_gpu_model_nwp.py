"""GPU parity of the NWP-only Conv3D model (mirror of predict_pv_yield/models/conv3d/model_nwp.py) against the golden
vectors produced by the reference's own module source (tests/golden/make_conv3d_nwp_golden.py) and against the
bf16-emulating torch-CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import conv3d_oracle as co
from tests.test_oracle_conv import GOLD_NWP, NWP, NWP_1CH, nwp_case

pytestmark = pytest.mark.gpu

UNUSED = ("pv_system_id_embedding", "pv_fc1")       # registered, never read by forward (model_nwp.py:112-153)


def _hip_model(oracle, kw, precision, device):
    from predict_pv_yield_amd.models.conv3d.model_nwp import Model
    model = Model(**kw, precision=precision)
    assert list(model.state_dict().keys()) == list(oracle.state_dict().keys())
    model.load_state_dict(oracle.state_dict())
    return model.to(device)


def _batch(t, device):
    return {"pv": {"pv_yield": t["pv"].to(device)}, "gsp": {"gsp_yield": t["gsp"].to(device)},
            "nwp": {"data": t["nwp"].to(device)}}


@pytest.mark.parametrize("tag,kw", [("nwp", NWP), ("nwp_1ch", NWP_1CH)])
def test_fp32_against_reference_golden(tag, kw, device):
    """HIP fp32 path vs the vectors the REFERENCE module produced: forward, first-step gradients, 3 Adam steps, and the
    parameters forward never reads stay bit-identical to their initial values."""
    g = np.load(GOLD_NWP)
    oracle, t = nwp_case(g, tag, kw)
    model = _hip_model(oracle, kw, "fp32", device)
    assert model.name == "conv3d_sat_nwp" and model.nwp_cnn_output_size == g[f"{tag}/attrs"][0]
    batch = _batch(t, device)
    np.testing.assert_allclose(model(batch).detach().cpu().numpy(), g[f"{tag}/y_hat"], rtol=1e-4, atol=1e-5)
    opt = model.configure_optimizers()
    losses = []
    for step in range(3):
        opt.zero_grad()
        loss = model.training_step(batch, 0)
        loss.backward()
        if step == 0:
            for k, p in model.named_parameters():
                if k.split(".")[0] in UNUSED:
                    assert p.grad is None or not p.grad.any(), k
                    continue
                ref = g[f"{tag}/grad/{k}"]                      # [sum, abs-sum, 64 samples]
                f = p.grad.detach().cpu().flatten()
                idx = torch.linspace(0, f.numel() - 1, min(64, f.numel())).long()
                scale = np.abs(ref[2:]).max() + 1e-12
                assert np.abs(f[idx].double().numpy() - ref[2:]).max() <= 1e-3 * scale + 1e-7, k
                assert abs(f.double().abs().sum().item() - ref[1]) <= 1e-3 * ref[1] + 1e-6, k
        opt.step()
        losses.append(float(loss.detach()))
    np.testing.assert_allclose(losses, g[f"{tag}/losses"], rtol=1e-4)
    for k, p in model.named_parameters():
        if k.split(".")[0] in UNUSED:
            assert torch.equal(p.detach().cpu(), torch.from_numpy(g[f"{tag}/init/{k}"])), k


def test_bf16_against_emulating_oracle(device):
    g = np.load(GOLD_NWP)
    oracle, t = nwp_case(g, "nwp", NWP, emulate_bf16=True)
    model = _hip_model(oracle, NWP, "bf16", device)
    batch = _batch(t, device)
    y_ref = oracle(t["nwp"])
    _, nmae, _, _ = co.forecast_losses(y_ref, co.select_target(t["gsp"], oracle.forecast_len))
    nmae.backward()
    loss = model.training_step(batch, 0)
    loss.backward()
    assert abs(float(loss) - float(nmae)) <= 2e-3 * abs(float(nmae))
    for (k, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
        if q.grad is None:
            continue
        rel = (p.grad.cpu() - q.grad).norm().item() / (q.grad.norm().item() + 1e-12)
        assert rel <= 3e-2, (k, rel)
    np.testing.assert_allclose(model(batch).detach().cpu().numpy(), g["nwp/y_hat"], rtol=3e-2, atol=3e-3)


def test_shipped_yaml_trains_on_fake_data(device):
    """configs/model/conv3d_nwp.yaml (6 layers, one NWP variable, 64 px) through the Trainer on fake batches."""
    import os
    from predict_pv_yield_amd import lightning as pl
    from predict_pv_yield_amd.data.fake import FakeDataConfiguration, FakeDataset
    from predict_pv_yield_amd.models.conv3d.model_nwp import Model
    from predict_pv_yield_amd.utils import load_config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    model = Model(**load_config(os.path.join(root, "configs", "model", "conv3d_nwp.yaml")))
    cfg = FakeDataConfiguration(batch_size=4, history_minutes=30, forecast_minutes=120, satellite_image_size_pixels=16,
                                nwp_image_size_pixels=64, number_nwp_channels=1)
    loader = torch.utils.data.DataLoader(FakeDataset(cfg, length=2), batch_size=None)
    trainer = pl.Trainer(gpus=1, max_epochs=1)
    trainer.fit(model, loader)
    y = trainer.predict(model, loader)
    assert y[0].shape == (4, model.forecast_len_30)
