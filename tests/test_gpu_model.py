"""GPU parity of the whole Conv3D train step (forward + NMAE + backward + Adam) vs the torch-CPU oracle,
driven through the reference's module interface (Model(**yaml), training_step, configure_optimizers)."""
import numpy as np
import pytest
import torch

from conftest import ref_order

from oracle import conv3d_oracle as co

pytestmark = pytest.mark.gpu

# reduced config = tests/configs/model/conv3d.yaml of the reference (16 px, 60/60 min, fc 16)
SMALL = dict(include_pv_yield=False, include_nwp=False, forecast_minutes=60, history_minutes=60,
             number_of_conv3d_layers=4, conv3d_channels=32, image_size_pixels=16, number_sat_channels=11,
             fc1_output_features=16, fc2_output_features=16, fc3_output_features=16)


def _pair(kw, precision, device, seed=518):
    from predict_pv_yield_amd.models.conv3d.model import Model
    torch.manual_seed(seed)
    oracle = co.OracleConv3dModel(**kw)
    model = Model(**{k: v for k, v in kw.items() if k != "emulate_bf16"}, precision=precision)
    model.load_state_dict(oracle.state_dict())
    return oracle, model.to(device)


def _data(kw, b, seed=1):
    t = kw["history_minutes"] // 5 + kw["forecast_minutes"] // 5 + 1
    g = torch.Generator().manual_seed(seed)
    sat = torch.randn(b, kw["number_sat_channels"], t, kw["image_size_pixels"], kw["image_size_pixels"], generator=g)
    pv = torch.rand(b, t, 128, generator=g)
    return sat, pv


def _hip_steps(model, sat, pv, n, device):
    opt = model.configure_optimizers()
    batch = {"satellite": {"data": sat.to(device)}, "pv": {"pv_yield": pv.to(device)}}
    losses = []
    for _ in range(n):
        opt.zero_grad()
        loss = model.training_step(batch, 0)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    return losses


def test_state_dict_keys_match_reference_names(device):
    oracle, model = _pair(SMALL, "fp32", device)
    assert list(model.state_dict().keys()) == list(oracle.state_dict().keys())
    assert [k for k in model.state_dict()][:2] == ["sat_conv0.weight", "sat_conv0.bias"]


def test_fp32_forward_and_three_steps(device):
    oracle, model = _pair(SMALL, "fp32", device)
    sat, pv = _data(SMALL, 2)
    y_ref = oracle(sat)
    y = model({"satellite": {"data": sat.to(device)}, "pv": {"pv_yield": pv.to(device)}})
    assert y.shape == (2, model.forecast_len_5)
    torch.testing.assert_close(y.detach().cpu(), y_ref.detach(), rtol=1e-4, atol=1e-5)
    ref_losses = co.train_steps(oracle, sat, pv, 3)
    losses = _hip_steps(model, sat, pv, 3, device)
    np.testing.assert_allclose(losses, ref_losses, rtol=1e-4)
    # parameters after 3 Adam steps: Adam normalises tiny gradients to +-lr steps, so compare with an
    # absolute tolerance of a fraction of one step (lr = 5e-4)
    for (k, p), (_, q) in zip(model.state_dict().items(), oracle.state_dict().items()):
        assert (p.cpu() - q).abs().max().item() <= 2e-4, k


def test_fp32_gradients(device):
    oracle, model = _pair(SMALL, "fp32", device)
    sat, pv = _data(SMALL, 2)
    y_ref = oracle(sat)
    _, nmae, _, _ = co.forecast_losses(y_ref, co.select_target(pv, oracle.forecast_len))
    nmae.backward()
    loss = model.training_step({"satellite": {"data": sat.to(device)}, "pv": {"pv_yield": pv.to(device)}}, 0)
    loss.backward()
    for (k, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
        scale = q.grad.abs().max().item() + 1e-12
        assert (ref_order(p, p.grad).cpu() - q.grad).abs().max().item() <= 1e-3 * scale + 1e-7, k


def test_bf16_forward_and_steps(device):
    oracle, model = _pair(SMALL, "bf16", device)
    sat, pv = _data(SMALL, 2)
    y_ref = oracle(sat)
    y = model({"satellite": {"data": sat.to(device)}, "pv": {"pv_yield": pv.to(device)}})
    # bf16 operands, f32 accumulation: activations within 2e-2 relative of the f32 oracle
    torch.testing.assert_close(y.detach().cpu(), y_ref.detach(), rtol=2e-2, atol=2e-3)
    ref_losses = co.train_steps(oracle, sat, pv, 3)
    losses = _hip_steps(model, sat, pv, 3, device)
    np.testing.assert_allclose(losses, ref_losses, rtol=1e-2)


def test_bf16_gradients_vs_f32_oracle_loose(device):
    """Against the pure-f32 oracle the bf16 path carries bf16 rounding noise through four conv layers: the fc
    gradients stay within a few percent, the earliest conv weights within ~15 % (norm-wise)."""
    oracle, model = _pair(SMALL, "bf16", device)
    sat, pv = _data(SMALL, 4, seed=3)
    y_ref = oracle(sat)
    _, nmae, _, _ = co.forecast_losses(y_ref, co.select_target(pv, oracle.forecast_len))
    nmae.backward()
    loss = model.training_step({"satellite": {"data": sat.to(device)}, "pv": {"pv_yield": pv.to(device)}}, 0)
    loss.backward()
    for (k, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
        rel = (ref_order(p, p.grad).cpu() - q.grad).norm().item() / (q.grad.norm().item() + 1e-12)
        assert rel <= (0.2 if "conv" in k else 5e-2), (k, rel)


def test_bf16_gradients(device):
    """Tight check: the oracle rounds exactly the tensors the MFMA path rounds (emulate_bf16), so what is left is
    f32 accumulation order and isolated 1-ulp bf16 flips."""
    oracle, model = _pair(dict(SMALL, emulate_bf16=True), "bf16", device)
    sat, pv = _data(SMALL, 4, seed=3)
    y_ref = oracle(sat)
    _, nmae, _, _ = co.forecast_losses(y_ref, co.select_target(pv, oracle.forecast_len))
    nmae.backward()
    loss = model.training_step({"satellite": {"data": sat.to(device)}, "pv": {"pv_yield": pv.to(device)}}, 0)
    loss.backward()
    for (k, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
        ref = q.grad
        rel = (ref_order(p, p.grad).cpu() - ref).norm().item() / (ref.norm().item() + 1e-12)
        assert rel <= 3e-2, (k, rel)


def test_canonical_64px_forward_bf16(device):
    """BASELINE config 2 shape (T=18, 64 px, fc 128/128/64), B=2: forward + loss against the oracle."""
    kw = dict(include_pv_yield=False, include_nwp=False, forecast_minutes=30, history_minutes=55,
              number_of_conv3d_layers=4, conv3d_channels=32, image_size_pixels=64, number_sat_channels=11)
    oracle, model = _pair(kw, "bf16", device)
    sat, pv = _data(kw, 2)
    with torch.no_grad():
        y_ref = oracle(sat)
        y = model({"satellite": {"data": sat.to(device)}, "pv": {"pv_yield": pv.to(device)}})
    assert y.shape == (2, 6)
    torch.testing.assert_close(y.cpu(), y_ref, rtol=2e-2, atol=2e-3)


def test_include_pv_yield_and_nwp_path(device):
    """production yaml shape (configs/model/conv3d.yaml): gsp_yield history + NWP fully-connected branch."""
    from predict_pv_yield_amd.data.fake import FakeDataConfiguration, make_fake_batch
    from predict_pv_yield_amd.models.conv3d.model import Model
    kw = dict(include_pv_yield=True, include_nwp=True, forecast_minutes=120, history_minutes=30,
              number_of_conv3d_layers=6, image_size_pixels=24, number_sat_channels=11, conv3d_channels=32,
              output_variable="gsp_yield")
    torch.manual_seed(0)
    oracle = co.OracleConv3dModel(**kw)
    model = Model(**kw, precision="fp32")
    model.load_state_dict(oracle.state_dict())
    model.to(device)
    cfg = FakeDataConfiguration(batch_size=2, history_minutes=30, forecast_minutes=120, satellite_image_size_pixels=24,
                                nwp_image_size_pixels=2)
    cfg_nwp_t = 19  # number_of_nwp_features = 10*19*2*2 (model.py:60)
    batch = make_fake_batch(cfg, torch.Generator().manual_seed(4))
    batch.nwp.data = torch.randn(2, 10, cfg_nwp_t, 2, 2, generator=torch.Generator().manual_seed(5))
    y_ref = oracle(batch.satellite.data, batch.gsp.gsp_yield, batch.nwp.data)
    y = model(batch.to(device))
    assert y.shape == (2, model.forecast_len_30)
    torch.testing.assert_close(y.detach().cpu(), y_ref.detach(), rtol=1e-4, atol=1e-5)


def test_model_on_cpu_fails_loudly():
    from predict_pv_yield_amd.models.conv3d.model import Model
    model = Model(**SMALL)
    sat, pv = _data(SMALL, 2)
    with pytest.raises(RuntimeError, match="MI355X"):
        model({"satellite": {"data": sat}, "pv": {"pv_yield": pv}})


def test_fused_fc1_adam_path_in_the_model(device, monkeypatch):
    """HipAdam(fuse_large_linear): fc1's gradient is never materialised; the training trajectory is unchanged."""
    from predict_pv_yield_amd.optim import HipAdam
    sat, pv = _data(SMALL, 2)
    _, model_a = _pair(SMALL, "bf16", device)
    losses_a = _hip_steps(model_a, sat, pv, 3, device)          # fc1 is below the fusion threshold: two-pass path
    monkeypatch.setattr(HipAdam, "FUSE_MIN_NUMEL", 1)
    _, model_b = _pair(SMALL, "bf16", device)
    opt = model_b.configure_optimizers()
    assert getattr(model_b.fc1.weight, "_pv_grad_mode", "autograd") == "fused"
    batch = {"satellite": {"data": sat.to(device)}, "pv": {"pv_yield": pv.to(device)}}
    losses_b = []
    for _ in range(3):
        opt.zero_grad()
        loss = model_b.training_step(batch, 0)
        loss.backward()
        # the update was launched from inside backward on the optimiser's side stream (or left pending for step())
        # applied from inside backward by the single-pass kernel (or launched / left pending for step() by the older forms)
        assert model_b.fc1.weight.grad is None and (model_b.fc1.weight._pv_applied or opt._inflight
                                                    or model_b.fc1.weight._pv_pending is not None)
        opt.step()
        losses_b.append(float(loss))
    # fc1's dx comes from a different kernel in the single-pass form (same operands, another summation order): isolated
    # 1-ulp bf16 differences in dx, so the trajectories agree closely instead of bit for bit
    np.testing.assert_allclose(losses_a, losses_b, rtol=1e-5)
    # (Adam turns a sign flip of a near-zero gradient into a difference of up to 2 lr per step, so the LARGEST difference is
    # bounded by construction and says nothing: the mean distance and the share of weights a whole step apart do)
    for (k, a), (_, b) in zip(model_a.state_dict().items(), model_b.state_dict().items()):
        d = (a - b).abs()
        whole_step = (d > 5e-4).float().mean().item()
        assert d.mean().item() <= 0.1 * 5e-4 and whole_step <= 0.02, (k, d.mean().item(), whole_step)


def test_deferred_and_eager_fused_updates_agree(device, monkeypatch):
    """overlap_large_update=False (update inside step()) and True (update launched from backward on a side stream)
    give bit-identical parameters."""
    from predict_pv_yield_amd import optim
    from predict_pv_yield_amd.optim import HipAdam
    monkeypatch.setattr(HipAdam, "FUSE_MIN_NUMEL", 1)
    monkeypatch.setattr(optim, "FUSE_DX_INTO_UPDATE", False)   # both arms use the two-kernel form (dx kernel + update pass)
    sat, pv = _data(SMALL, 2)
    batch = None
    results = []
    for overlap in (False, True):
        _, model = _pair(SMALL, "bf16", device)
        opt = HipAdam(model.parameters(), lr=0.0005, overlap_large_update=overlap)
        batch = {"satellite": {"data": sat.to(device)}, "pv": {"pv_yield": pv.to(device)}}
        for _ in range(3):
            opt.zero_grad()
            model.training_step(batch, 0).backward()
            opt.step()
        torch.cuda.synchronize()
        results.append({k: v.clone() for k, v in model.state_dict().items()})
    for k in results[0]:
        assert torch.equal(results[0][k], results[1][k]), k


def test_bf16_gradient_side_channel_in_the_model(device, monkeypatch):
    """HipAdam large_grad_mode="bf16" (the data-parallel wire format), exercised in one process: fc1's gradient goes
    through the bf16 side channel and the sync callback; training follows the f32-gradient run closely."""
    from predict_pv_yield_amd.optim import HipAdam
    sat, pv = _data(SMALL, 2)
    _, model_a = _pair(SMALL, "bf16", device)
    losses_a = _hip_steps(model_a, sat, pv, 3, device)
    monkeypatch.setattr(HipAdam, "FUSE_MIN_NUMEL", 1)
    _, model_b = _pair(SMALL, "bf16", device)
    opt = model_b.configure_optimizers()
    opt.set_large_grad_mode("bf16")
    seen = []
    model_b.fc1.weight._pv_on_grad = lambda t: seen.append(t.dtype)
    batch = {"satellite": {"data": sat.to(device)}, "pv": {"pv_yield": pv.to(device)}}
    losses_b = []
    for _ in range(3):
        opt.zero_grad()
        loss = model_b.training_step(batch, 0)
        loss.backward()
        assert model_b.fc1.weight.grad is None and model_b.fc1.weight._pv_grad_bf16.dtype == torch.bfloat16
        opt.step()
        losses_b.append(float(loss))
    assert seen == [torch.bfloat16] * 3
    np.testing.assert_allclose(losses_b, losses_a, rtol=2e-3)
