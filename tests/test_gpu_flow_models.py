"""GPU parity of the general Conv3D / MaxPool3d / MSE kernels and of the two modules built on them — LitAutoEncoder
(notebooks/13_3d_conv_with_optical_flow_predictions.ipynb:962-1027) and Conv3dMaxPool
(predict_pv_yield/models/perceiver/perceiver_conv3d_nwp_sat.py:42-57) — against the golden vectors of the reference's
own source and against torch CPU ops on seeded inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.test_oracle_flow_models import GOLD, load_autoencoder_case

pytestmark = pytest.mark.gpu

CASES = [  # (B, Ci, Co, T, H, W, kernel, stride, padding)
    (2, 2, 16, 5, 20, 24, (2, 3, 3), (1, 1, 1), (0, 1, 1)),
    (2, 16, 32, 4, 17, 19, (2, 3, 3), (1, 1, 1), (0, 1, 1)),
    (2, 32, 1, 2, 20, 22, (2, 3, 3), (1, 2, 2), (0, 1, 1)),
    (1, 5, 7, 6, 11, 13, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 4, 20, 5, 12, 12, (3, 3, 3), (2, 2, 1), (1, 0, 2)),
    (2, 3, 6, 3, 9, 10, (1, 3, 3), (1, 2, 1), (0, 1, 0)),
    (2, 6, 5, 3, 8, 8, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    # 16..32 channels, unit stride, wide rows: the f32-MFMA wgrad (one and two row tiles, ragged channel counts)
    (2, 16, 32, 3, 12, 40, (2, 3, 3), (1, 1, 1), (0, 1, 1)),
    (1, 32, 32, 3, 9, 150, (2, 3, 3), (1, 1, 1), (0, 1, 1)),
    (1, 20, 24, 4, 8, 70, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (2, 17, 32, 2, 6, 33, (1, 3, 3), (1, 1, 1), (0, 1, 0)),
    # rows >= 64 columns wide, (2,3,3), unit stride: the f32-MFMA forward / dgrad (conv3d_fwd_mfma_f32) in its three
    # forms -- channel-half split (c_in > 16), tap-plane split (c_in = 16), 16-row tiles (c_out <= 16) -- with ragged
    # column tiles, several row segments and time padding
    (2, 16, 32, 3, 10, 70, (2, 3, 3), (1, 1, 1), (0, 1, 1)),    # fwd: tap-plane split; dgrad (32 -> 16): 16-row tiles
    (1, 24, 12, 4, 9, 100, (2, 3, 3), (1, 1, 1), (0, 1, 1)),    # fwd: 16-row tiles, ragged channel counts
    (1, 32, 16, 2, 40, 64, (2, 3, 3), (1, 1, 1), (0, 1, 1)),    # one full column tile, five row segments
    (2, 32, 32, 3, 7, 129, (2, 3, 3), (1, 1, 1), (1, 1, 1)),    # time padding, a single ragged column in the third tile
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("relu", [False, True])
def test_general_conv3d_fwd_bwd_vs_torch_cpu(case, relu, device):
    from predict_pv_yield_amd import functional as Fn
    b, ci, co, t, h, w, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(b, ci, t, h, w, generator=g)
    wt = torch.randn(co, ci, *k, generator=g) * 0.2
    bias = torch.randn(co, generator=g)
    xr, wr, br = (a.clone().requires_grad_(True) for a in (x, wt, bias))
    ref = F.conv3d(xr, wr, br, stride=s, padding=p)
    ref = F.relu(ref) if relu else ref
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    xh, wh, bh = (a.to(device).requires_grad_(True) for a in (x, wt, bias))
    out = Fn.conv3d_general_f32(xh, wh, bh, stride=s, padding=p, relu=relu)
    assert out.shape == ref.shape
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-5)
    out.backward(dy.to(device))
    torch.testing.assert_close(xh.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(wh.grad.cpu(), wr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(bh.grad.cpu(), br.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("shape,k,s,p", [((2, 3, 5, 13, 16), 3, (1, 2, 2), 1), ((1, 2, 4, 8, 9), (2, 3, 3), (2, 1, 2), (1, 1, 0)),
                                         ((1, 1, 3, 6, 6), 2, None, 0)])
def test_maxpool3d_vs_torch_cpu(shape, k, s, p, device):
    from predict_pv_yield_amd import functional as Fn
    g = torch.Generator().manual_seed(5)
    x = torch.randn(shape, generator=g)
    x[0, 0, 1, 2, 2:5] = 1.5                     # ties inside a window: the first maximum takes the gradient
    x[-1, -1, 0, 0, 0] = float("nan")            # NaN propagates
    xr = x.clone().requires_grad_(True)
    ref = F.max_pool3d(xr, k, s, p)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    xh = x.to(device).requires_grad_(True)
    out = Fn.maxpool3d_f32(xh, k, s, p)
    assert np.array_equal(out.detach().cpu().numpy(), ref.detach().numpy(), equal_nan=True)
    out.backward(dy.to(device))
    torch.testing.assert_close(xh.grad.cpu(), xr.grad, rtol=1e-6, atol=1e-6)


def test_mse_loss_kernel(device):
    from predict_pv_yield_amd import functional as Fn
    g = torch.Generator().manual_seed(3)
    a, b = torch.randn(7, 33, 31, generator=g), torch.randn(7, 33, 31, generator=g)
    ar = a.clone().requires_grad_(True)
    ref = F.mse_loss(ar, b)
    (ref * 3.0).backward()
    ah = a.to(device).requires_grad_(True)
    out = Fn.mse_loss(ah, b.to(device))
    assert abs(float(out) - float(ref)) <= 1e-6 * abs(float(ref))
    (out * 3.0).backward()
    torch.testing.assert_close(ah.grad.cpu(), ar.grad, rtol=1e-6, atol=1e-9)


def test_conv3d_maxpool_against_reference_golden(device):
    from predict_pv_yield_amd.models.perceiver.perceiver_conv3d_nwp_sat import Conv3dMaxPool
    g = np.load(GOLD)
    block = Conv3dMaxPool(out_channels=8, in_channels=3)
    sd = {k[len("mp/init/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("mp/init/")}
    assert list(sd) == list(block.state_dict())
    block.load_state_dict(sd)
    block.to(device)
    x = torch.from_numpy(g["mp/x"]).to(device).requires_grad_(True)
    y = block(x)
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["mp/y"], rtol=1e-4, atol=1e-5)
    y.backward(torch.from_numpy(g["mp/dy"]).to(device))
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["mp/dx"], rtol=1e-4, atol=1e-5)
    for k, p in block.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), g[f"mp/grad/{k}"], rtol=1e-4, atol=1e-4, err_msg=k)


def _hip_autoencoder(oracle, device):
    from predict_pv_yield_amd.models.conv3d.flow_autoencoder import LitAutoEncoder
    model = LitAutoEncoder()
    assert list(model.state_dict()) == list(oracle.state_dict())
    model.load_state_dict(oracle.state_dict())
    return model.to(device)


def test_autoencoder_against_notebook_golden(device):
    from predict_pv_yield_amd.models.conv3d import flow_autoencoder as fa
    g = np.load(GOLD)
    oracle, (hist, pred, hz, target) = load_autoencoder_case(g)
    model = _hip_autoencoder(oracle, device)
    batch = {fa.HISTORICAL_SAT_IMAGES: hist.to(device), fa.OPTICAL_FLOW_PREDICTIONS: pred.to(device),
             fa.FORECAST_HORIZON: hz.to(device), fa.TARGET_SAT_IMAGE: target.to(device)}
    y = model(batch)
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["ae/y_hat"], rtol=1e-4, atol=1e-5)
    opt = model.configure_optimizers()
    losses = []
    for step in range(3):
        opt.zero_grad()
        loss = model.training_step(batch, 0)
        loss.backward()
        if step == 0:
            for k, p in model.named_parameters():
                ref = g[f"ae/grad/{k}"]
                assert np.abs(p.grad.cpu().numpy() - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-7, k
        opt.step()
        losses.append(float(loss.detach()))
    np.testing.assert_allclose(losses, g["ae/losses"], rtol=1e-4)
    for k, p in model.named_parameters():   # three Adam steps of lr 1e-4: within a fraction of one step
        assert np.abs(p.detach().cpu().numpy() - g[f"ae/step3/{k}"]).max() <= 4e-5, k


def test_autoencoder_full_size_step_vs_oracle(device):
    """[B,2,5,128,128] -> [B,1,1,64,64] (the notebook's shapes), B=2: loss and gradients vs the torch-CPU oracle."""
    from oracle import conv3d_oracle as co
    from predict_pv_yield_amd.models.conv3d import flow_autoencoder as fa
    torch.manual_seed(11)
    oracle = co.OracleLitAutoEncoder()
    model = _hip_autoencoder(oracle, device)
    g = torch.Generator().manual_seed(12)
    hist, pred = torch.randn(2, 4, 128, 128, generator=g), torch.randn(2, 128, 128, generator=g)
    hz = torch.tensor([float(fa.normalise_forecast_horizon(300.0)), float(fa.normalise_forecast_horizon(3600.0))])
    target = torch.randn(2, 64, 64, generator=g)
    ref = oracle.loss(hist, pred, hz, target)
    ref.backward()
    batch = {fa.HISTORICAL_SAT_IMAGES: hist.to(device), fa.OPTICAL_FLOW_PREDICTIONS: pred.to(device),
             fa.FORECAST_HORIZON: hz.to(device), fa.TARGET_SAT_IMAGE: target.to(device)}
    loss = model.training_step(batch, 0)
    assert model(batch).shape == (2, 1, 1, 64, 64)
    loss.backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref))
    for (k, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
        assert (p.grad.cpu() - q.grad).abs().max().item() <= 1e-3 * q.grad.abs().max().item() + 1e-7, k


def test_autoencoder_on_cpu_fails_loudly():
    from predict_pv_yield_amd.models.conv3d import flow_autoencoder as fa
    model = fa.LitAutoEncoder()
    with pytest.raises(RuntimeError, match="MI355X"):
        model({fa.HISTORICAL_SAT_IMAGES: torch.zeros(1, 4, 8, 8), fa.OPTICAL_FLOW_PREDICTIONS: torch.zeros(1, 8, 8),
               fa.FORECAST_HORIZON: torch.zeros(1)})
