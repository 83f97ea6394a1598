"""GPU parity of the Perceiver module (models/perceiver/perceiver_core.py) against the torch-CPU restatement of
perceiver_pytorch.Perceiver (oracle/perceiver_oracle.py; the package itself is absent and unpinned: parity with it is
unpinned, parity between the HIP path and the restatement is what these tests hold)."""
import pytest
import torch

from oracle import perceiver_oracle as po

pytestmark = pytest.mark.gpu

KW = dict(input_channels=11, input_axis=2, num_freq_bands=6, max_freq=10, depth=4, num_latents=32, latent_dim=64,
          num_classes=24, weight_tie_layers=True)


def _pair(kw, device, seed=0):
    from predict_pv_yield_amd.models.perceiver.perceiver_core import Perceiver
    torch.manual_seed(seed)
    oracle = po.OraclePerceiver(**kw)
    model = Perceiver(**kw)
    assert list(model.state_dict().keys()) == list(oracle.state_dict().keys())
    model.load_state_dict(oracle.state_dict())
    return oracle, model.to(device)


def test_position_features_match():
    from predict_pv_yield_amd.models.perceiver.perceiver_core import fourier_position_features
    a = fourier_position_features((5, 7), 10.0, 6)
    b = po.position_encoding((5, 7), 10.0, 6)
    assert a.shape == (5, 7, 26) and torch.equal(a, b)


def test_weight_tying_structure(device):
    oracle, model = _pair(KW, device)
    assert model.layers[1][0] is model.layers[3][0] and model.layers[0][0] is not model.layers[1][0]
    n_unique = sum(p.numel() for p in model.parameters())
    assert n_unique == sum(p.numel() for p in oracle.parameters())
    assert any(k.startswith("layers.3.0.fn.to_kv") for k in model.state_dict())       # tied blocks appear under every layer


@pytest.mark.parametrize("tie", [True, False])
def test_forward_and_gradients_vs_oracle(tie, device):
    kw = dict(KW, weight_tie_layers=tie, depth=3)
    oracle, model = _pair(kw, device, seed=1)
    g = torch.Generator().manual_seed(2)
    data = torch.randn(3, 12, 10, 11, generator=g)
    ref = oracle(data)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    out = model(data.to(device))
    assert out.shape == (3, 24)
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=2e-4, atol=2e-4)
    out.backward(dy.to(device))
    for (k, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
        scale = q.grad.abs().max().item() + 1e-12
        assert (p.grad.cpu() - q.grad).abs().max().item() <= 2e-3 * scale + 1e-6, k


def test_reference_shape_forward(device):
    """perceiver.py:70-80 as configured by the reference: 11 channels, 64 x 64, 128 latents x 64, 512 classes, tied."""
    kw = dict(input_channels=11, input_axis=2, num_freq_bands=6, max_freq=10, depth=3, num_latents=128, latent_dim=64,
              num_classes=512, weight_tie_layers=True)
    oracle, model = _pair(kw, device, seed=3)
    data = torch.randn(2, 64, 64, 11, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        ref = oracle(data)
        out = model(data.to(device))
    torch.testing.assert_close(out.cpu(), ref, rtol=2e-4, atol=2e-4)


def test_on_cpu_fails_loudly():
    from predict_pv_yield_amd.models.perceiver.perceiver_core import Perceiver
    with pytest.raises(RuntimeError, match="MI355X"):
        Perceiver(**KW)(torch.zeros(1, 4, 4, 11))
