"""CPU: oracle restatements of LitAutoEncoder (nb-13) and Conv3dMaxPool against golden vectors produced by the
reference's own source (tests/golden/make_flow_model_golden.py -> flow_models_small.npz): bit-exact."""
import os

import numpy as np
import torch

from oracle import conv3d_oracle as co

GOLD = os.path.join(os.path.dirname(__file__), "golden", "flow_models_small.npz")


def load_autoencoder_case(g):
    model = co.OracleLitAutoEncoder()
    sd = {k[len("ae/init/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("ae/init/")}
    assert list(sd) == list(model.state_dict())          # conv.0 / conv.2 / conv.4 / conv.6, same order
    model.load_state_dict(sd)
    args = tuple(torch.from_numpy(g[f"ae/{k}"]) for k in ("HISTORICAL_SAT_IMAGES", "OPTICAL_FLOW_PREDICTIONS",
                                                            "FORECAST_HORIZON", "TARGET_SAT_IMAGE"))
    return model, args


def test_autoencoder_oracle_matches_notebook_cell():
    g = np.load(GOLD)
    model, (hist, pred, hz, target) = load_autoencoder_case(g)
    y_hat = model(hist, pred, hz)
    assert y_hat.shape == (3, 1, 1, 16, 16)
    assert np.array_equal(y_hat.detach().numpy(), g["ae/y_hat"])
    loss = model.loss(hist, pred, hz, target)
    loss.backward()
    for k, p in model.named_parameters():
        np.testing.assert_array_equal(p.grad.numpy(), g[f"ae/grad/{k}"], err_msg=k)
    model.zero_grad()
    losses = model.train_steps(hist, pred, hz, target, 3)
    np.testing.assert_array_equal(np.array(losses), g["ae/losses"])
    for k, p in model.named_parameters():
        np.testing.assert_array_equal(p.detach().numpy(), g[f"ae/step3/{k}"], err_msg=k)


def test_autoencoder_full_size_shapes_and_macs():
    # [B,2,5,128,128] -> [B,1,1,64,64]; 28.9 k parameters; 1.097 GMAC per sample
    model = co.OracleLitAutoEncoder()
    assert sum(p.numel() for p in model.parameters()) == 28881
    with torch.no_grad():
        y = model(torch.zeros(1, 4, 128, 128), torch.zeros(1, 128, 128), torch.zeros(1))
    assert y.shape == (1, 1, 1, 64, 64)
    macs = (2 * 16 * 18 * 4 + 16 * 32 * 18 * 3 + 32 * 32 * 18 * 2) * 128 * 128 + 32 * 18 * 64 * 64
    assert abs(macs / 1e9 - 1.097) < 1e-3


def test_conv3d_maxpool_oracle_matches_reference_class():
    g = np.load(GOLD)
    block = co.OracleConv3dMaxPool(out_channels=8, in_channels=3)
    sd = {k[len("mp/init/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("mp/init/")}
    assert list(sd) == list(block.state_dict())
    block.load_state_dict(sd)
    x = torch.from_numpy(g["mp/x"]).requires_grad_(True)
    y = block(x)
    assert y.shape == (2, 8, 5, 7, 8)
    assert np.array_equal(y.detach().numpy(), g["mp/y"])
    y.backward(torch.from_numpy(g["mp/dy"]))
    assert np.array_equal(x.grad.numpy(), g["mp/dx"])
    for k, p in block.named_parameters():
        np.testing.assert_array_equal(p.grad.numpy(), g[f"mp/grad/{k}"], err_msg=k)
