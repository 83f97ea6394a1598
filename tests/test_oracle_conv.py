"""CPU: the torch-CPU oracle (oracle/conv3d_oracle.py) against golden vectors produced by the REFERENCE's own
module source (tests/golden/make_conv3d_golden.py -> conv3d_small.npz), plus known-answer tests of the loss and
Adam restatements.  This is what pins the oracle the GPU parity tests are judged against."""
import os

import numpy as np
import pytest
import torch

from oracle import conv3d_oracle as co

GOLD = os.path.join(os.path.dirname(__file__), "golden", "conv3d_small.npz")
KW = dict(include_pv_yield=False, include_nwp=False, forecast_minutes=30, history_minutes=30,
          number_of_conv3d_layers=4, conv3d_channels=32, image_size_pixels=10, number_sat_channels=11,
          fc1_output_features=16, fc2_output_features=16, fc3_output_features=16)
KW_PV = dict(KW, include_pv_yield=True, include_nwp=True, output_variable="gsp_yield", forecast_minutes=60)


def checksum(t, n=64):
    f = t.detach().double().flatten()
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long()
    return np.concatenate([[f.sum().item(), f.abs().sum().item()], t.detach().flatten()[idx].double().numpy()])


@pytest.mark.parametrize("tag,kw", [("sat_only", KW), ("pv_nwp", KW_PV)])
def test_oracle_matches_reference_module(tag, kw):
    g = np.load(GOLD)
    model = co.OracleConv3dModel(**kw)
    sd = {k[len(f"{tag}/init/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{tag}/init/")}
    assert list(sd) == list(model.state_dict())          # same parameter names, same order
    model.load_state_dict(sd)
    attrs = [model.cnn_output_size, model.forecast_len, model.history_len_5, model.forecast_len_5,
             model.history_len_30, model.forecast_len_30, model.history_len_60, model.number_of_samples_per_batch]
    assert attrs == list(g[f"{tag}/attrs"])
    sat, pv, gsp, nwp = (torch.from_numpy(g[f"{tag}/{k}"]) for k in ("sat", "pv", "gsp", "nwp"))
    yld = gsp if kw.get("output_variable") == "gsp_yield" else pv
    args = (sat, yld, nwp) if kw["include_pv_yield"] else (sat,)
    y_hat = model(*args)
    assert np.array_equal(y_hat.detach().numpy(), g[f"{tag}/y_hat"])   # bit-exact on the same torch build
    opt = co.make_optimizer(model)
    losses = []
    for step in range(3):
        opt.zero_grad()
        y_hat = model(*args)
        mse, nmae, mse_exp, mae_exp = co.forecast_losses(y_hat, co.select_target(yld, model.forecast_len))
        nmae.backward()
        if step == 0:
            np.testing.assert_allclose([float(mse), float(nmae), float(mse_exp), float(mae_exp)], g[f"{tag}/logged"],
                                       rtol=1e-6)
            for k, p in model.named_parameters():
                np.testing.assert_array_equal(checksum(p.grad), g[f"{tag}/grad/{k}"], err_msg=k)
        opt.step()
        losses.append(float(nmae.detach()))
        if step in (0, 2):
            for k, p in model.named_parameters():
                np.testing.assert_array_equal(checksum(p), g[f"{tag}/step{step + 1}/{k}"], err_msg=k)
    np.testing.assert_array_equal(np.array(losses), g[f"{tag}/losses"])


SN = dict(include_pv_or_gsp_yield_history=False, include_nwp=True, forecast_minutes=60, history_minutes=60,
          number_of_conv3d_layers=4, conv3d_channels=32, image_size_pixels=10, nwp_image_size_pixels=10,
          number_sat_channels=11, number_nwp_channels=10, fc1_output_features=16, fc2_output_features=16,
          fc3_output_features=16, output_variable="gsp_yield", include_pv_yield_history=True)
SN_PV = dict(SN, include_pv_or_gsp_yield_history=True, output_variable="pv_yield", include_future_satellite=False,
             include_pv_yield_history=False, forecast_minutes=30)


def sat_nwp_case(g, tag, kw, **extra):
    """(oracle model with the golden initial parameters, forward args, target yield) of one sat+nwp golden case."""
    model = co.OracleConv3dSatNwpModel(**kw, **extra)
    sd = {k[len(f"{tag}/init/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{tag}/init/")}
    assert list(sd) == list(model.state_dict())
    model.load_state_dict(sd)
    t = {k: torch.from_numpy(g[f"{tag}/{k}"]) for k in ("sat", "pv", "gsp", "nwp", "row", "gsp_id")}
    gsp_out = kw["output_variable"] == "gsp_yield"
    args = (t["sat"], t["pv"], t["gsp"], t["nwp"], t["gsp_id"] if gsp_out else t["row"])
    return model, args, (t["gsp"] if gsp_out else t["pv"]), t


@pytest.mark.parametrize("tag,kw", [("sat_nwp", SN), ("sat_nwp_pv", SN_PV)])
def test_sat_nwp_oracle_matches_reference_module(tag, kw):
    """model_sat_nwp.py executed from the reference's own source (golden) vs OracleConv3dSatNwpModel: bit-exact."""
    g = np.load(GOLD)
    model, args, yld, _ = sat_nwp_case(g, tag, kw)
    assert [model.cnn_output_size, model.nwp_cnn_output_size, model.forecast_len, model.fc3.in_features] == list(g[f"{tag}/attrs"])
    assert np.array_equal(model(*args).detach().numpy(), g[f"{tag}/y_hat"])
    opt = co.make_optimizer(model)
    losses = []
    for step in range(3):
        opt.zero_grad()
        _, nmae, _, _ = co.forecast_losses(model(*args), co.select_target(yld, model.forecast_len))
        nmae.backward()
        if step == 0:
            for k, p in model.named_parameters():
                np.testing.assert_array_equal(checksum(p.grad), g[f"{tag}/grad/{k}"], err_msg=k)
        opt.step()
        losses.append(float(nmae.detach()))
    for k, p in model.named_parameters():
        np.testing.assert_array_equal(checksum(p), g[f"{tag}/step3/{k}"], err_msg=k)
    np.testing.assert_array_equal(np.array(losses), g[f"{tag}/losses"])


def test_timestep_arithmetic_matches_base_model():
    # base_model.py:41-73; the default Model() sees T = 19, BASELINE's "12 -> 6" is history_minutes = 55 (T = 18)
    d = co.timestep_arithmetic(60, 30)
    assert (d["history_len_5"], d["forecast_len_5"], d["history_len_30"], d["forecast_len_30"], d["history_len_60"]) == (12, 6, 2, 1, 1)
    assert d["forecast_len"] == 6 and d["number_of_samples_per_batch"] == 128
    d = co.timestep_arithmetic(55, 30)
    assert d["history_len_5"] + d["forecast_len_5"] + 1 == 18
    d = co.timestep_arithmetic(30, 120, "gsp_yield")
    assert d["forecast_len"] == 4 and d["history_len_60"] == 1 and d["number_of_samples_per_batch"] == 32
    m = co.OracleConv3dModel(include_pv_yield=False, include_nwp=False, number_sat_channels=11, fc1_output_features=1)
    assert m.cnn_output_size == 32 * 56 * 56 * 11 == 1103872          # model.py:74-78 at the reference defaults
    m18 = co.OracleConv3dModel(include_pv_yield=False, include_nwp=False, number_sat_channels=11, history_minutes=55,
                                 fc1_output_features=1)
    assert m18.cnn_output_size == 1003520


def test_weighted_losses_kat():
    w = co.weighted_losses_weights(3)
    raw = np.array([1.0, 0.5, 0.25])
    np.testing.assert_allclose(w.numpy(), raw / raw.sum() * 3, rtol=1e-6)
    assert abs(float(w.mean()) - 1.0) < 1e-6
    y_hat = torch.tensor([[1.0, 2.0, 3.0]])
    y = torch.zeros(1, 3)
    mse, nmae, mse_exp, mae_exp = co.forecast_losses(y_hat, y)
    assert abs(float(mse) - 14 / 3) < 1e-6 and abs(float(nmae) - 2.0) < 1e-6
    assert abs(float(mae_exp) - float((w * torch.tensor([1.0, 2.0, 3.0])).mean())) < 1e-6


def test_select_target_is_first_site_last_steps():
    y = torch.arange(2 * 5 * 3, dtype=torch.float32).reshape(2, 5, 3)
    t = co.select_target(y, 2)
    assert t.tolist() == [[9.0, 12.0], [24.0, 27.0]]


def test_bf16_emulation_rounds_only_operands():
    torch.manual_seed(0)
    a = co.OracleConv3dModel(**KW)
    b = co.OracleConv3dModel(**KW, emulate_bf16=True)
    b.load_state_dict(a.state_dict())
    sat = torch.randn(2, 11, 13, 10, 10)
    ya, yb = a(sat), b(sat)
    assert not torch.equal(ya, yb)
    torch.testing.assert_close(ya, yb, rtol=5e-2, atol=5e-3)


# ---- NWP-only model (predict_pv_yield/models/conv3d/model_nwp.py) ---------------------------------------------------
GOLD_NWP = os.path.join(os.path.dirname(__file__), "golden", "conv3d_nwp_small.npz")
NWP = dict(forecast_minutes=120, history_minutes=30, number_of_conv3d_layers=4, conv3d_channels=32, nwp_image_size_pixels=12,
           number_nwp_channels=10, fc1_output_features=16, fc2_output_features=16, fc3_output_features=16,
           output_variable="gsp_yield")
NWP_1CH = dict(NWP, number_nwp_channels=1, number_of_conv3d_layers=2, nwp_image_size_pixels=6, output_variable="pv_yield",
               forecast_minutes=60, embedding_dem=0, include_pv_yield_history=False)


def nwp_case(g, tag, kw, **extra):
    """(oracle model with the golden initial parameters, tensors) of one NWP-only golden case."""
    model = co.OracleConv3dNwpModel(**kw, **extra)
    sd = {k[len(f"{tag}/init/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{tag}/init/")}
    assert list(sd) == list(model.state_dict())
    model.load_state_dict(sd)
    return model, {k: torch.from_numpy(g[f"{tag}/{k}"]) for k in ("pv", "gsp", "nwp")}


@pytest.mark.parametrize("tag,kw", [("nwp", NWP), ("nwp_1ch", NWP_1CH)])
def test_nwp_oracle_matches_reference_module(tag, kw):
    """model_nwp.py executed from the reference's own source (golden) vs OracleConv3dNwpModel: bit-exact, including
    the parameters forward never touches (no gradient, unchanged after three Adam steps)."""
    g = np.load(GOLD_NWP)
    model, t = nwp_case(g, tag, kw)
    yld = t["gsp"] if kw["output_variable"] == "gsp_yield" else t["pv"]
    assert [model.nwp_cnn_output_size, model.forecast_len, model.fc3.in_features] == list(g[f"{tag}/attrs"])
    assert np.array_equal(model(t["nwp"]).detach().numpy(), g[f"{tag}/y_hat"])
    opt = co.make_optimizer(model)
    losses = []
    for step in range(3):
        opt.zero_grad()
        _, nmae, _, _ = co.forecast_losses(model(t["nwp"]), co.select_target(yld, model.forecast_len))
        nmae.backward()
        if step == 0:
            for k, p in model.named_parameters():
                if p.grad is None:
                    assert f"{tag}/grad/{k}" not in g.files and k.split(".")[0] in ("pv_system_id_embedding", "pv_fc1")
                else:
                    np.testing.assert_array_equal(checksum(p.grad), g[f"{tag}/grad/{k}"], err_msg=k)
        opt.step()
        losses.append(float(nmae.detach()))
    for k, p in model.named_parameters():
        np.testing.assert_array_equal(checksum(p), g[f"{tag}/step3/{k}"], err_msg=k)
    np.testing.assert_array_equal(np.array(losses), g[f"{tag}/losses"])


def test_bf16_gradient_distance_split_into_relu_flips_and_rounding():
    """VERDICT r4 (weak point 3 / next 4): the bf16 path's conv-weight gradients sit 3-11 % (norm-wise) from the pure-f32
    oracle's at the benched model size -- the loosest bound of the suite (tests/test_gpu_headline.py, 15 %).  Which part is
    operand rounding and which part is ReLU units that flip?  Three backward passes of the oracle on one batch of the benched
    model (B = 4, T = 18, 64 px):
      f32      the reference's arithmetic;
      bf16     the bf16-emulating oracle (the values the MFMA path rounds; the HIP kernels follow it to 2 %);
      bf16|m   the same rounding, but every ReLU of the conv tower and fc1 uses the f32 run's mask (no unit can flip).
    dist(bf16, f32) is the figure the GPU test bounds; dist(bf16|m, f32) is rounding alone; the rest is flips.  Asserted: the
    total is inside the GPU test's bound, and rounding alone stays below 2 % -- i.e. the distance the 15 % bound admits is the
    signature of units within bf16 rounding of zero, not lost precision of the products."""
    import torch.nn.functional as F
    kw = dict(include_pv_yield=False, include_nwp=False, forecast_minutes=30, history_minutes=55, number_of_conv3d_layers=4,
              conv3d_channels=32, image_size_pixels=64, number_sat_channels=11, fc1_output_features=128, fc2_output_features=128,
              fc3_output_features=64)
    torch.manual_seed(518)
    model = co.OracleConv3dModel(**kw)
    g = torch.Generator().manual_seed(3)
    sat = torch.randn(4, 11, 18, 64, 64, generator=g)
    pv = torch.rand(4, 18, 128, generator=g)
    y = co.select_target(pv, model.forecast_len)
    convs = [model.sat_conv0] + [getattr(model, f"conv3d_{i + 1}") for i in range(3)]

    def grads(rounding, masks=None):
        model.zero_grad()
        out = co.bf16_round(sat) if rounding else sat
        used = []
        for i, layer in enumerate(convs):
            w = co._RoundWeightBF16.apply(layer.weight) if rounding else layer.weight
            z = F.conv3d(out, w, layer.bias)
            m = (z > 0) if masks is None else masks[i]
            used.append(m)
            out = z * m
            if rounding:
                out = co._RoundBF16.apply(out)
        out = out.reshape(4, model.cnn_output_size)
        z = F.linear(out, co._RoundWeightBF16.apply(model.fc1.weight) if rounding else model.fc1.weight, model.fc1.bias)
        m = (z > 0) if masks is None else masks[4]
        used.append(m)
        out = F.relu(model.fc2(z * m))
        out = model.fc4(F.relu(model.fc3(out))).reshape(4, model.forecast_len)
        co.forecast_losses(out, y)[1].backward()
        return {k: p.grad.clone() for k, p in model.named_parameters()}, used

    g_f32, masks_f32 = grads(False)
    g_b16, masks_b16 = grads(True)
    g_b16m, _ = grads(True, masks_f32)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    flips = [float((a != b).float().mean()) for a, b in zip(masks_f32, masks_b16)]
    rows = []
    for k in g_f32:
        if "conv" in k and k.endswith("weight"):
            total, rounding = rel(g_b16[k], g_f32[k]), rel(g_b16m[k], g_f32[k])
            rows.append((k, total, rounding))
            assert total <= 0.15, (k, total)                # the GPU test's bound for the conv layers
            assert rounding <= 0.02, (k, rounding)          # operand rounding alone
            assert rounding <= total + 1e-6
    print("\nconv weight gradients, bf16 vs f32 (norm-wise): total / rounding alone (ReLU masks of the f32 run forced)")
    for k, total, rounding in rows:
        print(f"  {k:18s} {total:7.4f} / {rounding:7.4f}   -> flips account for {1 - (rounding / total) ** 2:5.1%} of the squared distance")
    print("  flipped ReLU units per layer (conv 1-4, fc1):", ["%.2e" % f for f in flips])
