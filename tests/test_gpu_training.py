"""GPU: the reference's end-to-end harnesses driven against this package through the same surface
(tests/test_training.py::test_train of the reference: compose -> train(config); model tests: Trainer.fit/predict)."""
import os

import numpy as np
import pytest
import torch

from oracle import flow_oracle as fo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_example_simple(device, tmp_path, monkeypatch):
    """reference tests/test_training.py:8-25: logger=csv, experiment=example_simple, fake data, fast_dev_run."""
    from predict_pv_yield_amd import hydra_lite as H
    from predict_pv_yield_amd.training import train
    monkeypatch.chdir(tmp_path)
    cfg = H.compose(os.path.join(ROOT, "configs"), "config",
                    ["logger=csv", "experiment=example_simple", "datamodule.fake_data=true",
                     f"datamodule.data_path={ROOT}/tests/configs/dataset", "trainer.fast_dev_run=true", "trainer.gpus=1"])
    train(config=cfg)


def test_train_conv3d_experiment_two_epochs(device, tmp_path, monkeypatch):
    from predict_pv_yield_amd import hydra_lite as H
    from predict_pv_yield_amd.training import train
    monkeypatch.chdir(tmp_path)
    cfg = H.compose(os.path.join(ROOT, "configs"), "config",
                    ["experiment=conv3d", "model.image_size_pixels=16", "model.history_minutes=30",
                     "model.forecast_minutes=120", "model.fc1_output_features=16", "datamodule.n_train_data=3",
                     f"datamodule.data_path={ROOT}/tests/configs/dataset", "trainer.max_epochs=2",
                     "optimized_metric=MSE/Validation_epoch", "test_after_training=false"])
    score = train(config=cfg)
    assert np.isfinite(score)
    assert os.path.exists("checkpoints/last.ckpt")
    ck = torch.load("checkpoints/last.ckpt")
    assert "sat_conv0.weight" in ck["state_dict"] and "fc4.bias" in ck["state_dict"]
    assert set(ck["optimizer_states"][0]["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}   # torch.optim.Adam keys


def test_model_trainer_fit_and_predict(device):
    """reference tests/models/conv3d/test_conv3d_model.py:42-62."""
    from predict_pv_yield_amd import lightning as pl
    from predict_pv_yield_amd.data.fake import FakeDataConfiguration, FakeDataset
    from predict_pv_yield_amd.models.conv3d.model import Model
    from predict_pv_yield_amd.utils import load_config
    config = load_config("tests/configs/model/conv3d.yaml")
    model = Model(**config)
    ds = FakeDataset(FakeDataConfiguration(batch_size=2, history_minutes=60, forecast_minutes=60,
                                           satellite_image_size_pixels=16), length=2)
    loader = torch.utils.data.DataLoader(ds, batch_size=None)
    trainer = pl.Trainer(gpus=1, max_epochs=1)
    trainer.fit(model, loader)
    out = trainer.predict(model, loader)
    assert len(out) == 2 and out[0].shape == (2, model.forecast_len_5)


def test_trainer_in_hip_graph_mode_equals_the_eager_trainer_on_the_conv3d_model(device):
    """Trainer(hip_graph=True) on the Conv3D model fed BatchML batches (attribute containers, the large-layer fused backward in
    its capturable form): parameters and metrics of the eager fit, bit for bit."""
    from predict_pv_yield_amd import lightning as pl
    from predict_pv_yield_amd.data.fake import FakeDataConfiguration, FakeDataset
    from predict_pv_yield_amd.models.conv3d.model import Model
    from predict_pv_yield_amd.utils import load_config
    config = load_config("tests/configs/model/conv3d.yaml")
    results = []
    for graph in (False, True):
        torch.manual_seed(33)
        model = Model(**config)
        ds = FakeDataset(FakeDataConfiguration(batch_size=2, history_minutes=60, forecast_minutes=60,
                                               satellite_image_size_pixels=16), length=7)
        loader = torch.utils.data.DataLoader(ds, batch_size=None)
        trainer = pl.Trainer(gpus=1, max_epochs=1, hip_graph=graph, log_every_n_steps=1)
        trainer.fit(model, loader)
        results.append(({k: v.detach().clone() for k, v in model.state_dict().items()}, dict(trainer.callback_metrics)))
    (p0, m0), (p1, m1) = results
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k
    assert m0.keys() == m1.keys() and all(m0[k] == m1[k] for k in m0), (m0, m1)


def test_advect_future_frames_matches_oracle(device):
    from predict_pv_yield_amd import optical_flow as of
    from predict_pv_yield_amd.data.synthetic import advected_counts
    raw, vel = advected_counts(batch=2, t=6, channels=3, h=64, w=64, seed=21, vmax=2.0)
    mean, std = of.SAT_MEAN[1:4], of.SAT_STD[1:4]
    got = of.advect_future_frames(torch.from_numpy(raw).to(device), n_future=3, mean=torch.from_numpy(mean).to(device),
                                  std=torch.from_numpy(std).to(device)).cpu().numpy()
    ref = fo.advect_frames(raw, mean, std, n_future=3)
    assert got.shape == ref.shape == (2, 3, 9, 64, 64)
    assert np.array_equal(got[:, :, :6], ref[:, :, :6])                       # normalisation: bit-exact
    assert np.abs(got[:, :, 6:] - ref[:, :, 6:]).max() < 5e-3                  # warp of a 1e-3-px-accurate flow
    # the advected frame is a better predictor of the true next texture position than persistence
    t0 = ref[:, :, 5]
    assert np.abs(got[:, :, 6] - t0).mean() > 1e-3


def test_cv2_style_api(device):
    from predict_pv_yield_amd import optical_flow as of
    from predict_pv_yield_amd.data.synthetic import advected_counts
    raw, _ = advected_counts(batch=1, t=4, channels=1, h=64, w=64, seed=2)
    stack = raw[0, :, 0]
    u8 = of.convert_10bpp_to_uint8(stack)
    assert isinstance(u8, np.ndarray) and u8.dtype == np.uint8
    assert np.array_equal(u8, fo.convert_10bpp_to_uint8(stack, 0)[0])
    flow = of.calcOpticalFlowFarneback(u8[0], u8[1], None, 0.5, 2, 40, 3, 5, 0.7, of.OPTFLOW_FARNEBACK_GAUSSIAN)
    ref = fo.calc_optical_flow_farneback(u8[0], u8[1])
    assert flow.shape == (64, 64, 2) and np.abs(flow - ref).max() <= 1e-3
    flows = of.compute_optical_flow(stack)
    assert flows.shape == (3, 64, 64, 2) and np.abs(flows[0] - ref).max() <= 1e-3
    wavg = of.weighted_average(flows)
    assert np.array_equal(wavg, np.average(flows, axis=0, weights=range(1, 4)).astype(np.float32))
    img = stack[0].astype(np.float32)
    warped = of.remap_image(img, wavg)
    refw = fo.remap_image(img, wavg, 1.0, fo.BORDER_CONSTANT, np.nan)
    assert np.array_equal(np.isnan(warped), np.isnan(refw)) and np.array_equal(warped[~np.isnan(refw)], refw[~np.isnan(refw)])
    preds, index = of.compute_optical_flow_predictions(stack.astype(np.float32), flows)
    assert preds.shape == (6, 64, 64) and index.tolist() == [[0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3]]
    r = fo.remap_image(stack[1].astype(np.float32), flows[1], 2.0, fo.BORDER_CONSTANT, np.nan)
    m = ~np.isnan(r)
    assert np.array_equal(preds[4][m], r[m])
    with pytest.raises(AssertionError):
        of.convert_10bpp_to_uint8(np.array([4000, 1, 2, 3, 4, 5, 6, 7], np.int16))


def test_model_with_optical_flow_future_frames(device):
    from predict_pv_yield_amd.models.conv3d.model import Model
    kw = dict(include_pv_yield=False, include_nwp=False, forecast_minutes=30, history_minutes=25, image_size_pixels=32,
              number_sat_channels=2, fc1_output_features=8, fc2_output_features=8, fc3_output_features=8)
    torch.manual_seed(0)
    m_true = Model(**kw, precision="fp32", future_frames="true").to(device)
    m_flow = Model(**kw, precision="fp32", future_frames="optical_flow").to(device)
    m_flow.load_state_dict(m_true.state_dict())
    sat = torch.randn(2, 2, 12, 32, 32, device=device)
    pv = torch.rand(2, 12, 128, device=device)
    batch = {"satellite": {"data": sat}, "pv": {"pv_yield": pv}}
    y_true, y_flow = m_true(batch), m_flow(batch)
    assert y_flow.shape == (2, 6) and torch.isfinite(y_flow).all()
    assert not torch.equal(y_true, y_flow)       # the 6 future slices were replaced
    # the observed slices are untouched by the replacement
    from predict_pv_yield_amd.optical_flow import replace_future_frames_with_flow
    rep = replace_future_frames_with_flow(sat, n_future=6)
    assert torch.equal(rep[:, :, :6], sat[:, :, :6]) and not torch.equal(rep[:, :, 6:], sat[:, :, 6:])


def test_matched_validation_nmae_after_training(device):
    """BASELINE 'at matched validation NMAE': the bf16 MFMA path and the f32 torch-CPU oracle, trained for 40 Adam steps
    from the same initial weights on the same learnable synthetic task (PV yield = a smooth function of the mean
    brightness of the last observed frames), reach the same validation NMAE (within 2 % relative after 40 bf16 steps) and
    both improve on the untrained model."""
    from oracle import conv3d_oracle as co
    from predict_pv_yield_amd.models.conv3d.model import Model
    kw = dict(include_pv_yield=False, include_nwp=False, forecast_minutes=30, history_minutes=30,
              number_of_conv3d_layers=2, conv3d_channels=32, image_size_pixels=12, number_sat_channels=11,
              fc1_output_features=16, fc2_output_features=16, fc3_output_features=16)

    def make(n, seed):
        g = torch.Generator().manual_seed(seed)
        sat = torch.randn(n, 11, 13, 12, 12, generator=g)
        level = sat[:, :3, 4:7].mean(dim=(1, 2, 3, 4))                       # [n]
        pv = torch.rand(n, 13, 128, generator=g)
        steps = torch.arange(6, dtype=torch.float32)
        pv[:, -6:, 0] = torch.sigmoid(4.0 * level[:, None] + 0.2 * steps[None])   # the target slice y[:, -6:, 0]
        return sat, pv

    train_sat, train_pv = make(16, 1)
    val_sat, val_pv = make(16, 2)
    torch.manual_seed(3)
    oracle = co.OracleConv3dModel(**kw)
    model = Model(**kw, precision="bf16")
    model.load_state_dict(oracle.state_dict())
    model.to(device)

    def val_nmae_oracle():
        with torch.no_grad():
            return float((oracle(val_sat) - co.select_target(val_pv, 6)).abs().mean())

    def val_nmae_hip():
        with torch.no_grad():
            y = model({"satellite": {"data": val_sat.to(device)}, "pv": {"pv_yield": val_pv.to(device)}})
            return float((y.cpu() - co.select_target(val_pv, 6)).abs().mean())

    before = val_nmae_oracle()
    assert abs(val_nmae_hip() - before) < 5e-3
    co.train_steps(oracle, train_sat, train_pv, 40)
    opt = model.configure_optimizers()
    batch = {"satellite": {"data": train_sat.to(device)}, "pv": {"pv_yield": train_pv.to(device)}}
    for _ in range(40):
        opt.zero_grad()
        model.training_step(batch, 0).backward()
        opt.step()
    ref, got = val_nmae_oracle(), val_nmae_hip()
    assert ref < 0.95 * before and got < 0.95 * before, (before, ref, got)      # both learned (16 samples: modest)
    assert abs(got - ref) <= 2e-2 * ref, (ref, got)                            # and ended at the same validation NMAE


def test_validation_results_csv_and_horizon_metrics_on_the_gpu(device, tmp_path):
    """SURVEY §8f row 4 on the device path: Trainer.validate of the Conv3D model (gsp_yield output) writes
    `{results_file_name}_0.csv` with n_batches * batch * forecast_len_30 rows / the five reference columns
    (reference tests/models/baseline/test_baseline_model_gsp.py:81-111), and logs, under `MSE_forecast_horizon_i`,
    the per-horizon MAE (base_model.py:121-141) -- both compared with the oracle's restatement on the model's outputs."""
    import pandas as pd
    from oracle import conv3d_oracle as co
    from predict_pv_yield_amd import lightning as pl
    from predict_pv_yield_amd.data.batch import BatchML
    from predict_pv_yield_amd.data.fake import FakeDataConfiguration, FakeDataset
    from predict_pv_yield_amd.models.conv3d.model import Model
    torch.manual_seed(0)
    model = Model(include_pv_yield=False, include_nwp=False, forecast_minutes=120, history_minutes=30, image_size_pixels=16,
                  number_sat_channels=11, number_of_conv3d_layers=4, fc1_output_features=16, output_variable="gsp_yield")
    cfg = FakeDataConfiguration(batch_size=4, history_minutes=30, forecast_minutes=120, satellite_image_size_pixels=16)
    ds = FakeDataset(cfg, length=3)
    loader = torch.utils.data.DataLoader(ds, batch_size=None)
    model.results_file_name = f"{tmp_path}/temp"
    trainer = pl.Trainer(gpus=1, max_epochs=1)
    out = trainer.validate(model, loader)[0]
    results_df = pd.read_csv(f"{model.results_file_name}_0.csv")
    assert len(results_df) == len(loader) * cfg.batch_size * model.forecast_len_30 == 48
    assert list(results_df.keys()) == ["t0_datetime_utc", "target_datetime_utc", "gsp_id", "actual_gsp_pv_outturn_mw",
                                       "forecast_gsp_pv_outturn_mw"]
    want_rows, horizon = [], {}
    with torch.no_grad():
        for item in ds:
            b = BatchML(**item)
            y_hat = model(b.to(device)).cpu()
            want_rows += co.validation_results_rows(y_hat.numpy(), b.gsp.gsp_yield.numpy(), b.gsp.gsp_capacity.numpy(),
                                                    b.gsp.gsp_id.numpy(), b.metadata.t0_datetime_utc.numpy(), 4)
            for k, v in co.logged_horizon_metrics(y_hat, b.gsp.gsp_yield[:, -4:, 0], 4, "Validation").items():
                horizon.setdefault(k, []).append(v)
    assert np.allclose(results_df["forecast_gsp_pv_outturn_mw"], [r[4] for r in want_rows], rtol=1e-5, atol=1e-6)
    assert np.allclose(results_df["actual_gsp_pv_outturn_mw"], [r[3] for r in want_rows], rtol=1e-6)
    assert list(results_df["gsp_id"]) == [r[2] for r in want_rows]
    for k, vs in horizon.items():                      # epoch value = mean of the per-batch values
        assert out[f"{k}_epoch"] == pytest.approx(np.mean(vs), rel=1e-5), k
    assert "MSE_forecast_horizon_4/Validation_epoch" not in out


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_hip_graph_train_step_matches_eager(device, precision):
    """graphs.GraphedTrainStep (forward + NMAE + backward + HipAdam(capturable=True) in ONE captured HIP graph, replayed per
    batch) against the same steps taken eagerly with the same device-side Adam scalars: identical losses and parameters,
    bit for bit, over replays with different batches; and against the default (host-scalar) HipAdam within float rounding.
    fp32: the half-float conv form's scales and maxima are device-side values, its operand images ride on tensors whose
    addresses a graph keeps -- the captured step replays them like any other launch."""
    import copy
    from predict_pv_yield_amd.graphs import GraphedTrainStep
    from predict_pv_yield_amd.models.conv3d.model import Model
    from predict_pv_yield_amd.optim import HipAdam

    kw = dict(include_pv_yield=False, include_nwp=False, forecast_minutes=30, history_minutes=55, number_of_conv3d_layers=4,
              conv3d_channels=32, image_size_pixels=64, number_sat_channels=11, fc1_output_features=128,
              fc2_output_features=128, fc3_output_features=64, output_variable="pv_yield", precision=precision)
    torch.manual_seed(3)
    base = Model(**kw).to(device)
    g = torch.Generator(device=device).manual_seed(5)
    batches = [{"satellite": {"data": torch.randn(4, 11, 18, 64, 64, generator=g, device=device)},
                "pv": {"pv_yield": torch.rand(4, 18, 128, generator=g, device=device)}} for _ in range(6)]

    def run(make_opt, graphed):
        model = copy.deepcopy(base)
        opt = make_opt(model)
        losses = []
        if graphed:
            step = GraphedTrainStep(model, opt, batches[0], warmup=3)       # 3 eager steps on batches[0]; the capture only records
            for b in batches[1:]:
                losses.append(float(step(b)))
        else:
            for b in [batches[0]] * 3 + batches[1:]:
                opt.zero_grad(set_to_none=True)
                loss = model.training_step(b, 0)
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
            losses = losses[3:]
        torch.cuda.synchronize()
        return losses, [p.detach().clone() for p in model.parameters()], opt

    cap = lambda m: HipAdam(m.parameters(), lr=5e-4, capturable=True)
    l_graph, p_graph, o_graph = run(cap, True)
    l_eager, p_eager, o_eager = run(cap, False)
    assert l_graph == l_eager, (l_graph, l_eager)
    for a, b in zip(p_graph, p_eager):
        assert torch.equal(a, b)
    assert o_graph.device_step() == o_eager.device_step() == 8
    l_host, p_host, _ = run(lambda m: HipAdam(m.parameters(), lr=5e-4), False)
    for a, b in zip(p_graph, p_host):
        torch.testing.assert_close(a, b, rtol=0, atol=2e-6)
    assert max(abs(x - y) for x, y in zip(l_graph, l_host)) < 1e-5


def test_capturable_checkpoint_carries_the_device_step(device):
    """graph replays advance only the device-side step counter.  state_dict() must store THAT count, and a
    resumed optimiser (capturable or not) must continue with its bias corrections; lr frozen in the graph may not change
    silently; a workspace the graph replays into may not be replaced."""
    import copy
    from predict_pv_yield_amd import hip_ops as K
    from predict_pv_yield_amd.graphs import GraphedTrainStep
    from predict_pv_yield_amd.models.conv3d.model import Model
    from predict_pv_yield_amd.optim import HipAdam

    kw = dict(include_pv_yield=False, include_nwp=False, forecast_minutes=30, history_minutes=55, number_of_conv3d_layers=4,
              conv3d_channels=32, image_size_pixels=64, number_sat_channels=11, fc1_output_features=128,
              fc2_output_features=128, fc3_output_features=64, output_variable="pv_yield", precision="bf16")
    torch.manual_seed(11)
    model = Model(**kw).to(device)
    g = torch.Generator(device=device).manual_seed(6)
    batches = [{"satellite": {"data": torch.randn(2, 11, 18, 64, 64, generator=g, device=device)},
                "pv": {"pv_yield": torch.rand(2, 18, 128, generator=g, device=device)}} for _ in range(7)]
    opt = HipAdam(model.parameters(), lr=5e-4, capturable=True)
    step = GraphedTrainStep(model, opt, batches[0], warmup=3)
    for b in batches[1:6]:
        step(b)
    torch.cuda.synchronize()
    assert opt.device_step() == 8
    sd = copy.deepcopy(opt.state_dict())
    assert all(float(st["step"]) == 8.0 for st in sd["state"].values())

    def resume_and_step(capturable):
        m = Model(**kw).to(device)                     # (a trained module holds non-leaf caches: rebuild, then load)
        m.load_state_dict(model.state_dict())
        o = HipAdam(m.parameters(), lr=5e-4, capturable=capturable)
        o.load_state_dict(copy.deepcopy(sd))
        o.zero_grad(set_to_none=True)
        m.training_step(batches[6], 0).backward()
        o.step()
        torch.cuda.synchronize()
        if capturable:
            assert o.device_step() == 9
        return [p.detach().clone() for p in m.parameters()]

    p_host, p_dev = resume_and_step(False), resume_and_step(True)
    step(batches[6])                                   # the original optimiser simply goes on
    torch.cuda.synchronize()
    for a, b, c in zip(p_host, p_dev, model.parameters()):
        torch.testing.assert_close(a, b, rtol=0, atol=2e-6)
        assert torch.equal(b, c.detach())              # same device-side scalars: bit for bit
    # a loaded state dict lands in the live device counter AND in the live moment tensors in place (the graph holds their
    # addresses and, for fc1, the tile layout they were captured in): nothing is replaced under the graph
    fc1 = model.fc1.weight
    addr = opt.state[fc1]["exp_avg"].data_ptr()
    opt.load_state_dict(copy.deepcopy(sd))
    assert opt.device_step() == 8
    assert opt.state[fc1]["exp_avg"].data_ptr() == addr
    idx = [i for i, q in enumerate(opt._params_in_order()) if q is fc1][0]
    m_live, v_live = opt.moments(fc1)                  # row-major, reference column order -- what the dict holds
    assert torch.equal(m_live.cpu(), sd["state"][idx]["exp_avg"].cpu()) and torch.equal(v_live.cpu(), sd["state"][idx]["exp_avg_sq"].cpu())
    # kernel arguments are frozen: changing lr must not be ignored silently
    opt.param_groups[0]["lr"] = 1e-4
    with pytest.raises(RuntimeError, match="changed after capture"):
        step(batches[1])
    opt.param_groups[0]["lr"] = 5e-4
    # ... and a workspace the graph replays into is not replaced under it
    key = next(k for k in K._workspaces if k in K._pinned_workspaces)
    with pytest.raises(RuntimeError, match="held by a captured HIP graph"):
        K._workspace(key[0], K._workspaces[key].numel() * 2 + 1, torch.device(key[1]))
    step.close()
    K._workspace(key[0], K._workspaces[key].numel() + 1, torch.device(key[1]))       # released: grows again


def test_tiled_moments_are_private_to_the_one_pass_backward(device, monkeypatch):
    """HipAdam keeps exp_avg / exp_avg_sq of fc1 tile by tile while the one-pass backward owns them.  Whatever leaves the
    optimiser is torch's row-major layout: state_dict() between steps, a resumed optimiser, a change of gradient mode --
    and every number equals the run that never tiles (optim.TILE_LARGE_MOMENTS = False), bit for bit."""
    import copy
    from predict_pv_yield_amd import optim as O
    from predict_pv_yield_amd.models.conv3d.model import Model

    kw = dict(include_pv_yield=False, include_nwp=False, forecast_minutes=30, history_minutes=55, number_of_conv3d_layers=4,
              conv3d_channels=32, image_size_pixels=64, number_sat_channels=11, fc1_output_features=128,
              fc2_output_features=128, fc3_output_features=64, output_variable="pv_yield", precision="bf16")
    torch.manual_seed(11)
    base = Model(**kw).to(device)
    g = torch.Generator(device=device).manual_seed(12)
    batches = [{"satellite": {"data": torch.randn(4, 11, 18, 64, 64, generator=g, device=device)},
                "pv": {"pv_yield": torch.rand(4, 18, 128, generator=g, device=device)}} for _ in range(5)]

    def run(tile):
        monkeypatch.setattr(O, "TILE_LARGE_MOMENTS", tile)
        model = copy.deepcopy(base)
        opt = model.configure_optimizers()
        fc1 = model.fc1.weight
        snaps = []

        def step(b):
            opt.zero_grad(set_to_none=True)
            model.training_step(b, 0).backward()
            opt.step()

        step(batches[0]); step(batches[1])
        assert opt._is_tiled(fc1) == tile
        sd = copy.deepcopy(opt.state_dict())                 # row-major, whatever the stored layout was
        assert not opt._is_tiled(fc1)
        snaps.append([t.clone() for t in opt.moments(fc1)])
        step(batches[2])                                     # the next one-pass backward tiles again
        assert opt._is_tiled(fc1) == tile
        snaps.append([t.clone() for t in opt.moments(fc1)])
        twin = copy.deepcopy(opt)                            # the layout flag travels with the state tensors
        pt = [q for grp in twin.param_groups for q in grp["params"] if q.shape == fc1.shape][0]
        assert twin._is_tiled(pt) == tile and all(torch.equal(a, b) for a, b in zip(twin.moments(pt), snaps[-1]))
        del twin, pt
        # resume in a fresh optimiser from the state saved after step 2 (and the weights of that moment are gone: only the
        # optimiser round trip is under test, so the moments are compared right after loading)
        opt2 = copy.deepcopy(base).configure_optimizers()
        opt2.load_state_dict(sd)
        p2 = [q for grp in opt2.param_groups for q in grp["params"] if q.shape == fc1.shape][0]
        assert all(torch.equal(a, b) for a, b in zip(opt2.moments(p2), snaps[0]))
        # leaving the fused mode hands the moments back row-major; the two-kernel path continues from them
        opt.set_large_grad_mode("autograd")
        assert not opt._is_tiled(fc1)
        step(batches[3])
        opt.set_large_grad_mode("fused")
        step(batches[4])
        snaps.append([t.clone() for t in opt.moments(fc1)])
        return snaps, {k: v.detach().clone() for k, v in model.state_dict().items()}, sd

    (a_m, a_w, a_sd), (b_m, b_w, b_sd) = run(True), run(False)
    for x, y in zip(a_m, b_m):
        assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1])
    for k in a_w:
        assert torch.equal(a_w[k], b_w[k]), k
    for sa, sb in zip(a_sd["state"].values(), b_sd["state"].values()):
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])


@pytest.mark.gpu
def test_trainer_advects_in_the_loader_wrapper(device):
    """Trainer(advect_on_side_stream=True) (optical_flow.AdvectingLoader in front of the model; the side stream itself was
    removed in round 4) on config 3 (raw int16 counts -> Model(future_frames="optical_flow")): the same parameters after
    the same batches as the advection inside the model's forward, bit for bit."""
    import copy
    from predict_pv_yield_amd.lightning import Trainer
    from predict_pv_yield_amd.models.conv3d.model import Model

    kw = dict(include_pv_yield=False, include_nwp=False, forecast_minutes=30, history_minutes=55, number_of_conv3d_layers=4,
              conv3d_channels=32, image_size_pixels=64, number_sat_channels=11, fc1_output_features=128,
              fc2_output_features=128, fc3_output_features=64, output_variable="pv_yield", precision="bf16",
              future_frames="optical_flow")
    torch.manual_seed(11)
    base = Model(**kw)
    g = torch.Generator().manual_seed(12)
    batches = [{"satellite": {"data": (torch.rand(2, 12, 11, 64, 64, generator=g) * 1023).to(torch.int16)},
                "pv": {"pv_yield": torch.rand(2, 18, 128, generator=g)}} for _ in range(4)]
    finals = []
    for side in (False, True):
        model = copy.deepcopy(base)
        trainer = Trainer(gpus=1, max_epochs=1, advect_on_side_stream=side)
        trainer.fit(model, train_dataloaders=batches)
        torch.cuda.synchronize()
        finals.append([p.detach().clone() for p in model.parameters()])
        assert trainer.global_step == 4
    for a, b in zip(*finals):
        assert torch.equal(a, b)


SHIPPED_EXPERIMENTS = ["baseline", "conv3d", "conv3d_nwp", "conv3d_optical_flow", "conv3d_sat_nwp", "example_simple",
                       "exp001_plumbing", "exp003_perceiver", "perceiver", "perceiver_conv3d_sat_nwp", "perceiver_sat_nwp"]


@pytest.mark.gpu
@pytest.mark.parametrize("experiment", SHIPPED_EXPERIMENTS)
def test_shipped_experiments_run_with_their_own_shapes(device, tmp_path, experiment):
    """`python run.py experiment=<name> trainer.fast_dev_run=true` exactly as a user types it, from the repository root with
    the run directory elsewhere: the model yaml and the dataset configuration the experiment points at must agree, relative
    paths must survive the change into hydra.run.dir, and a fast_dev_run stays on the GPU (utils.extras)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "run.py", f"experiment={experiment}", "trainer.fast_dev_run=true",
                        f"hydra.run.dir={tmp_path}"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "no configuration.yaml under data_path" not in r.stderr + r.stdout
