"""nb-13 super-batch -> example pipeline (predict_pv_yield_amd/data/flow_examples.py): host logic on CPU against the
NumPy restatement of the notebook's super_batch_to_example / sample_squares; device pipeline on the GPU against the
C/NumPy oracle."""
import numpy as np
import pytest
import torch

from oracle import flow_oracle as fo
from predict_pv_yield_amd.data import flow_examples as fe
from predict_pv_yield_amd.data.synthetic import blob_texture_sequence
from predict_pv_yield_amd.models.conv3d import flow_autoencoder as fa


def _fake_super_batch(t=20, h=150, w=170, seed=0, nan_border=9):
    """Normalised images + (n-1)n/2 'predictions' with a NaN frame that grows with the forecast step."""
    rng = np.random.default_rng(seed)
    sat = rng.standard_normal((t, h, w)).astype(np.float32)
    preds, index = [], []
    for flow_i in range(t - 1):
        for step in range(1, t - flow_i):
            p = rng.standard_normal((h, w)).astype(np.float32)
            m = min(nan_border + step, 40)
            p[:m] = np.nan
            p[:, -m:] = np.nan
            preds.append(p)
            index.append((flow_i, flow_i + step))
    return sat, np.stack(preds), np.array(index, dtype=np.int64)


def test_host_sampling_matches_notebook_restatement():
    sat, preds, index = _fake_super_batch()
    sb = {fe.SAT_IMAGES: torch.from_numpy(sat), fe.OPTICAL_FLOW_PREDICTIONS: torch.from_numpy(preds),
          fe.PREDICTION_INDEX: torch.from_numpy(index)}
    rng_a, rng_b = np.random.default_rng(42), np.random.default_rng(42)
    for _ in range(25):
        ex = fe.super_batch_to_example(sb, rng=rng_a, n_pixels_per_side_large=64, n_pixels_per_side_small=32)
        h_c, p_c, horizon_s, t_c = fo.super_batch_to_example_np(sat, preds, index, rng_b, large=64, small=32)
        assert np.array_equal(ex[fa.HISTORICAL_SAT_IMAGES].numpy(), h_c)
        assert np.array_equal(ex[fa.OPTICAL_FLOW_PREDICTIONS].numpy(), p_c)
        assert np.array_equal(ex[fa.TARGET_SAT_IMAGE].numpy(), t_c)
        assert float(ex[fa.FORECAST_HORIZON]) == float(fa.normalise_forecast_horizon(horizon_s))
        assert ex[fa.HISTORICAL_SAT_IMAGES].shape == (4, 64, 64) and ex[fa.TARGET_SAT_IMAGE].shape == (32, 32)
        assert not torch.isnan(ex[fa.OPTICAL_FLOW_PREDICTIONS]).any()
    assert rng_a.integers(0, 1 << 30) == rng_b.integers(0, 1 << 30)        # same number of draws consumed
    batch = fe.collate([fe.super_batch_to_example(sb, rng=rng_a, n_pixels_per_side_large=64, n_pixels_per_side_small=32)
                        for _ in range(3)])
    assert batch[fa.HISTORICAL_SAT_IMAGES].shape == (3, 4, 64, 64) and batch[fa.FORECAST_HORIZON].shape == (3,)


def test_sampler_matches_the_notebook_cells_themselves():
    """tests/golden/flow_sampler.npz was produced by exec'ing the notebook's sample_squares / normalise_forecast_horizon /
    super_batch_to_example cells (13_...ipynb:604-728; make_flow_sampler_golden.py) on this very super batch with rng 42:
    the NumPy restatement (the oracle of the GPU tests) and the product's host sampler reproduce every draw -- crops,
    horizons and the position of the random stream."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "flow_sampler.npz"))
    n_draws, seed, rng_seed = (int(v) for v in g["params"])
    sat, preds, index = _fake_super_batch(seed=seed)
    sb = {fe.SAT_IMAGES: torch.from_numpy(sat), fe.OPTICAL_FLOW_PREDICTIONS: torch.from_numpy(preds),
          fe.PREDICTION_INDEX: torch.from_numpy(index)}
    rng_a, rng_b = np.random.default_rng(rng_seed), np.random.default_rng(rng_seed)
    for i in range(n_draws):
        h_c, p_c, horizon_s, t_c = fo.super_batch_to_example_np(sat, preds, index, rng_b)           # 128 / 64 px defaults
        ex = fe.super_batch_to_example(sb, rng=rng_a)
        for name, ours, prod in (("hist", h_c, ex[fa.HISTORICAL_SAT_IMAGES]), ("pred", p_c, ex[fa.OPTICAL_FLOW_PREDICTIONS]),
                                 ("target", t_c, ex[fa.TARGET_SAT_IMAGE])):
            assert ours.astype(np.float64).sum() == g[f"{name}_sum"][i], (i, name)
            assert np.float32(ours.reshape(-1)[0]) == g[f"{name}_corner"][i], (i, name)
            assert np.array_equal(prod.numpy(), ours), (i, name)
        assert np.float32(fa.normalise_forecast_horizon(horizon_s)) == g["horizon"][i]
        assert float(ex[fa.FORECAST_HORIZON]) == float(g["horizon"][i])
    assert rng_a.integers(0, 1 << 30) == rng_b.integers(0, 1 << 30) == int(g["rng_probe"])


def test_pick_example_indices_ranges():
    _, _, index = _fake_super_batch(t=14, h=8, w=8)
    rng = np.random.default_rng(1)
    seen = set()
    for _ in range(200):
        start, end, t0, row, target = fe.pick_example_indices(rng, 14, index)
        assert end - start == 12 and t0 == end - 1 and index[row, 0] == t0 and t0 < target <= 13
        seen.add(start)
    assert seen == {0}                     # 14 images: max_hist_start_idx = 1 -> the only legal start is 0


def test_all_nan_predictions_raise():
    sat, preds, index = _fake_super_batch(t=14, h=70, w=70)
    preds[:] = np.nan
    sb = {fe.SAT_IMAGES: torch.from_numpy(sat), fe.OPTICAL_FLOW_PREDICTIONS: torch.from_numpy(preds),
          fe.PREDICTION_INDEX: index}
    with pytest.raises(fe.ImageHasNansError):
        fe.super_batch_to_example(sb, rng=np.random.default_rng(0), max_retries=3, n_pixels_per_side_large=64,
                                  n_pixels_per_side_small=32)


def test_horizon_normalisation_constants():
    # 13_…ipynb:655-668: mean / std of arange(1, 24) * 300 s
    assert abs(float(fa.FCST_HORIZON_MEAN) - 3600.0) < 1e-3
    assert abs(float(fa.normalise_forecast_horizon(3600.0))) < 1e-6
    assert abs(float(fa.normalise_forecast_horizon(300.0)) - (300.0 - 3600.0) / float(np.arange(1, 24).std() * 300)) < 1e-5


def test_load_super_batch_on_cpu_fails_loudly():
    with pytest.raises(RuntimeError, match="MI355X"):
        fe.load_super_batch(torch.zeros(4, 64, 64, dtype=torch.int16))


@pytest.mark.gpu
def test_super_batch_pipeline_vs_oracle(device):
    """raw counts -> Farnebäck per pair -> normalise -> all predictions, then examples: device vs oracle."""
    rng = np.random.default_rng(5)
    seq = blob_texture_sequence(rng, 15, 150, 170, (1.3, -0.8))
    raw = np.clip(np.rint(seq), 0, 1023).astype(np.int16)
    sb = fe.load_super_batch(torch.from_numpy(raw).to(device))
    flows = sb[fe.OPTICAL_FLOW_FIELDS].cpu().numpy()
    assert flows.shape == (14, 150, 170, 2)
    u8, _ = fo.convert_10bpp_to_uint8(raw)
    for i in (0, 7, 13):
        ref = fo.calc_optical_flow_farneback(u8[i], u8[i + 1])
        assert np.abs(flows[i] - ref).max() < 1e-3
    sat_ref = fo.normalise(raw, np.array([fe.SAT_IMAGE_MEAN]), np.array([fe.SAT_IMAGE_STD]), inner=raw.size)
    assert np.array_equal(sb[fe.SAT_IMAGES].cpu().numpy(), sat_ref)
    # predictions from the DEVICE flows: the warp itself must be bit-exact (same fixed-point coordinates)
    preds_ref, index_ref = fo.compute_optical_flow_predictions_np(sat_ref, flows)
    assert np.array_equal(sb[fe.PREDICTION_INDEX].numpy(), index_ref)
    assert np.array_equal(sb[fe.OPTICAL_FLOW_PREDICTIONS].cpu().numpy(), preds_ref, equal_nan=True)
    # examples: same random choices, identical crops
    rng_a, rng_b = np.random.default_rng(9), np.random.default_rng(9)
    examples = [fe.super_batch_to_example(sb, rng=rng_a) for _ in range(6)]
    for ex in examples:
        h_c, p_c, horizon_s, t_c = fo.super_batch_to_example_np(sat_ref, preds_ref, index_ref, rng_b)
        assert np.array_equal(ex[fa.HISTORICAL_SAT_IMAGES].cpu().numpy(), h_c)
        assert np.array_equal(ex[fa.OPTICAL_FLOW_PREDICTIONS].cpu().numpy(), p_c)
        assert np.array_equal(ex[fa.TARGET_SAT_IMAGE].cpu().numpy(), t_c)
    batch = fe.collate(examples)
    # the flow prediction is a better guess of the target's centre than persistence of the t0 image
    centre = slice(32, 96)
    err_flow = (batch[fa.OPTICAL_FLOW_PREDICTIONS][:, centre, centre] - batch[fa.TARGET_SAT_IMAGE]).abs().mean()
    err_pers = (batch[fa.HISTORICAL_SAT_IMAGES][:, -1, centre, centre] - batch[fa.TARGET_SAT_IMAGE]).abs().mean()
    assert float(err_flow) < 0.5 * float(err_pers)


@pytest.mark.gpu
def test_autoencoder_trains_on_flow_examples(device):
    """End to end on the device: super batch -> examples -> LitAutoEncoder; the MSE falls over 30 Adam steps."""
    rng = np.random.default_rng(6)
    seq = blob_texture_sequence(rng, 15, 150, 170, (-1.1, 0.6))
    raw = torch.from_numpy(np.clip(np.rint(seq), 0, 1023).astype(np.int16)).to(device)
    sb = fe.load_super_batch(raw)
    sampler = np.random.default_rng(3)
    batch = fe.collate([fe.super_batch_to_example(sb, rng=sampler) for _ in range(8)])
    torch.manual_seed(0)
    model = fa.LitAutoEncoder().to(device)
    opt = model.configure_optimizers()
    for g in opt.param_groups:
        g["lr"] = 2e-3
    losses = []
    for _ in range(30):
        opt.zero_grad()
        loss = model.training_step(batch, 0)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert np.isfinite(losses).all() and losses[-1] < 0.8 * losses[0]
