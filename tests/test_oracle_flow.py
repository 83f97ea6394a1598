"""CPU: known-answer tests that pin the C restatement of Farnebäck / cv.remap / u8 conversion / weighted mean
(oracle/pv_oracle.c).  OpenCV is absent and the reference has no tests for this path, so parity with cv2 is
UNPINNED; these are the analytic KATs of SURVEY.md §8c (K1-K5, R1-R5) plus independent cross-checks against
NumPy / SciPy."""
import os

import numpy as np
import pytest
from scipy import ndimage

from oracle import flow_oracle as fo
from predict_pv_yield_amd.data.synthetic import advected_counts, blob_texture_sequence


def _u8_pair(rng, h, w, v):
    seq = blob_texture_sequence(rng, 2, h, w, v)
    return fo.convert_10bpp_to_uint8(np.clip(np.rint(seq), 0, 1023).astype(np.int16), 0)[0]


# ---- u8 conversion / weighted mean / normalise ----------------------------------------------------------
def test_u8_round_half_even_matches_numpy():
    x = np.arange(0, 1022, dtype=np.int16)              # 1022/4 rounds to 256: the reference asserts there
    got, flag = fo.convert_10bpp_to_uint8(x, 0)
    ref = (x / 4.0).round().astype(np.uint8)           # notebooks/13_...ipynb:112-119
    assert np.array_equal(got, ref) and not flag
    assert list(got[[2, 6, 10, 14]]) == [0, 2, 2, 4]   # ties go to even
    gotf, _ = fo.convert_10bpp_to_uint8(x.astype(np.float32), 0)
    assert np.array_equal(gotf, ref)


def test_u8_truncating_variant_matches_numpy():
    x = np.arange(0, 1024, dtype=np.float32)
    a = x.copy(); a -= 0; a /= 1023; a *= 255          # notebooks/optical_flow_1.ipynb:129-134
    got, _ = fo.convert_10bpp_to_uint8(x, 1)
    assert np.array_equal(got, a.astype(np.uint8))


def test_u8_range_flag():
    _, flag = fo.convert_10bpp_to_uint8(np.array([0, 1030], np.int16), 0)
    assert flag


def test_weighted_average_matches_numpy():
    rng = np.random.default_rng(0)
    flows = rng.normal(0, 2, (6, 9, 7, 2)).astype(np.float32)
    ref = np.average(flows, axis=0, weights=range(1, 7)).astype(np.float32)   # optical_flow_1.ipynb:293-294
    assert np.array_equal(fo.weighted_average(flows), ref)


def test_normalise_matches_numpy_inplace_ops():
    raw = np.arange(0, 1024, 7, dtype=np.float32)
    a = raw.copy(); a -= np.float32(93.23458); a /= np.float32(115.34247)     # 13_...ipynb:463-464
    got = fo.normalise(raw, np.array([93.23458], np.float32), np.array([115.34247], np.float32), inner=raw.size)
    assert np.array_equal(got, a)


# ---- remap (R1-R5) ------------------------------------------------------------------------------------------
def test_R1_integer_flow_is_exact_shift():
    rng = np.random.default_rng(1)
    img = rng.normal(size=(12, 15)).astype(np.float32)
    flow = np.zeros((12, 15, 2), np.float32); flow[..., 0] = 2; flow[..., 1] = -1
    out = fo.remap_image(img, flow, 1.0, fo.BORDER_REPLICATE)
    yy, xx = np.mgrid[0:12, 0:15]
    assert np.array_equal(out, img[np.clip(yy + 1, 0, 11), np.clip(xx - 2, 0, 14)])
    out_nan = fo.remap_image(img, flow, 1.0, fo.BORDER_CONSTANT, np.nan)
    # R4: every 2x2 footprint that touches the outside is NaN, even at zero weight (0 * NaN): the two left
    # columns, the last row (source row 12) and the row before it (source row 11 = h-1, second tap outside)
    assert np.isnan(out_nan[:, :2]).all() and np.isnan(out_nan[-2:]).all()
    assert np.array_equal(out_nan[:-2, 2:], out[:-2, 2:])


def test_R4_zero_flow_nan_border_contaminates_last_row_and_column():
    img = np.ones((6, 7), np.float32)
    out = fo.remap_image(img, np.zeros((6, 7, 2), np.float32), 1.0, fo.BORDER_CONSTANT, np.nan)
    assert np.isnan(out[-1]).all() and np.isnan(out[:, -1]).all() and np.all(out[:-1, :-1] == 1)


def test_R2_fractions_of_one_32nd_match_scipy():
    rng = np.random.default_rng(2)
    img = rng.normal(size=(20, 24)).astype(np.float32)
    flow = (rng.integers(-64, 64, (20, 24, 2)) / 32.0).astype(np.float32)
    out, idx = fo.remap_image(img, flow, 1.0, fo.BORDER_REPLICATE, return_indices=True)
    yy, xx = np.mgrid[0:20, 0:24].astype(np.float64)
    ref = ndimage.map_coordinates(img.astype(np.float64), [yy - flow[..., 1], xx - flow[..., 0]], order=1, mode="nearest")
    np.testing.assert_allclose(out, ref, rtol=0, atol=2e-6)
    # integer index contract: sx = round(32 * map) -> (sx >> 5, sx & 31)
    sx = np.rint((xx - flow[..., 0]) * 32).astype(np.int64)
    assert np.array_equal(idx[..., 0], sx >> 5) and np.array_equal(idx[..., 2], sx & 31)


def test_R3_cvround_ties_to_even_and_nan_coordinates():
    img = np.arange(16, dtype=np.float32).reshape(4, 4)
    flow = np.zeros((4, 4, 2), np.float32)
    flow[0, 1, 0] = -0.5 / 32      # map.x*32 = 32.5 -> 32 (even)
    flow[0, 2, 0] = -1.5 / 32      # map.x*32 = 65.5 -> 66 (even)
    flow[1, 1] = [np.nan, 0]       # NaN -> INT_MIN -> far outside
    _, idx = fo.remap_image(img, flow, 1.0, fo.BORDER_REPLICATE, return_indices=True)
    assert tuple(idx[0, 1, [0, 2]]) == (1, 0) and tuple(idx[0, 2, [0, 2]]) == (2, 2)
    assert idx[1, 1, 0] == -32768
    out = fo.remap_image(img, flow, 1.0, fo.BORDER_CONSTANT, np.nan)
    assert np.isnan(out[1, 1])


def test_R5_u8_fixed_point():
    img = np.array([[0, 100], [200, 255]], np.uint8)
    flow = np.zeros((2, 2, 2), np.float32)
    assert np.array_equal(fo.remap_image(img, flow, 1.0, fo.BORDER_REPLICATE), img)    # identity despite the 32767 weight
    flow[0, 0] = [-0.5, -0.5]      # samples the centre: (0+100+200+255)/4 = 138.75 -> 139
    out = fo.remap_image(img, flow, 1.0, fo.BORDER_REPLICATE)
    assert out[0, 0] == (0 * 8192 + 100 * 8192 + 200 * 8192 + 255 * 8192 + 16384) >> 15 == 139


def test_remap_scale_k_is_f32_product():
    rng = np.random.default_rng(3)
    img = rng.normal(size=(8, 8)).astype(np.float32)
    flow = rng.normal(0, 1, (8, 8, 2)).astype(np.float32)
    a = fo.remap_image(img, flow, 3.0, fo.BORDER_REPLICATE)
    b = fo.remap_image(img, (flow * np.float32(3.0)).astype(np.float32), 1.0, fo.BORDER_REPLICATE)   # 13_...ipynb:320
    assert np.array_equal(a, b)


# ---- Farnebäck (K1-K5) -----------------------------------------------------------------------------------------
def test_K3_level_rule():
    assert fo.farneback_num_levels(64, 64) == 1          # 32x32 is the only coarse level
    assert fo.farneback_num_levels(704, 548) == 2        # 176 x 137 >= 32
    assert fo.farneback_num_levels(63, 64) == 0
    assert fo.farneback_num_levels(128, 128, levels=5) == 2


def test_K4_gaussian_tables_closed_form():
    g, xg, xxg, ig = fo.farneback_poly_tables(5, 0.7)
    x = np.arange(-5, 6, dtype=np.float64)
    gd = np.exp(-x * x / (2 * 0.7 ** 2)); gd /= gd.sum()
    np.testing.assert_allclose(g, gd, rtol=2e-7, atol=1e-12)
    np.testing.assert_allclose(xg, x * gd, rtol=2e-7, atol=1e-12)
    np.testing.assert_allclose(xxg, x * x * gd, rtol=2e-7, atol=1e-12)
    g64 = g.astype(np.float64)
    G = np.zeros((6, 6))
    s2 = (np.outer(g64, g64) * (x ** 2)[None]).sum(); s4 = (np.outer(g64, g64) * (x ** 4)[None]).sum()
    s22 = (np.outer(g64 * x * x, g64 * x * x)).sum()
    G[0, 0] = np.outer(g64, g64).sum(); G[1, 1] = G[2, 2] = G[0, 3] = G[0, 4] = G[3, 0] = G[4, 0] = s2
    G[3, 3] = G[4, 4] = s4; G[3, 4] = G[4, 3] = G[5, 5] = s22
    inv = np.linalg.inv(G)
    np.testing.assert_allclose(ig, [inv[1, 1], inv[0, 3], inv[3, 3], inv[5, 5]], rtol=1e-6)
    k = fo.farneback_window_taps(40)
    t = np.exp(-np.arange(21.0) ** 2 / (2 * 6.0 ** 2))
    np.testing.assert_allclose(k, t / (t[0] + 2 * t[1:].sum()), rtol=3e-7)
    assert abs(k[0] + 2 * k[1:].sum() - 1) < 1e-6 and len(k) == 21


def test_K1_identical_frames():
    rng = np.random.default_rng(4)
    u8 = _u8_pair(rng, 64, 64, (0, 0))
    flow = fo.calc_optical_flow_farneback(u8[0], u8[0])
    assert np.abs(flow[8:-8, 8:-8]).max() <= 1e-3 or np.abs(flow).max() < 0.2
    assert np.abs(flow).max() < 0.2                    # border branch of UpdateMatrices leaks at most ~0.1 px


@pytest.mark.parametrize("v", [(1.0, 0.0), (0.0, -2.0), (1.5, 0.75), (3.0, -2.0)])
@pytest.mark.parametrize("hw", [(64, 64), (160, 200)])
def test_K2_translation_recovered(v, hw):
    rng = np.random.default_rng(5)
    u8 = _u8_pair(rng, hw[0], hw[1], v)
    flow = fo.calc_optical_flow_farneback(u8[0], u8[1])
    inner = flow[12:-12, 12:-12]
    err = np.hypot(inner[..., 0] - v[0], inner[..., 1] - v[1])
    assert np.percentile(err, 95) <= 0.15, np.percentile(err, 95)


def test_K5_border_attenuation_limits_edge_flow():
    rng = np.random.default_rng(6)
    u8 = _u8_pair(rng, 64, 64, (2.0, 0.0))
    flow = fo.calc_optical_flow_farneback(u8[0], u8[1], winsize=5, iterations=1, levels=0)
    # with a tiny window the 5-px frame (scale .14/.4472) visibly damps the solution next to the border
    assert np.abs(flow[32, 0, 0]) < np.abs(np.median(flow[20:44, 20:44, 0]))


def test_polyexp_constant_image_has_zero_derivatives():
    img = np.full((40, 40), 77, np.uint8)
    I, R = fo.farneback_level_polyexp(img, 0)
    assert np.all(I == 77)
    assert np.abs(R).max() < 1e-3                      # first and second derivatives of a constant


def test_polyexp_linear_ramp_gives_unit_gradient():
    img = np.tile(np.arange(60, dtype=np.uint8)[None] * 2, (60, 1))   # I = 2x
    _, R = fo.farneback_level_polyexp(img, 0)
    c = R[20:40, 20:40]
    np.testing.assert_allclose(c[..., 1], 2.0, atol=1e-3)   # r_x
    np.testing.assert_allclose(c[..., 0], 0.0, atol=1e-3)   # r_y
    assert np.abs(c[..., 2:]).max() < 1e-3


def test_polyexp_quadratic_image_closed_form():
    """A polynomial of degree 2 is reproduced exactly by the weighted least-squares fit of PolyExp (basis 1, x, y, x^2,
    y^2, xy under the separable Gaussian weight): at pixel (x0, y0) the expansion of
        I = a + bx x + by y + cxx x^2 + cyy y^2 + cxy x y
    is r_x = bx + 2 cxx x0 + cxy y0, r_y = by + 2 cyy y0 + cxy x0, r_xx = cxx, r_yy = cyy, r_xy = cxy (interior pixels:
    the replicated border is not a polynomial)."""
    h, w = 36, 44
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    a, bx, by, cxx, cyy, cxy = 7.0, 3.0, -2.0, 0.5, 0.25, 0.125          # every value of I is exact in f32
    img = (a + bx * xx + by * yy + cxx * xx * xx + cyy * yy * yy + cxy * xx * yy).astype(np.float32)
    R = fo.poly_exp(img)
    inner = (slice(5, h - 5), slice(5, w - 5))
    np.testing.assert_allclose(R[..., 1][inner], (bx + 2 * cxx * xx + cxy * yy)[inner], rtol=2e-5, atol=2e-4)   # r_x
    np.testing.assert_allclose(R[..., 0][inner], (by + 2 * cyy * yy + cxy * xx)[inner], rtol=2e-5, atol=2e-4)   # r_y
    np.testing.assert_allclose(R[..., 3][inner], cxx, rtol=0, atol=2e-4)                                        # r_xx
    np.testing.assert_allclose(R[..., 2][inner], cyy, rtol=0, atol=2e-4)                                        # r_yy
    np.testing.assert_allclose(R[..., 4][inner], cxy, rtol=0, atol=2e-4)                                        # r_xy


def test_window_blur_rows_and_columns_match_scipy_correlate1d():
    """The 41-tap Gaussian window (winsize 40: m = 20, sigma = 6) is a separable, border-REPLICATED correlation:
    scipy.ndimage.correlate1d(mode="nearest") with the symmetric tap vector, vertical then horizontal, per channel."""
    rng = np.random.default_rng(11)
    M = rng.normal(0, 1, (50, 70, 5)).astype(np.float32)
    blurred, flow = fo.window_blur_solve(M, 40)
    k = fo.farneback_window_taps(40).astype(np.float64)
    taps = np.concatenate([k[:0:-1], k])                                  # k20 .. k1 k0 k1 .. k20
    ref = ndimage.correlate1d(ndimage.correlate1d(M.astype(np.float64), taps, axis=0, mode="nearest"), taps, axis=1,
                              mode="nearest")
    np.testing.assert_allclose(blurred, ref, rtol=0, atol=5e-6)
    # a narrow window exercises the replicated border on every pixel of a small image
    b3, _ = fo.window_blur_solve(M[:7, :9], 6)
    k3 = fo.farneback_window_taps(6).astype(np.float64)
    t3 = np.concatenate([k3[:0:-1], k3])
    ref3 = ndimage.correlate1d(ndimage.correlate1d(M[:7, :9].astype(np.float64), t3, axis=0, mode="nearest"), t3, axis=1,
                               mode="nearest")
    np.testing.assert_allclose(b3, ref3, rtol=0, atol=5e-6)
    # 2x2 solve on the blurred matrices: flow = G^-1 h with the +1e-3 regulariser on the determinant
    g11, g12, g22, h1, h2 = (blurred[..., i].astype(np.float64) for i in range(5))
    idet = 1.0 / (g11 * g22 - g12 * g12 + 1e-3)
    np.testing.assert_allclose(flow[..., 0], (g11 * h2 - g12 * h1) * idet, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(flow[..., 1], (g22 * h1 - g12 * h2) * idet, rtol=1e-6, atol=1e-6)


def test_update_matrices_closed_form_for_a_pure_translation_of_a_quadratic():
    """For R1 = the expansion of the same quadratic image shifted by d and flow = d, UpdateMatrices' sampled R1
    coefficients equal R0's second-order terms exactly and r2/r3 reduce to A d: the solve then returns d itself.
    Checked on interior pixels with a constant (hence trivially blurred) coefficient field."""
    h, w = 40, 40
    cxx, cyy, cxy = 0.5, 0.25, 0.125
    d = np.array([1.5, -0.75])                       # (dx, dy)
    # PolyExp fields of I(p) and of I2(p) = I(p - d): same A, linear term shifted by -2 A d
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    def field(ox, oy):
        rx = 3.0 + 2 * cxx * (xx - ox) + cxy * (yy - oy)
        ry = -2.0 + 2 * cyy * (yy - oy) + cxy * (xx - ox)
        return np.stack([ry, rx, np.full_like(xx, cyy), np.full_like(xx, cxx), np.full_like(xx, cxy)], -1).astype(np.float32)
    R0, R1 = field(0.0, 0.0), field(d[0], d[1])
    flow = np.broadcast_to(d.astype(np.float32), (h, w, 2)).copy()
    M = fo.update_matrices(R0, R1, flow)
    inner = (slice(8, h - 8), slice(8, w - 8))
    g11, g12, g22, h1, h2 = (M[..., i][inner].astype(np.float64) for i in range(5))
    # A = [[r_yy, r_xy/2], [r_xy/2, r_xx]] in (y, x) order: G = A^T A, h = A^T (A d) with d = (dy, dx)
    A = np.array([[cyy, cxy / 2], [cxy / 2, cxx]])
    G = A.T @ A
    hv = A.T @ (A @ np.array([d[1], d[0]]))
    np.testing.assert_allclose(g11, G[0, 0], rtol=1e-5)
    np.testing.assert_allclose(g12, G[0, 1], rtol=1e-5)
    np.testing.assert_allclose(g22, G[1, 1], rtol=1e-5)
    np.testing.assert_allclose(h1, hv[0], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(h2, hv[1], rtol=1e-4, atol=1e-6)


def test_advect_frames_pipeline_shapes_and_known_motion():
    raw, vel = advected_counts(batch=1, t=5, channels=1, h=64, w=64, seed=9, vmax=1.5)
    mean, std = np.array([93.23458], np.float32), np.array([115.34247], np.float32)
    out = fo.advect_frames(raw, mean, std, n_future=2)
    assert out.shape == (1, 1, 7, 64, 64) and np.isfinite(out).all()
    # the advected frame t0+1 should look like the texture moved one more step: compare with persistence
    nxt = blob_texture_sequence(np.random.default_rng(9), 1, 64, 64, (0, 0))  # only a smoke check on statistics
    assert out[0, 0, 5, 16:-16, 16:-16].std() > 0.1 and nxt.shape == (1, 64, 64)


# ---- two independent restatements of cv.calcOpticalFlowFarneback (what can be pinned without cv2) -------
@pytest.mark.parametrize("h,w", [(64, 64), (32, 48), (96, 80), (160, 200), (65, 67), (20, 200), (131, 70)])
def test_two_independent_statements_agree(h, w):
    """oracle/pv_oracle.c (OpenCV's loop structure and float accumulators, transcribed) against oracle/farneback_f64.py
    (float64 NumPy written from SURVEY.md Appendix A.1): <= 1e-4 px on the sizes the GPU parity test uses -- one- and
    two-level pyramids, odd sizes, a level whose height is 20 px.  Measured 1e-6 .. 3e-6 px."""
    from oracle import farneback_f64 as f64
    from predict_pv_yield_amd.data.synthetic import blob_texture_sequence
    rng = np.random.default_rng(h + w)
    for v in [(1.5, 0.75), (-2.25, 1.3), (0.4, -2.6)]:
        seq = blob_texture_sequence(rng, 2, h, w, v)
        u8, _ = fo.convert_10bpp_to_uint8(np.clip(np.rint(seq), 0, 1023).astype(np.int16), 0)
        a = fo.calc_optical_flow_farneback(u8[0], u8[1])
        b = f64.calc_optical_flow_farneback(u8[0], u8[1])
        assert f64.num_levels(h, w, 0.5, 2) == fo.farneback_num_levels(h, w)
        err = np.abs(a - b).max()
        assert err <= 1e-4, (h, w, v, err)


def test_two_statements_on_a_degenerate_motion():
    """A flow component that is exactly zero (velocity (1, 0)) puts fy = y + dy on an integer: the sign of a 1e-9 decides
    whether row 0 samples inside the image (Appendix A.1 item 4's branch), and the 41-tap window spreads that choice -- the two
    statements (float against double sums) then differ by 1e-3 .. 2e-2 px on a few per cent of the pixels.  Bounded here so
    that the sensitivity is on record: it is a property of the algorithm at such inputs, cv2's included."""
    from oracle import farneback_f64 as f64
    from predict_pv_yield_amd.data.synthetic import blob_texture_sequence
    for h, w in [(64, 64), (160, 200)]:
        rng = np.random.default_rng(h + w)
        seq = blob_texture_sequence(rng, 2, h, w, (1.0, 0.0))
        u8, _ = fo.convert_10bpp_to_uint8(np.clip(np.rint(seq), 0, 1023).astype(np.int16), 0)
        d = np.abs(fo.calc_optical_flow_farneback(u8[0], u8[1]) - f64.calc_optical_flow_farneback(u8[0], u8[1])).max(axis=-1)
        assert d.max() <= 5e-2 and np.median(d) <= 1e-5 and (d > 1e-4).mean() <= 0.2, (h, w, d.max(), np.median(d))


def test_f64_statement_pieces_against_the_c_oracle():
    """Stage by stage, so that a disagreement of the whole can be located: PolyExp, UpdateMatrices, window blur + solve."""
    from oracle import farneback_f64 as f64
    rng = np.random.default_rng(5)
    img = rng.uniform(0, 255, (48, 56)).astype(np.float32)
    R_c, R_n = fo.poly_exp(img), f64.poly_exp(img, 5, 0.7)
    assert np.abs(R_c - R_n).max() <= 2e-4 * np.abs(R_n).max()
    img2 = np.roll(img, (1, 2), axis=(0, 1))
    R1_c = fo.poly_exp(img2)
    flow = rng.uniform(-3, 3, (48, 56, 2)).astype(np.float32)
    M_c = fo.update_matrices(R_c, R1_c, flow)
    M_n = f64.update_matrices(R_c.astype(np.float64), R1_c.astype(np.float64), flow.astype(np.float64))
    assert np.abs(M_c - M_n).max() <= 1e-5 * np.abs(M_n).max()
    _, fl_c = fo.window_blur_solve(M_c, 40)
    fl_n = f64.blur_and_solve(M_c.astype(np.float64), 40)
    assert np.abs(fl_c - fl_n).max() <= 1e-4
    assert np.allclose(f64.window_taps(40)[20:], fo.farneback_window_taps(40), rtol=0, atol=0)
    g, xg, xxg, ig = fo.farneback_poly_tables(5, 0.7)
    g2, xg2, xxg2, ig2 = f64.poly_tables(5, 0.7)
    assert np.array_equal(g, g2.astype(np.float32)) and np.array_equal(xg, xg2.astype(np.float32))
    assert np.allclose(ig, ig2, rtol=1e-8)      # (two 6 x 6 inversions in double: LU here, the C statement's own elimination)


# ---- structural similarity of the forecasts: the reference's quality number (optical_flow_1.ipynb cells 31, 35, 38) ----------
SSIM_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ssim_skimage.npz")


def test_ssim_restatement_reproduces_skimage():
    """oracle/ssim.py against scikit-image 0.18.3's own scores (tests/golden/make_ssim_golden.py)."""
    from oracle.ssim import structural_similarity
    d = np.load(SSIM_GOLDEN)
    frames, forecasts, w0 = d["frames"], d["forecasts"], int(d["warm_up"])
    for i in range(forecasts.shape[0]):
        assert abs(structural_similarity(frames[w0 - 1 + i], forecasts[i]) - d["ssim_flow"][i]) <= 1e-12
        assert abs(structural_similarity(frames[w0 - 1], frames[w0 - 1 + i]) - d["ssim_persistence"][i]) <= 1e-12


def test_oracle_forecasts_score_as_in_the_golden_and_beat_persistence():
    """The oracle's pipeline (u8 conversion, Farneback per warm-up pair, weighted average, remap of image_t0 by flow * i)
    re-run here reproduces the fixture's forecasts bit for bit, and the scores behave as the notebook's plot shows: the
    advected forecast stays near 1 while persistence decays."""
    d = np.load(SSIM_GOLDEN)
    frames, w0 = d["frames"], int(d["warm_up"])
    flows = np.stack([fo.calc_optical_flow_farneback(frames[i], frames[i + 1]) for i in range(w0 - 1)])
    flow = fo.weighted_average(flows)
    assert np.array_equal(flow, d["flow"])
    for i in range(d["forecasts"].shape[0]):
        assert np.array_equal(fo.remap_image(frames[w0 - 1], flow, float(i), fo.BORDER_REPLICATE, 0), d["forecasts"][i])
    assert (d["ssim_flow"][1:] > 0.97).all() and (d["ssim_flow"][1:] > d["ssim_persistence"][1:] + 0.05).all()
    assert d["ssim_persistence"][-1] < 0.3
