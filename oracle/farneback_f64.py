"""A SECOND, independent statement of cv.calcOpticalFlowFarneback: float64 NumPy / SciPy, whole-array operations, written
straight from SURVEY.md Appendix A.1 (the published algorithm: G. Farneback, "Two-Frame Motion Estimation Based on
Polynomial Expansion", SCIA 2003, as OpenCV 4.5 parameterises it) -- NOT from oracle/pv_oracle.c, whose statement-by-statement
C transcription it cross-checks (tests/test_oracle_flow.py::test_two_independent_statements_agree).

TEST INFRASTRUCTURE ONLY (tests/).  The product package never imports this module.

Why a second statement: the reference calls OpenCV (notebooks/13_3d_conv_with_optical_flow_predictions.ipynb:133-135), and
OpenCV is neither vendored under /root/reference nor installable here, so parity with cv2 itself stays unpinned
(tests/golden/make_flow_golden.py produces the fixture on any machine that has it).  Two restatements written separately --
one in C with OpenCV's loop structure and float accumulators, one in vectorised float64 -- agreeing to 1e-4 px is weaker
than a cv2 fixture but stronger than one restatement checked against itself: a transcription slip in either (a swapped
channel, an off-by-one border, a wrong normalisation) shows up as a disagreement of whole pixels.

Differences from OpenCV's arithmetic that are deliberate here: every accumulation is float64 (OpenCV: float in the
vertical passes and the window blur, double in PolyExp's horizontal pass); tap TABLES are rounded to float32 exactly where
OpenCV stores them as float (Appendix A.1 items 2, 3, 5), so what differs is rounding of sums, not coefficients.
"""
import numpy as np
from scipy.ndimage import correlate1d

BORDER_ATTENUATION = np.array([0.14, 0.14, 0.4472, 0.4472, 0.4472], dtype=np.float32).astype(np.float64)


def num_levels(rows: int, cols: int, pyr_scale: float, levels: int) -> int:
    """A.1 item 1."""
    scale, k = 1.0, 0
    for k in range(levels):
        scale *= pyr_scale
        if cols * scale < 32 or rows * scale < 32:
            return k
    return levels


def _smoothing_taps(ksize: int, sigma: float) -> np.ndarray:
    """GaussianBlur's row / column kernel as OpenCV builds it for a float image (float32 taps)."""
    if sigma <= 0:
        fixed = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
                 7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}
        return np.asarray(fixed[ksize], dtype=np.float64)
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    t = np.exp(-(x * x) / (2.0 * sigma * sigma)).astype(np.float32).astype(np.float64)
    return (t * (1.0 / t.sum())).astype(np.float32).astype(np.float64)


def _resize_linear(img: np.ndarray, height: int, width: int) -> np.ndarray:
    """cv.resize(..., INTER_LINEAR) on the two leading axes: pixel centres aligned, edge-clamped."""
    def axis_plan(src_n, dst_n):
        f = (np.arange(dst_n, dtype=np.float64) + 0.5) * (src_n / dst_n) - 0.5
        i0 = np.floor(f)
        w = f - i0
        i0 = i0.astype(np.int64)
        low = i0 < 0
        i0[low], w[low] = 0, 0.0
        high = i0 >= src_n - 1
        i0[high], w[high] = src_n - 1, 0.0
        return i0, np.minimum(i0 + 1, src_n - 1), w
    y0, y1, wy = axis_plan(img.shape[0], height)
    x0, x1, wx = axis_plan(img.shape[1], width)
    extra = (1,) * (img.ndim - 2)
    wy = wy.reshape((-1, 1) + extra)
    wx = wx.reshape((1, -1) + extra)
    top = img[y0][:, x0] * (1.0 - wx) + img[y0][:, x1] * wx
    bot = img[y1][:, x0] * (1.0 - wx) + img[y1][:, x1] * wx
    return top * (1.0 - wy) + bot * wy


def poly_tables(n: int, sigma: float):
    """A.1 item 3: (g, xg, xxg) over x in [-n, n] as float32-rounded values, and (ig11, ig03, ig33, ig55)."""
    if sigma < np.finfo(np.float32).eps:
        sigma = n * 0.3
    x = np.arange(-n, n + 1, dtype=np.float64)
    g = np.exp(-(x * x) / (2.0 * sigma * sigma)).astype(np.float32).astype(np.float64)
    g = (g * (1.0 / g.sum())).astype(np.float32).astype(np.float64)
    xg = (x * g).astype(np.float32).astype(np.float64)
    xxg = (x * x * g).astype(np.float32).astype(np.float64)
    gg = np.outer(g, g)                     # g[y] g[x]
    xx = np.outer(np.ones_like(x), x * x)   # x^2
    yy = xx.T
    G = np.zeros((6, 6))
    G[0, 0] = gg.sum()
    s2 = (gg * xx).sum()
    G[1, 1] = G[2, 2] = G[0, 3] = G[0, 4] = G[3, 0] = G[4, 0] = s2
    G[3, 3] = G[4, 4] = (gg * xx * xx).sum()
    G[3, 4] = G[4, 3] = G[5, 5] = (gg * xx * yy).sum()
    inv = np.linalg.inv(G)
    return g, xg, xxg, (inv[1, 1], inv[0, 3], inv[3, 3], inv[5, 5])


def poly_exp(img: np.ndarray, n: int, sigma: float) -> np.ndarray:
    """A.1 item 3: image [h, w] -> R [h, w, 5] = (r_y, r_x, r_yy, r_xx, r_xy); replicate borders in both passes."""
    g, xg, xxg, (ig11, ig03, ig33, ig55) = poly_tables(n, sigma)
    I = np.asarray(img, dtype=np.float64)
    t0 = correlate1d(I, g, axis=0, mode="nearest")
    t1 = correlate1d(I, xg, axis=0, mode="nearest")
    t2 = correlate1d(I, xxg, axis=0, mode="nearest")
    b1 = correlate1d(t0, g, axis=1, mode="nearest")
    b2 = correlate1d(t0, xg, axis=1, mode="nearest")
    b3 = correlate1d(t1, g, axis=1, mode="nearest")
    b4 = correlate1d(t0, xxg, axis=1, mode="nearest")
    b5 = correlate1d(t2, g, axis=1, mode="nearest")
    b6 = correlate1d(t1, xg, axis=1, mode="nearest")
    return np.stack([b3 * ig11, b2 * ig11, b1 * ig03 + b5 * ig33, b1 * ig03 + b4 * ig33, b6 * ig55], axis=-1)


def update_matrices(R0: np.ndarray, R1: np.ndarray, flow: np.ndarray) -> np.ndarray:
    """A.1 item 4: -> M [h, w, 5] = (G11, G12, G22, h1, h2)."""
    h, w = flow.shape[:2]
    ys, xs = np.mgrid[0:h, 0:w]
    dx, dy = flow[..., 0], flow[..., 1]
    fx, fy = xs + dx, ys + dy
    x1, y1 = np.floor(fx), np.floor(fy)
    inside = (x1 >= 0) & (x1 < w - 1) & (y1 >= 0) & (y1 < h - 1)
    xi = np.clip(x1, 0, w - 2).astype(np.int64)
    yi = np.clip(y1, 0, h - 2).astype(np.int64)
    ax, ay = (fx - x1)[..., None], (fy - y1)[..., None]
    s = ((1 - ay) * ((1 - ax) * R1[yi, xi] + ax * R1[yi, xi + 1]) + ay * ((1 - ax) * R1[yi + 1, xi] + ax * R1[yi + 1, xi + 1]))
    r_y0, r_x0, r_yy0, r_xx0, r_xy0 = (R0[..., c] for c in range(5))
    r4 = np.where(inside, (r_yy0 + s[..., 2]) * 0.5, r_yy0)
    r5 = np.where(inside, (r_xx0 + s[..., 3]) * 0.5, r_xx0)
    r6 = np.where(inside, (r_xy0 + s[..., 4]) * 0.25, r_xy0 * 0.5)
    r2 = np.where(inside, s[..., 0], 0.0)
    r3 = np.where(inside, s[..., 1], 0.0)
    r2 = (r_y0 - r2) * 0.5
    r3 = (r_x0 - r3) * 0.5
    r2 = r2 + r4 * dy + r6 * dx
    r3 = r3 + r6 * dy + r5 * dx

    def edge(n):
        a = np.ones(n)
        for d in range(min(5, n)):
            a[d] *= BORDER_ATTENUATION[d]
            a[n - 1 - d] *= BORDER_ATTENUATION[d]
        return a
    att = np.outer(edge(h), edge(w))
    r2, r3, r4, r5, r6 = (v * att for v in (r2, r3, r4, r5, r6))
    return np.stack([r4 * r4 + r6 * r6, (r4 + r5) * r6, r5 * r5 + r6 * r6, r4 * r2 + r6 * r3, r6 * r2 + r5 * r3], axis=-1)


def window_taps(winsize: int) -> np.ndarray:
    """A.1 item 5: the symmetric (2 m + 1)-tap Gaussian window, float32-rounded, k0 + 2 sum(k) = 1."""
    m = winsize // 2
    sigma = m * 0.3
    i = np.arange(m + 1, dtype=np.float64)
    k = np.exp(-(i * i) / (2.0 * sigma * sigma)).astype(np.float32).astype(np.float64)
    k = (k * (1.0 / (k[0] + 2.0 * k[1:].sum()))).astype(np.float32).astype(np.float64)
    return np.concatenate([k[:0:-1], k])


def blur_and_solve(M: np.ndarray, winsize: int) -> np.ndarray:
    """A.1 item 5: window blur (vertical, then horizontal, replicate borders) and the regularised 2 x 2 solve."""
    taps = window_taps(winsize)
    B = correlate1d(correlate1d(M, taps, axis=0, mode="nearest"), taps, axis=1, mode="nearest")
    g11, g12, g22, h1, h2 = (B[..., c] for c in range(5))
    idet = 1.0 / (g11 * g22 - g12 * g12 + 1e-3)
    return np.stack([(g11 * h2 - g12 * h1) * idet, (g22 * h1 - g12 * h2) * idet], axis=-1)


def calc_optical_flow_farneback(prev: np.ndarray, nxt: np.ndarray, pyr_scale=0.5, levels=2, winsize=40, iterations=3,
                                poly_n=5, poly_sigma=0.7, flags=256) -> np.ndarray:
    """uint8 [h, w] x 2 -> flow float64 [h, w, 2] (x, y).  OPTFLOW_FARNEBACK_GAUSSIAN (256) only; no initial flow."""
    if flags != 256:
        raise ValueError("farneback_f64: only OPTFLOW_FARNEBACK_GAUSSIAN is stated here")
    rows, cols = prev.shape
    flow = None
    for k in range(num_levels(rows, cols, pyr_scale, levels), -1, -1):
        scale = pyr_scale ** k
        sigma = (1.0 / scale - 1.0) * 0.5
        smooth = max(int(np.rint(sigma * 5.0)) | 1, 3)
        width, height = int(np.rint(cols * scale)), int(np.rint(rows * scale))
        if flow is None:
            flow = np.zeros((height, width, 2))
        else:
            flow = _resize_linear(flow, height, width) * (1.0 / pyr_scale)
        taps = _smoothing_taps(smooth, sigma)
        R = []
        for img in (prev, nxt):
            f = np.asarray(img, dtype=np.float64)
            f = correlate1d(correlate1d(f, taps, axis=1, mode="mirror"), taps, axis=0, mode="mirror")   # REFLECT_101
            R.append(poly_exp(_resize_linear(f, height, width), poly_n, poly_sigma))
        M = update_matrices(R[0], R[1], flow)
        for i in range(iterations):
            flow = blur_and_solve(M, winsize)
            if i < iterations - 1:
                M = update_matrices(R[0], R[1], flow)
    return flow
