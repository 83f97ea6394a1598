"""Synthetic satellite stacks with a known-answer motion field.

Frame 0 of each (b, c) is a dense band-limited blob texture; frames 1..T-1 are the same texture
advected by a constant sub-pixel velocity drawn U(-vmax, vmax) px/frame, so the true optical flow
is known.  Values are 10-bit counts (0..1023) as in the EUMETSAT int16 zarr the reference loads
(notebooks/13_3d_conv_with_optical_flow_predictions.ipynb:415-441).
"""
import numpy as np


def blob_texture_sequence(rng: np.random.Generator, t: int, h: int, w: int, velocity, n_blobs: int = 60,
                          margin: int = 10) -> np.ndarray:
    """[t, h, w] float64 in ~0..1000: sum of Gaussian blobs translated by `velocity` (vx, vy) px/frame."""
    vx, vy = velocity
    # scale blob count with area so bigger tiles stay dense
    n = max(8, int(round(n_blobs * (h + 2 * margin) * (w + 2 * margin) / float((64 + 2 * margin) ** 2))))
    reach_x = abs(vx) * t + margin
    reach_y = abs(vy) * t + margin
    cx = rng.uniform(-reach_x, w + reach_x, n)
    cy = rng.uniform(-reach_y, h + reach_y, n)
    sig = rng.uniform(2.5, 7.0, n)
    amp = rng.uniform(0.3, 1.0, n)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    out = np.zeros((t, h, w), np.float64)
    for i in range(t):
        dx = xx[None] - (cx[:, None, None] + vx * i)
        dy = yy[None] - (cy[:, None, None] + vy * i)
        out[i] = (amp[:, None, None] * np.exp(-(dx * dx + dy * dy) / (2.0 * sig[:, None, None] ** 2))).sum(0)
    peak = out.max()
    return out * (1000.0 / max(peak, 1e-9))


def advected_counts(batch: int, t: int = 12, channels: int = 11, h: int = 64, w: int = 64, seed: int = 1234,
                    vmax: float = 3.0):
    """Returns (raw int16 [B, T, C, H, W] 10-bit counts, velocity f32 [B, C, 2] px/frame)."""
    rng = np.random.default_rng(seed)
    raw = np.empty((batch, t, channels, h, w), np.int16)
    vel = np.empty((batch, channels, 2), np.float32)
    for b in range(batch):
        for c in range(channels):
            v = rng.uniform(-vmax, vmax, 2)
            vel[b, c] = v
            seq = blob_texture_sequence(rng, t, h, w, v)
            raw[b, :, c] = np.clip(np.rint(seq), 0, 1023).astype(np.int16)
    return raw, vel
