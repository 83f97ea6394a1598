"""Ill-conditioned cases of tools/fuzz_flow_fused.py (winsize 5 on shifted noise, flows of 10-25 px): both HIP forms against
the CPU oracle, pair by pair."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import flow_oracle as fo
from predict_pv_yield_amd import hip_ops as K
dev = torch.device("cuda:0")
rng = np.random.default_rng(11)
for (h, w, t, stacks, kw) in [
        (64, 44, 3, 3, dict(levels=3, winsize=5, iterations=1, poly_n=5, poly_sigma=1.1)),
        (44, 60, 2, 3, dict(levels=3, winsize=5, iterations=2, poly_n=5, poly_sigma=1.5)),
        (64, 64, 3, 3, dict(levels=3, winsize=5, iterations=3, poly_n=5, poly_sigma=1.5)),
        (64, 32, 3, 3, dict(levels=1, winsize=5, iterations=2, poly_n=7, poly_sigma=1.5)),
        (64, 64, 3, 3, dict(levels=2, winsize=21, iterations=3, poly_n=5, poly_sigma=1.1)),
        (64, 64, 3, 3, dict(levels=2, winsize=9, iterations=3, poly_n=5, poly_sigma=1.1))]:
    base = rng.integers(0, 256, (stacks, 1, h + 8, w + 8), dtype=np.uint8)
    frames = np.stack([np.roll(base[:, 0], (i, 2 * i), axis=(1, 2))[:, 4:4 + h, 4:4 + w] for i in range(t)], axis=1)
    frames = np.ascontiguousarray((frames.astype(np.int16) + rng.integers(0, 6, frames.shape)).clip(0, 255).astype(np.uint8))
    u8 = torch.from_numpy(frames).to(dev)
    os.environ["PV_FARNEBACK_TWO_LAUNCH_ITERATION"] = "1"
    two = K.farneback_stack(u8, **kw).cpu().numpy()
    del os.environ["PV_FARNEBACK_TWO_LAUNCH_ITERATION"]
    lev = K.farneback_stack(u8, **kw).cpu().numpy()
    e_two = e_lev = 0.0
    mx = 0.0
    for i in range(stacks):
        for j in range(t - 1):
            ref = fo.calc_optical_flow_farneback(frames[i, j], frames[i, j + 1], **kw)
            e_two = max(e_two, float(np.abs(two[i, j] - ref).max()))
            e_lev = max(e_lev, float(np.abs(lev[i, j] - ref).max()))
            mx = max(mx, float(np.abs(ref).max()))
    print(f"{h}x{w} {kw}: max|flow| {mx:.1f} px; |two-launch - oracle| {e_two:.2e}  |level kernel - oracle| {e_lev:.2e}  "
          f"|level - two-launch| {float(np.abs(lev - two).max()):.2e}", flush=True)
