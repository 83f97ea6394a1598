"""Distance of the level kernel's flows from the CPU oracle on the bench input and on the fuzz tool's ill-conditioned input, and
the pyramid's time at B = 32 -- the yardstick of a numerical experiment on the flow kernels: build a second library with the
change and select it with PV_YIELD_LIB (round 6 measured the window matrices as ONE half-float term this way: -46 us, errors up
to 2.9e-2 px, rejected; profiles/r06/NOTES.md).
  python tools/probes/flow_error_vs_oracle.py
  PV_YIELD_LIB=/path/to/libpvyield_experiment.so python tools/probes/flow_error_vs_oracle.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import flow_oracle as fo
from predict_pv_yield_amd import hip_ops as K
from predict_pv_yield_amd.data.synthetic import advected_counts

dev = torch.device("cuda:0")
print("library:", os.environ.get("PV_YIELD_LIB", "(product)"))
# 1. the bench input: advected blob textures, reference parameters
raw, vel = advected_counts(batch=2, seed=1234)
u8 = fo.convert_10bpp_to_uint8(raw.transpose(0, 2, 1, 3, 4).reshape(-1, 12, 64, 64).astype(np.float32))[0]      # [22, 12, 64, 64]
got = K.farneback_stack(torch.from_numpy(u8).to(dev)).cpu().numpy()
errs = []
for s in range(u8.shape[0]):
    for j in range(11):
        errs.append(float(np.abs(got[s, j] - fo.calc_optical_flow_farneback(u8[s, j], u8[s, j + 1])).max()))
errs = np.array(errs)
print(f"bench input, 242 pairs, reference parameters: max {errs.max():.2e} px, 99th percentile {np.percentile(errs, 99):.2e}, median {np.median(errs):.2e}")
# 2. shifted noise (the fuzz tool's input), several parameter sets
rng = np.random.default_rng(11)
for (h, w, t, stacks, kw) in [(64, 64, 3, 6, dict(levels=2, winsize=40, iterations=3, poly_n=5, poly_sigma=0.7)),
                              (64, 64, 3, 6, dict(levels=2, winsize=21, iterations=3, poly_n=5, poly_sigma=1.1)),
                              (64, 64, 3, 6, dict(levels=2, winsize=9, iterations=3, poly_n=5, poly_sigma=1.1)),
                              (48, 40, 3, 6, dict(levels=2, winsize=15, iterations=3, poly_n=7, poly_sigma=1.5))]:
    base = rng.integers(0, 256, (stacks, 1, h + 8, w + 8), dtype=np.uint8)
    frames = np.stack([np.roll(base[:, 0], (i, 2 * i), axis=(1, 2))[:, 4:4 + h, 4:4 + w] for i in range(t)], axis=1)
    frames = np.ascontiguousarray((frames.astype(np.int16) + rng.integers(0, 6, frames.shape)).clip(0, 255).astype(np.uint8))
    lev = K.farneback_stack(torch.from_numpy(frames).to(dev), **kw).cpu().numpy()
    e, mx = [], 0.0
    for i in range(stacks):
        for j in range(t - 1):
            ref = fo.calc_optical_flow_farneback(frames[i, j], frames[i, j + 1], **kw)
            e.append(float(np.abs(lev[i, j] - ref).max()))
            mx = max(mx, float(np.abs(ref).max()))
    print(f"shifted noise {h}x{w} {kw}: max |flow| {mx:.1f} px; error max {max(e):.2e} median {np.median(e):.2e}")
# 3. time of the pyramid at B = 32
raw32, _ = advected_counts(batch=32, seed=1234)
u8b = torch.from_numpy(fo.convert_10bpp_to_uint8(raw32.transpose(0, 2, 1, 3, 4).reshape(-1, 12, 64, 64).astype(np.float32))[0]).to(dev)
for _ in range(3):
    K.farneback_stack(u8b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    K.farneback_stack(u8b)
e1.record()
torch.cuda.synchronize()
print(f"Farneback of 3 872 pairs: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
