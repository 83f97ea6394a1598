"""Farnebäck on the reference's own frame size: one stack of 25 consecutive 704 x 548 frames (24 fields, the super-batch of
notebooks/13_...ipynb:415-441 / 12_just_3d_conv.ipynb:611), per-stage device time.  `python tools/time_flow_fullframe.py [frames]`"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from predict_pv_yield_amd import hip_ops as K

t = int(sys.argv[1]) if len(sys.argv) > 1 else 25
h, w = 548, 704
dev = torch.device("cuda:0")
rng = np.random.default_rng(7)
# a band-limited texture drifting ~1.5 px per frame (what the satellite frames look like to the algorithm)
base = rng.standard_normal((h + 128, w + 128)).astype(np.float32)
f = np.fft.rfft2(base)
ky, kx = np.meshgrid(np.fft.fftfreq(base.shape[0]), np.fft.rfftfreq(base.shape[1]), indexing="ij")
f *= np.exp(-(kx ** 2 + ky ** 2) / (2 * 0.03 ** 2))
base = np.fft.irfft2(f, s=base.shape)
base = (base - base.min()) / (base.max() - base.min()) * 255
frames = np.stack([np.roll(base, (i, 2 * i), (0, 1))[:h, :w] for i in range(t)]).astype(np.uint8)
stack = torch.from_numpy(frames).to(dev)
for _ in range(2):
    flow = K.farneback_stack(stack)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 10
e0.record()
for _ in range(n):
    flow = K.farneback_stack(stack)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
print(f"{t - 1} fields of {w} x {h}: {us:9.1f} us = {us / (t - 1):7.1f} us per field; mean flow "
      f"{flow[..., 0].mean().item():+.3f}, {flow[..., 1].mean().item():+.3f} px")
with K.stage_timing() as st:
    for _ in range(n):
        K.farneback_stack(stack)
torch.cuda.synchronize()
for k, (ms, c) in st.stages.items():
    print(f"  {k:58s} {ms / n * 1e3:9.1f} us  ({c // n} launch groups)")
