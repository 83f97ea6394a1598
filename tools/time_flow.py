"""Times the config-3 advection pipeline (u8 -> Farneback x11 -> weighted mean -> normalise -> 6 remaps) on the GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predict_pv_yield_amd import optical_flow as of
from predict_pv_yield_amd import hip_ops as K
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
raw = torch.randint(0, 1021, (b, 12, 11, 64, 64), dtype=torch.int16, device=dev)
for _ in range(2):
    out = of.advect_future_frames(raw, 6)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 5
for _ in range(n):
    out = of.advect_future_frames(raw, 6)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"advect_future_frames B={b}: {dt*1e3:.2f} ms  -> {b/dt:.0f} samples/s, {b*121/dt:.0f} pairs/s")
# remap alone: 16 B per output pixel algorithmic
src = torch.randn(b * 11, 64, 64, device=dev); fl = torch.randn(b * 11, 64, 64, 2, device=dev)
for _ in range(3): o = K.remap_bilinear(src, fl, 6, 1.0, 1, 0.0)
torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): o = K.remap_bilinear(src, fl, 6, 1.0, 1, 0.0)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
byt = b * 11 * 6 * 4096 * 16
print(f"remap x6 B={b}: {ms*1e3:.1f} us, {byt/ms/1e6:.1f} GB/s algorithmic (16 B/px)")
