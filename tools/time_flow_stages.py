"""Per-stage device time of the config-3 advection pipeline (pv_stage_timing), B = 32 by default."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predict_pv_yield_amd import hip_ops as K
from predict_pv_yield_amd import optical_flow as of

b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
# the SURVEY section-8(d) input (the tensor bench.py's config-3 leg, the CPU leg and the joined-model parity test use);
# "noise" as a second argument gives the uniform-noise stacks of rounds 1-3
if len(sys.argv) > 2 and sys.argv[2] == "noise":
    raw = torch.randint(0, 1021, (b, 12, 11, 64, 64), dtype=torch.int16, device=dev)
else:
    from predict_pv_yield_amd.data.synthetic import advected_counts
    raw = torch.from_numpy(advected_counts(batch=b, seed=1234)[0]).to(dev)
for _ in range(3):
    of.advect_future_frames(raw, 6)
iters = 10
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    of.advect_future_frames(raw, 6)
e1.record()
torch.cuda.synchronize()
print(f"pipeline {e0.elapsed_time(e1) / 30 * 1e3:8.1f} us  (B={b}) as the product runs it (30 batches back to back, no events between the launches)")
with K.stage_timing() as st:
    e0.record()
    for _ in range(iters):
        of.advect_future_frames(raw, 6)
    e1.record()
torch.cuda.synchronize()
print(f"pipeline {e0.elapsed_time(e1) / iters * 1e3:8.1f} us  (B={b}) with a HIP event at every stage boundary (~25 us each)")
for k, (ms, n) in st.stages.items():
    print(f"  {k:58s} {ms / iters * 1e3:8.1f} us  ({n // iters} launch groups)")
