"""Config-3 advection pipeline at B = 32, timed as the product runs it (20 batches back to back) and with pv_stage_timing
events at every stage boundary: what the instrumentation of the per-stage table costs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predict_pv_yield_amd import hip_ops as K
from predict_pv_yield_amd import optical_flow as of
dev = torch.device("cuda:0")
raw = torch.randint(0, 1021, (32, 12, 11, 64, 64), dtype=torch.int16, device=dev)
def t(iters=20, timing=False):
    for _ in range(5): of.advect_future_frames(raw, 6)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if timing:
        with K.stage_timing() as st:
            e0.record()
            for _ in range(iters): of.advect_future_frames(raw, 6)
            e1.record()
    else:
        e0.record()
        for _ in range(iters): of.advect_future_frames(raw, 6)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for r in range(3):
    print(f"plain {t():8.1f} us   with stage timing {t(timing=True):8.1f} us")
