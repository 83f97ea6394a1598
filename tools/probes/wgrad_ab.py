"""Weight-gradient launches of the headline step's four conv layers, timed alone (HIP events, median of 30): run once per form,
e.g. `PV_WGRAD_MWV=4 python tools/probes/wgrad_ab.py; python tools/probes/wgrad_ab.py` on the same box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from predict_pv_yield_amd import hip_ops as K

dev = torch.device("cuda:0")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = torch.Generator(device=dev).manual_seed(1)
print("PV_WGRAD_MWV =", os.environ.get("PV_WGRAD_MWV", "(default)"))
tot = 0.0
for ci, t, h in ((11, 18, 64), (32, 16, 62), (32, 14, 60), (32, 12, 58)):
    x = torch.randn(b, t, h, h, K.bf16_cpad(ci), device=dev, generator=g).to(torch.bfloat16)
    dy = torch.randn(b, t - 2, h - 2, h - 2, 32, device=dev, generator=g).to(torch.bfloat16)
    fn = lambda: K.conv3d_bwd_weight_bf16(x, dy, None, ci, 32, (0, 0, 0))
    for _ in range(5):
        out = fn()
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    tot += ts[15]
    dw = out[0] if isinstance(out, (tuple, list)) else out
    print(f"  ci={ci:2d} [{b},{t},{h},{h}]: {ts[15]:7.1f} us (min {ts[0]:.1f})  checksum {float(dw.double().abs().sum()):.6e}")
print(f"  sum {tot:.1f} us")
