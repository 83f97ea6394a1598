"""A/B sweep: 32->32 bf16 forward with and without the relu-mask output over output extents (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predict_pv_yield_amd import hip_ops as K
dev = torch.device("cuda:0")
w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
bias = torch.randn(32, device=dev) * 0.1
wp = K.conv3d_pack_weight_bf16(w)

def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

b = 32
for t in (14, 16):
    for h in (58, 60, 62, 64, 66):
        x = torch.randn(b, t, h, h, 32, device=dev).to(torch.bfloat16)
        a = timeit(lambda: K.conv3d_fwd_bf16(x, None, wp, bias, 32, 32, (0, 0, 0), True, False))
        m = timeit(lambda: K.conv3d_fwd_bf16(x, None, wp, bias, 32, 32, (0, 0, 0), True, False, want_relu_mask=True))
        a2 = timeit(lambda: K.conv3d_fwd_bf16(x, None, wp, bias, 32, 32, (0, 0, 0), True, False))
        print(f"in t={t} h={h}: out {t-2}x{h-2}: plain {a:6.1f} / {a2:6.1f} us   +mask {m:6.1f} us")
