"""Times the bf16 Conv3D weight-gradient launches of the headline model (B=32)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predict_pv_yield_amd import hip_ops as K

b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")


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
    return e0.elapsed_time(e1) / n


for (ci, t, h) in [(11, 18, 64), (32, 16, 62), (32, 14, 60), (32, 12, 58)]:
    cpad = K.bf16_cpad(ci)
    x = torch.randn(b, t, h, h, cpad, device=dev).to(torch.bfloat16)
    to, ho = t - 2, h - 2
    dy = torch.randn(b, to, ho, ho, 32, device=dev).to(torch.bfloat16)
    ms = timeit(lambda: K.conv3d_bwd_weight_bf16(x, dy, None, ci, 32, (0, 0, 0)))
    fl = 2.0 * b * 32 * ci * 27 * to * ho * ho
    print(f"wgrad cin {ci:2d} in {t}x{h}x{h}: {ms * 1e3:7.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s (incl. slab reduce)")
