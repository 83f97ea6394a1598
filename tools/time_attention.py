"""Times the fused attention forward / backward (bf16 operands) on the shapes the two Perceiver models use."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predict_pv_yield_amd import hip_ops as K

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
    return e0.elapsed_time(e1) / n * 1e3


for name, b, nq, nk, heads in [("PerceiverModel cross (8 x 19 frames, 64x64 context)", 152, 128, 4096, 1),
                               ("exp003 cross (8 x 19 images, 128x128 context)", 152, 128, 16384, 1),
                               ("exp003 latent self-attention", 152, 128, 128, 8)]:
    q = torch.randn(b, nq, heads * 64, device=dev)
    kv = torch.randn(b, nk, 2 * heads * 64, device=dev)
    dout = torch.randn_like(q)
    for bf in (True, False):
        out, lse = K.attention_fwd(q, kv, heads, 0.125, bf16_operands=bf)
        tf = timeit(lambda: K.attention_fwd(q, kv, heads, 0.125, bf16_operands=bf))
        tb = timeit(lambda: K.attention_bwd(q, kv, out, dout, lse, heads, 0.125, bf16_operands=bf))
        fl = 4.0 * b * heads * nq * nk * 64
        print(f"{name:48s} {'bf16' if bf else 'f32 '}: fwd {tf:8.1f} us ({fl / tf / 1e6:6.1f} TF)  bwd {tb:8.1f} us ({2.5 * fl / tb / 1e6:6.1f} TF)")
