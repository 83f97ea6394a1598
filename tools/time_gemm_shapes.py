"""Times pv_gemm_f32 on the PerceiverModel's actual linear-layer shapes (19 456 rows = 152 frames x 128 latents):
forward x @ W^T, backward dx = dy @ W and dW = dy^T @ x (split-K)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predict_pv_yield_amd import hip_ops as K
dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 19456


def bench(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


tot = 0.0
for (i, o, count) in [(64, 64, 3), (64, 512, 2), (256, 64, 2), (64, 1024, 1), (512, 64, 1)]:
    x = torch.randn(rows, i, device=dev); w = torch.randn(o, i, device=dev); dy = torch.randn(rows, o, device=dev)
    tf = bench(lambda: K.gemm(x, w.t()))
    tdx = bench(lambda: K.gemm(dy, w))
    tdw = bench(lambda: K.gemm_splitk(dy.t(), x))
    fl = 2.0 * rows * i * o
    tot += count * (tf + tdx + tdw)
    print(f"linear {i:4d} -> {o:4d}: fwd {tf:7.1f} us ({fl / tf / 1e6:6.1f} TF)  dx {tdx:7.1f} us ({fl / tdx / 1e6:6.1f} TF)  dW {tdw:7.1f} us ({fl / tdw / 1e6:6.1f} TF)")
print(f"per layer (3x 64->64, 2x 64->512, 2x 256->64, 64->1024, 512->64), fwd + dx + dW: {tot:.0f} us")
