"""Sweeps the split-K sizing of the weight-gradient GEMMs (dW = dY^T X, K = 19 456 rows) on the PerceiverModel's shapes:
GEMM + slab sum per (target workgroups, minimum K chunk).   python tools/sweep_splitk.py"""
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
    for _ in range(30): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 30 * 1e3


shapes = [(64, 64, 3), (64, 512, 2), (256, 64, 2), (64, 1024, 1), (512, 64, 1), (37, 128, 1)]
data = [(torch.randn(rows, i, device=dev), torch.randn(rows, o, device=dev)) for i, o, _ in shapes]
for target in (128, 256, 512, 1024):
    for chunk in (256, 512, 1024, 2048):
        K.SPLITK_TARGET_WORKGROUPS, K.SPLITK_MIN_CHUNK = target, chunk
        ts = [bench(lambda: K.gemm_splitk(dy.t(), x)) for x, dy in data]
        tot = sum(t * c for t, (_, _, c) in zip(ts, shapes))
        print(f"target {target:5d} min chunk {chunk:5d}: " + "  ".join(f"{t:6.1f}" for t in ts) + f"   weighted sum {tot:7.1f} us")
