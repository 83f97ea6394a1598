"""LayerNorm forward / backward on the Perceiver shapes: time and error against torch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predict_pv_yield_amd import hip_ops as K
dev = torch.device("cuda:0")
def bench(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3
for rows, d in [(152 * 16384, 38), (19456, 64), (152 * 4096, 38), (5000, 64), (70001, 6), (65536, 12)]:
    x = torch.randn(rows, d, device=dev); g = torch.randn(d, device=dev); bb = torch.randn(d, device=dev); dy = torch.randn(rows, d, device=dev)
    y, mean, rstd = K.layernorm_fwd(x, g, bb)
    ref = torch.nn.functional.layer_norm(x, (d,), g, bb)
    dx, dw, db = K.layernorm_bwd(x, g, dy, mean, rstd)
    xr = x.clone().requires_grad_(True); gr = g.clone().requires_grad_(True); br = bb.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (d,), gr, br).backward(dy)
    e = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-12))
    tf = bench(lambda: K.layernorm_fwd(x, g, bb)); tb = bench(lambda: K.layernorm_bwd(x, g, dy, mean, rstd)); tb2 = bench(lambda: K.layernorm_bwd(x, g, dy, mean, rstd, need_dx=False))
    print(f"rows {rows} d {d}: fwd {tf:.1f} us ({rows*d*8/tf/1e6:.2f} TB/s) bwd {tb:.1f} us, bwd no dx {tb2:.1f} us; err y {e(y, ref):.1e} dx {e(dx, xr.grad):.1e} dw {e(dw, gr.grad):.1e} db {e(db, br.grad):.1e}")
