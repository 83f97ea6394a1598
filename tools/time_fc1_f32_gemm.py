"""fc1 of the f32 model (32 x 1 003 520 -> 128) through pv_gemm_f32 (bf16x3 on the matrix cores) against the f32 kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predict_pv_yield_amd import hip_ops as K

dev = torch.device("cuda:0")
m, n, k = 32, 128, 1003520
g0 = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(m, k, device=dev, generator=g0)
w = torch.randn(n, k, device=dev, generator=g0) * 0.01
b = torch.zeros(n, device=dev)
g = torch.randn(m, n, device=dev, generator=g0)


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, out


us, y0 = t(lambda: K.linear_fwd_f32(x, w, b, relu=False))
print(f"linear_fwd_f32            {us:8.1f} us")
us, y1 = t(lambda: K.gemm_splitk(x, w.t()))
print(f"gemm_splitk(x, W^T)       {us:8.1f} us   max rel diff {((y1 - y0).abs().max() / y0.abs().max()).item():.2e}")
ref = (x.double() @ w.double().t()).float()
print(f"  vs f64: fwd kernel {((y0 - ref).abs().max() / ref.abs().max()).item():.2e}  gemm {((y1 - ref).abs().max() / ref.abs().max()).item():.2e}")
us, r = t(lambda: K.linear_bwd_f32(x, w, g, None))
print(f"linear_bwd_f32 (dx,dw,db) {us:8.1f} us")
us, dx = t(lambda: K.gemm(g, w))
print(f"gemm(g, W) = dx           {us:8.1f} us   max rel diff {((dx - r[0]).abs().max() / r[0].abs().max()).item():.2e}")
us, dw = t(lambda: K.gemm(g.t(), x))
print(f"gemm(g^T, x) = dW         {us:8.1f} us   max rel diff {((dw - r[1]).abs().max() / r[1].abs().max()).item():.2e}")
