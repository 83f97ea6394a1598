import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from predict_pv_yield_amd import hip_ops as K
d = "cuda"
b, nq, nk, h = 152, 128, 16384, 1
q = torch.randn(b, nq, 64, device=d); kv = torch.randn(b, nk, 128, device=d); dout = torch.randn(b, nq, 64, device=d)
kv16 = kv.to(torch.bfloat16)
def t(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out, lse = K.attention_fwd(q, kv, h, 0.125, bf16_operands=True)
print("fwd f32 kv  %.1f us" % t(lambda: K.attention_fwd(q, kv, h, 0.125, bf16_operands=True)))
print("fwd bf16 kv %.1f us" % t(lambda: K.attention_fwd(q, kv16, h, 0.125, bf16_operands=True)))
acc = torch.zeros_like(kv)
print("bwd f32 kv  %.1f us" % t(lambda: K.attention_bwd(q, kv, out, dout, lse, h, 0.125, bf16_operands=True, accumulate_dkv_into=acc)))
print("bwd bf16 kv %.1f us" % t(lambda: K.attention_bwd(q, kv16, out, dout, lse, h, 0.125, bf16_operands=True, accumulate_dkv_into=acc)))
x = torch.randn(b * nk, 38, device=d); w = torch.randn(128, 38, device=d)
print("to_kv f32 out  %.1f us" % t(lambda: K.gemm(x, w.t())))
print("to_kv bf16 out %.1f us" % t(lambda: K.gemm_rows_bf16out(x, w.t())))
