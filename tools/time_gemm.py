"""Times pv_gemm_f32 on the Perceiver's shapes (B' = 992 stacked frames, 128 latents x 64, 4096 context positions)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predict_pv_yield_amd import hip_ops as K
dev = torch.device("cuda:0")
bp = int(sys.argv[1]) if len(sys.argv) > 1 else 992


def bench(name, fn, flops):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name:44s} {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TFLOP/s")


q = torch.randn(bp, 128, 64, device=dev); kv = torch.randn(bp, 4096, 128, device=dev)
k, v = kv[..., :64], kv[..., 64:]
s = torch.empty(bp, 128, 4096, device=dev)
bench("scores  q.k^T  [128x64]x[64x4096] /frame", lambda: K.gemm(q, k.transpose(-1, -2), out=s), 2 * bp * 128 * 64 * 4096)
bench("out     p.v    [128x4096]x[4096x64]", lambda: K.gemm(s, v), 2 * bp * 128 * 64 * 4096)
bench("dv      p^T.do [4096x128]x[128x64]", lambda: K.gemm(s.transpose(-1, -2), q), 2 * bp * 128 * 64 * 4096)
ctx = torch.randn(bp * 4096, 37, device=dev); w = torch.randn(128, 37, device=dev)
bench("to_kv   [4.06M x 37] x [37 x 128]", lambda: K.gemm(ctx, w.t()), 2 * bp * 4096 * 37 * 128)
dy = torch.randn(bp * 4096, 128, device=dev)
bench("to_kv dW (split-K) [128 x 4.06M] x [4.06M x 37]", lambda: K.gemm_splitk(dy.t(), ctx), 2 * bp * 4096 * 37 * 128)
x = torch.randn(bp * 128, 64, device=dev); w1 = torch.randn(512, 64, device=dev)
bench("ff1     [127k x 64] x [64 x 512]", lambda: K.gemm(x, w1.t()), 2 * bp * 128 * 64 * 512)
bench("fused attention fwd (cross, 128 x 4096)", lambda: K.attention_fwd(q, kv, 1, 0.125), 2 * 2 * bp * 128 * 64 * 4096)
ql = torch.randn(bp, 128, 512, device=dev); kvl = torch.randn(bp, 128, 1024, device=dev)
bench("fused attention fwd (latent, 8 heads 128 x 128)", lambda: K.attention_fwd(ql, kvl, 8, 0.125), 2 * 2 * bp * 8 * 128 * 64 * 128)
o, l = K.attention_fwd(q, kv, 1, 0.125); do = torch.randn_like(o)
bench("fused attention bwd (cross, 128 x 4096)", lambda: K.attention_bwd(q, kv, o, do, l, 1, 0.125), 2 * 5 * bp * 128 * 64 * 4096)
o, l = K.attention_fwd(ql, kvl, 8, 0.125); do = torch.randn_like(o)
bench("fused attention bwd (latent, 8 heads 128 x 128)", lambda: K.attention_bwd(ql, kvl, o, do, l, 8, 0.125), 2 * 5 * bp * 8 * 128 * 64 * 128)
