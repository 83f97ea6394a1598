"""Runs one fused-attention launch shape a few times (bf16 operands): target for rocprofv3 --pmc passes.
   python tools/run_attn_only.py latent|cross fwd|bwd"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predict_pv_yield_amd import hip_ops as K
dev = torch.device("cuda:0")
shape, which = sys.argv[1], sys.argv[2]
b, nq, nk, heads = (152, 128, 128, 8) if shape == "latent" else (152, 128, 4096, 1)
q = torch.randn(b, nq, heads * 64, device=dev); kv = torch.randn(b, nk, 2 * heads * 64, device=dev); dout = torch.randn_like(q)
out, lse = K.attention_fwd(q, kv, heads, 0.125, bf16_operands=True)
torch.cuda.synchronize()
for _ in range(5):
    if which == "fwd":
        K.attention_fwd(q, kv, heads, 0.125, bf16_operands=True)
    else:
        K.attention_bwd(q, kv, out, dout, lse, heads, 0.125, bf16_operands=True)
torch.cuda.synchronize()
print("done")
