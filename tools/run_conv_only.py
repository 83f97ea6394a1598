"""Runs the layer-1 conv forward (or dgrad, or wgrad) kernel a few times: target for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predict_pv_yield_amd import hip_ops as K
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
b = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
x = torch.randn(b, 16, 62, 62, 32, device=dev).to(torch.bfloat16)
w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
bias = torch.zeros(32, device=dev)
wp = K.conv3d_pack_weight_bf16(w)
wpT = K.conv3d_pack_weight_bf16(w, transpose_flip=True)
y = K.conv3d_fwd_bf16(x, None, wp, bias, 32, 32, (0, 0, 0), True, False)
dy = torch.randn_like(y)
x0 = torch.randn(b, 11, 18, 64, 64, device=dev)
w0 = torch.randn(32, 11, 3, 3, 3, device=dev) * 0.05
wp0 = K.conv3d_pack_weight_bf16(w0)
y0, xp0 = K.conv3d_fwd_bf16_f32in(x0, wp0, bias, 32)[:2]
dy0 = torch.randn_like(y0)
torch.cuda.synchronize()
for _ in range(5):
    if which == "fwd":
        K.conv3d_fwd_bf16(x, None, wp, bias, 32, 32, (0, 0, 0), True, False)
    elif which == "dgrad":      # layer-1 dgrad: dy [B,14,60,60,32] -> dx [B,16,62,62,32], gated by the producer's activation x
        K.conv3d_fwd_bf16(dy, None, wpT, None, 32, 32, (2, 2, 2), False, False, out_gate=x)
    elif which == "first":      # first layer from the f32 NCDHW input [B,11,18,64,64]
        K.conv3d_fwd_bf16_f32in(x0, wp0, bias, 32)
    elif which == "wgrad16":
        K.conv3d_bwd_weight_bf16(xp0, dy0, None, 11, 32, (0, 0, 0))
    elif which == "wgrad":
        K.conv3d_bwd_weight_bf16(x, dy, None, 32, 32, (0, 0, 0))
torch.cuda.synchronize()
print("done")
