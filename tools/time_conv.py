"""Times the 32 -> 32 channel bf16 Conv3D forward / dgrad-shaped launches (the kernels behind bench.py's roofline line)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predict_pv_yield_amd import hip_ops as K

b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
bias = torch.randn(32, device=dev) * 0.1
wp = K.conv3d_pack_weight_bf16(w)


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


for (t, h) in [(17, 62), (15, 60), (13, 58)]:
    x = torch.randn(b, t, h, h, 32, device=dev).to(torch.bfloat16)
    ms = timeit(lambda: K.conv3d_fwd_bf16(x, None, wp, bias, 32, 32, (0, 0, 0), True, False))
    to, ho = t - 2, h - 2
    fl = 2.0 * b * 32 * 32 * 27 * to * ho * ho
    print(f"fwd   in {t}x{h}x{h}: {ms * 1e3:7.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s")
    ms = timeit(lambda: K.conv3d_fwd_bf16(x, None, wp, bias, 32, 32, (0, 0, 0), True, False, want_relu_mask=True))
    print(f"fwd + relu mask out : {ms * 1e3:7.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s")
    ms = timeit(lambda: K.conv3d_fwd_bf16(x, None, wp, bias, 32, 32, (0, 0, 0), True, True))
    print(f"fwd NCDHW out       : {ms * 1e3:7.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s")
    # dgrad shape: dy [to, ho, ho] padded by 2 -> dx [t, h, h], gated by the producer's input
    dy = torch.randn(b, to, ho, ho, 32, device=dev).to(torch.bfloat16)
    gate = torch.randn(b, t, h, h, 32, device=dev).to(torch.bfloat16)
    ms = timeit(lambda: K.conv3d_fwd_bf16(dy, None, wp, None, 32, 32, (2, 2, 2), False, False, out_gate=gate))
    fl = 2.0 * b * 32 * 32 * 27 * t * h * h
    print(f"dgrad out {t}x{h}x{h}: {ms * 1e3:7.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s (padded-tap flops) [bf16 gate]")
    gmask = torch.randint(-2 ** 31, 2 ** 31 - 1, K.relu_mask_shape(b, t, h, h), dtype=torch.int32, device=dev)
    ms = timeit(lambda: K.conv3d_fwd_bf16(dy, None, wp, None, 32, 32, (2, 2, 2), False, False, out_gate=gate, out_gate_mask=gmask))
    print(f"dgrad, 1-bit mask gate: {ms * 1e3:7.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s")
    ms = timeit(lambda: K.conv3d_fwd_bf16(dy, None, wp, None, 32, 32, (2, 2, 2), False, False))
    print(f"dgrad, no gate        : {ms * 1e3:7.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s")

# first layer: f32 NCDHW input, 11 channels
x = torch.randn(b, 11, 18, 64, 64, device=dev)
w1 = torch.randn(32, 11, 3, 3, 3, device=dev) * 0.05
wp1 = K.conv3d_pack_weight_bf16(w1)
ms = timeit(lambda: K.conv3d_fwd_bf16(K.pack_ncdhw_f32_to_ndhwc_bf16(x), None, wp1, bias, 11, 32, (0, 0, 0), True, False))
print(f"layer 1 pack + conv        : {ms * 1e3:7.1f} us")
ms = timeit(lambda: K.conv3d_fwd_bf16_f32in(x, wp1, bias, 32, (0, 0, 0), True, want_packed=True))
print(f"layer 1 fused (with xp)    : {ms * 1e3:7.1f} us")
ms = timeit(lambda: K.conv3d_fwd_bf16_f32in(x, wp1, bias, 32, (0, 0, 0), True, want_packed=True, want_relu_mask=True))
print(f"layer 1 fused (xp + mask)  : {ms * 1e3:7.1f} us")
ms = timeit(lambda: K.conv3d_fwd_bf16_f32in(x, wp1, bias, 32, (0, 0, 0), True, want_packed=False))
print(f"layer 1 fused (no xp)      : {ms * 1e3:7.1f} us")
xp1 = K.pack_ncdhw_f32_to_ndhwc_bf16(x)
ms = timeit(lambda: K.conv3d_fwd_bf16(xp1, None, wp1, bias, 11, 32, (0, 0, 0), True, False))
print(f"layer 1 conv on packed bf16: {ms * 1e3:7.1f} us")
