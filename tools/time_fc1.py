"""Times the fc1-shaped (m=32, n=128, k=1,003,520) bf16 linear forward / dx passes and checks them against torch."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predict_pv_yield_amd import hip_ops as K

dev = torch.device("cuda:0")
m, n, k = 32, 128, 32 * 10 * 56 * 56
x = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
w = (torch.randn(n, k, device=dev) * 0.01).to(torch.bfloat16)
bias = torch.randn(n, device=dev)
dy = torch.randn(m, n, device=dev)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


y = K.linear_fwd_bf16(x, w, bias)
ref = x.float() @ w.float().t() + bias
print("fwd max rel err", ((y - ref).abs().max() / ref.abs().max()).item())
dx, _, db = K.linear_bwd_bf16(x, w, dy, None, need_dx=True, need_dw=False)
refdx = dy @ w.float()
print("dx max rel err", ((dx.float() - refdx).abs().max() / refdx.abs().max()).item())
us = timeit(lambda: K.linear_fwd_bf16(x, w, bias))
print(f"fwd {us:7.1f} us  {(n + m) * k * 2 / us / 1e6:6.2f} TB/s (weights + x)")
us = timeit(lambda: K.linear_bwd_bf16(x, w, dy, None, need_dx=True, need_dw=False))
print(f"dx  {us:7.1f} us  {(n + m) * k * 2 / us / 1e6:6.2f} TB/s (weights + dx)")

# fc1's backward: two kernels (dx from the bf16 copy, then fused wgrad + Adam) vs the single pass that does both
p32 = (torch.randn(n, k, device=dev) * 0.01)
ma, va = torch.zeros_like(p32), torch.zeros_like(p32)
sh = p32.to(torch.bfloat16)
y = torch.relu(torch.randn(m, n, device=dev))
us_dx = timeit(lambda: K.linear_bwd_bf16(x, sh, dy, y, need_dx=True, need_dw=False), iters=10)
us_up = timeit(lambda: K.linear_wgrad_adam_bf16(x, dy, y, p32, ma, va, sh, 5), iters=10)
us_one = timeit(lambda: K.linear_wgrad_dx_adam_bf16(x, dy, y, p32, ma, va, sh, 5), iters=10)
byt = n * k * 26 + m * k * 2
print(f"dx kernel {us_dx:7.1f} us + fused wgrad/Adam {us_up:7.1f} us = {us_dx + us_up:7.1f} us;  single pass {us_one:7.1f} us "
      f"({(byt + m * k * 2) / us_one / 1e6:5.2f} TB/s)")
for _ in range(2):
    a = timeit(lambda: K.linear_wgrad_adam_bf16(x, dy, y, p32, ma, va, sh, 5), iters=10)
    b = timeit(lambda: K.linear_wgrad_dx_adam_bf16(x, dy, y, p32, ma, va, sh, 5), iters=10)
    print(f"  again: fused wgrad/Adam {a:7.1f} us   single pass with dx {b:7.1f} us")
