"""fc1's one-pass backward alone, by Adam-scalar source and moment layout: host scalars / device scalars x row-major / tiled moments
(why does the step replayed as a HIP graph run this kernel 65 us slower than the eager step?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from predict_pv_yield_amd import hip_ops as K
dev = torch.device("cuda:0")
m, n, k = 32, 128, 1003520
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(m, k, generator=g, device=dev).relu_().to(torch.bfloat16)
dy = torch.randn(m, n, generator=g, device=dev) * 1e-3
ym = torch.rand(m, n, generator=g, device=dev)
p = torch.randn(n, k, generator=g, device=dev) * 0.01
ea, es = torch.zeros_like(p), torch.zeros_like(p)
sh = p.to(torch.bfloat16)
scal = torch.zeros(8, device=dev)
stepc = torch.zeros(1, dtype=torch.int32, device=dev)
K.adam_scalars_advance(scal, stepc)


def timeit(fn, n_it=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n_it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n_it * 1e3


for rounds in range(2):
    for tiled in (False, True):
        a = timeit(lambda: K.linear_wgrad_dx_adam_bf16(x, dy, ym, p, ea, es, sh, 3, gate_dx_by_x=True, moments_tiled=tiled))
        b = timeit(lambda: K.linear_wgrad_dx_adam_dev_bf16(x, dy, ym, p, ea, es, sh, scal, gate_dx_by_x=True, moments_tiled=tiled))
        print(f"moments_tiled={tiled}: host scalars {a:7.1f} us   device scalars {b:7.1f} us")
