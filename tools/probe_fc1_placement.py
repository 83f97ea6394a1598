"""The fc1 backward pass (HBM-bound, 3.5 GB per launch) at the benched size, on freshly allocated tensors, with the Adam
moments row-major and tile by tile (moments_tiled), alternating inside one process: the measurement behind
optim.TILE_LARGE_MOMENTS (the row-major form ran 690 us on some devices / placements and 780 us on others).
   python tools/probe_fc1_placement.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predict_pv_yield_amd import hip_ops as K

dev = torch.device("cuda:0")
m, n, k = 32, 128, 1003520


def timeit(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def separate():
    return (torch.empty(n, k, device=dev), torch.empty(n, k, device=dev), torch.empty(n, k, device=dev),
            torch.empty(n, k, device=dev, dtype=torch.bfloat16))



# separate allocations (what the application does), row-major vs tiled moment arrays, alternating in one process
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(m, k, device=dev, generator=g).clamp_min(0).to(torch.bfloat16)
dy = torch.randn(m, n, device=dev, generator=g)
y = torch.rand(m, n, device=dev, generator=g)
for rep in range(4):
    w, ea, es, sh = separate()
    w.normal_(0, 0.01); ea.zero_(); es.zero_(); sh.copy_(w.to(torch.bfloat16))
    st = [0]
    out = []
    for tiled in (False, True, False, True):
        def f():
            st[0] += 1
            K.linear_wgrad_dx_adam_bf16(x, dy, y, w, ea, es, sh, st[0], lr=0.0, need_dx=True, need_db=True, gate_dx_by_x=True,
                                        moments_tiled=tiled)

        timeit(f, 4)
        out.append(timeit(f, 10))
    print(f"alloc {rep}: row-major {out[0]:6.1f} {out[2]:6.1f}   tiled {out[1]:6.1f} {out[3]:6.1f} us")
    del w, ea, es, sh
    junk = torch.empty((rep + 1) * 37 * (1 << 20) + 4096 * 13, dtype=torch.uint8, device=dev)   # shifts the next allocations
