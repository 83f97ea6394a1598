"""Does the fc1 backward pass (HBM-bound, 3.5 GB per launch) depend on WHERE its tensors were allocated?
Times pv_linear_wgrad_dx_adam_bf16 at the benched size for several allocation strategies inside one process.
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


def run(label, w, ea, es, sh):
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(m, k, device=dev, generator=g).clamp_min(0).to(torch.bfloat16)
    dy = torch.randn(m, n, device=dev, generator=g)
    y = torch.rand(m, n, device=dev, generator=g)
    w.normal_(0, 0.01); ea.zero_(); es.zero_(); sh.copy_(w.to(torch.bfloat16))
    step = [0]

    def f():
        step[0] += 1
        K.linear_wgrad_dx_adam_bf16(x, dy, y, w, ea, es, sh, step[0], need_dx=True, need_db=True, gate_dx_by_x=True)

    ts = [timeit(f) for _ in range(3)]
    print(f"{label:58s} " + "  ".join(f"{t:6.1f}" for t in ts) + " us   ptrs " +
          " ".join(hex(t.data_ptr() >> 21) for t in (w, ea, es, sh)))


def separate():
    return (torch.empty(n, k, device=dev), torch.empty(n, k, device=dev), torch.empty(n, k, device=dev),
            torch.empty(n, k, device=dev, dtype=torch.bfloat16))


def slab():
    per = ((n * k * 4 + (1 << 21) - 1) >> 21) << 21
    buf = torch.empty(3 * per + per // 2 + (1 << 21), dtype=torch.uint8, device=dev)
    off = (-buf.data_ptr()) % (1 << 21)
    views = [buf[off + i * per: off + i * per + n * k * 4].view(torch.float32).view(n, k) for i in range(3)]
    sh = buf[off + 3 * per: off + 3 * per + n * k * 2].view(torch.bfloat16).view(n, k)
    return (*views, sh), buf


def slab_skew(delta):
    """w, exp_avg, exp_avg_sq, shadow inside one allocation, consecutive pieces `delta` bytes further apart than their size."""
    sz = n * k * 4
    buf = torch.empty(3 * (sz + delta) + sz // 2 + (1 << 22), dtype=torch.uint8, device=dev)
    off = (-buf.data_ptr()) % (1 << 21)
    views = [buf[off + i * (sz + delta): off + i * (sz + delta) + sz].view(torch.float32).view(n, k) for i in range(3)]
    o3 = off + 3 * (sz + delta)
    sh = buf[o3: o3 + sz // 2].view(torch.bfloat16).view(n, k)
    return (*views, sh), buf


# separate allocations (what the application does), row-major vs tiled moment arrays, alternating in one process
import os as _os
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(m, k, device=dev, generator=g).clamp_min(0).to(torch.bfloat16)
dy = torch.randn(m, n, device=dev, generator=g)
y = torch.rand(m, n, device=dev, generator=g)
for rep in range(4):
    w, ea, es, sh = separate()
    w.normal_(0, 0.01); ea.zero_(); es.zero_(); sh.copy_(w.to(torch.bfloat16))
    st = [0]

    def f():
        st[0] += 1
        K.linear_wgrad_dx_adam_bf16(x, dy, y, w, ea, es, sh, st[0], lr=0.0, need_dx=True, need_db=True, gate_dx_by_x=True)

    out = []
    for tiled in (0, 1, 0, 1):
        if tiled:
            _os.environ["PV_FC1_TILED_MV"] = "1"
        else:
            _os.environ.pop("PV_FC1_TILED_MV", None)
        timeit(f, 4)
        out.append(timeit(f, 10))
    print(f"alloc {rep}: row-major {out[0]:6.1f} {out[2]:6.1f}   tiled {out[1]:6.1f} {out[3]:6.1f} us")
    del w, ea, es, sh
    junk = torch.empty((rep + 1) * 37 * (1 << 20) + 4096 * 13, dtype=torch.uint8, device=dev)   # shifts the next allocations
