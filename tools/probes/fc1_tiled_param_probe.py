"""Timing probe (results of the probed arm are garbage by construction): what the fc1 forward and the one-pass fc1 backward
would gain if the f32 master and its bf16 operand copy were stored tile by tile ([K/128][N][128]) like the two moment arrays.
PV_FC1_PROBE_TILED_PARAM=1 makes both kernels ADDRESS them that way (same bytes, same instruction stream, other addresses).
   python tools/probes/fc1_tiled_param_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from predict_pv_yield_amd import hip_ops as K

dev = torch.device("cuda:0")
m, n, k = 32, 128, 1003520
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(m, k, device=dev, generator=g).clamp_min(0).to(torch.bfloat16)
dy = torch.randn(m, n, device=dev, generator=g)
y = torch.rand(m, n, device=dev, generator=g)
bias = torch.zeros(n, device=dev)


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


for rep in range(3):
    w = torch.empty(n, k, device=dev).normal_(0, 0.01)
    ea, es = torch.zeros(n, k, device=dev), torch.zeros(n, k, device=dev)
    sh = w.to(torch.bfloat16)
    st = [0]

    def bwd():
        st[0] += 1
        K.linear_wgrad_dx_adam_bf16(x, dy, y, w, ea, es, sh, st[0], lr=0.0, need_dx=True, need_db=True, gate_dx_by_x=True,
                                    moments_tiled=True)

    def fwd():
        K.linear_fwd_bf16(x, sh, bias, True)

    out = {}
    for tiled in (False, True, False, True):
        if tiled:
            os.environ["PV_FC1_PROBE_TILED_PARAM"] = "1"
        else:
            os.environ.pop("PV_FC1_PROBE_TILED_PARAM", None)
        out.setdefault(tiled, []).append((timeit(fwd, 20), timeit(bwd, 10)))
    os.environ.pop("PV_FC1_PROBE_TILED_PARAM", None)
    print(f"alloc {rep}: row-major fwd/bwd {out[False]}   tiled-probe fwd/bwd {out[True]}", flush=True)
    del w, ea, es, sh
    junk = torch.empty((rep + 1) * 37 * (1 << 20) + 4096 * 13, dtype=torch.uint8, device=dev)
