import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from predict_pv_yield_amd import hip_ops as K
from predict_pv_yield_amd._lib import get_lib, ptr, check, current_stream_ptr
dev = torch.device("cuda:0")
lib = get_lib()
vps, b = 64, 2
one = torch.tensor([0.0, 1.0, 1.0], device=dev)
missed = []
for v in range(vps):
    for c in range(32):
        parts = torch.zeros((3, b, vps, 32), device=dev)
        parts[2, 1, v, c] = 3.0
        parts[2, 0, (v + 5) % vps, (c + 3) % 32] = 2.0
        y = torch.empty((b, 32, vps), device=dev)
        st = torch.zeros(3, device=dev)
        check(lib.pv_sum3_ndhwc_to_ncdhw_f32(ptr(parts), ptr(one), ptr(one), None, None, ptr(y), ptr(st), 1, b, vps, current_stream_ptr()), "sum3")
        if float(st[0]) != 3.0 or float(y[1, c, v]) != 3.0:
            missed.append((v, c, float(st[0]), float(y[1, c, v])))
print("missed", len(missed), missed[:40])
