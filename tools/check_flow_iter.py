"""The flow path (matrix-core PolyExp + fb_level_u_kernel) against the oracle, the vector-ALU PolyExp and the two-launch form: largest flow differences (px) over a few shapes, and
stack == pairs bit-identity.  Usage: python tools/check_flow_iter.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import flow_oracle as fo
from predict_pv_yield_amd import hip_ops as K
from predict_pv_yield_amd.data.synthetic import advected_counts

dev = torch.device("cuda:0")


def run(env, fn):
    for k in env:
        os.environ[k] = "1"
    try:
        return fn()
    finally:
        for k in env:
            del os.environ[k]


for h, w, t, batch in [(64, 64, 12, 3), (64, 64, 4, 100), (40, 56, 3, 3), (48, 64, 4, 3), (64, 36, 2, 3), (32, 48, 3, 2), (24, 28, 3, 5)]:
    raw, _ = advected_counts(batch=batch, t=t, channels=2, h=h, w=w, seed=3 * h + t)
    stacks = np.ascontiguousarray(raw.transpose(0, 2, 1, 3, 4)).reshape(2 * batch, t, h, w)
    u8 = torch.from_numpy(fo.convert_10bpp_to_uint8(stacks, 0)[0]).to(dev)
    prev = u8[:, :-1].reshape(-1, h, w).contiguous()
    nxt = u8[:, 1:].reshape(-1, h, w).contiguous()
    new_stack, new_pairs = K.farneback_stack(u8), K.farneback_pairs(prev, nxt)
    two = run(["PV_FARNEBACK_TWO_LAUNCH_ITERATION"], lambda: K.farneback_stack(u8))
    valu = run(["PV_FARNEBACK_POLYEXP_VALU"], lambda: K.farneback_stack(u8))
    torch.cuda.synchronize()
    errs = []
    for (i, j) in [(0, 0), (2, 0), (2 * batch - 1, t - 2)]:
        ref = fo.calc_optical_flow_farneback(u8[i, j].cpu().numpy(), u8[i, j + 1].cpu().numpy())
        errs.append(float(np.abs(new_stack[i, j].cpu().numpy() - ref).max()))
    print(f"{h}x{w} t={t} stacks={2 * batch}: finite={bool(torch.isfinite(new_stack).all())} "
          f"stack==pairs {torch.equal(new_stack.reshape(-1, h, w, 2), new_pairs)}  "
          f"|matrix-core - vector-ALU PolyExp|max {float((new_stack - valu).abs().max()):.1e}  "
          f"|new-two|max {float((new_stack - two).abs().max()):.3e}  "
          f"|new-oracle|max {max(errs):.3e}  max|flow| {float(two.abs().max()):.2f}", flush=True)
