"""Where the conv kernels' main loops spend their cycles: runs one launch of the DIAGNOSTIC library (make -C
predict_pv_yield_amd/csrc diag -> lib/libpvyield_diag.so: the product sources with s_memtime stamps around the phases,
-DPV_DIAG_STAMPS) and prints the per-phase shares, averaged over waves.  Shares, not lengths: the stamps' fences forbid
overlaps the product kernel has.  Usage: python tools/diag_stamps.py [wgrad|wgrad16|dgrad|fwd|first] [batch]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PV_YIELD_LIB"] = os.environ.get("PV_DIAG_LIB") or os.path.join(ROOT, "predict_pv_yield_amd", "lib", "libpvyield_diag.so")
import numpy as np
import torch
from predict_pv_yield_amd import hip_ops as K
from predict_pv_yield_amd._lib import get_lib

which = sys.argv[1] if len(sys.argv) > 1 else "wgrad"
b = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
SLOTS, WAVES = 8, 1 << 14
LABELS = {
    "wgrad": ("pv_diag_read_wgrad2", ["barrier (loaders / other waves)", "16 k-steps (MFMA)", "items"]),
    "v3": ("pv_diag_read_v3", ["vmcnt(0)", "barrier", "issue loads/stores", "MFMA kw=0", "(unused)", "gate request + MFMA kw=1,2",
                               "convert + swap + gate", "steps"]),
    "first": ("pv_diag_read_first", ["barrier (waiting for the loaders)", "MFMA groups", "epilogue", "slices"]),
    "v1": ("pv_diag_read_v1", ["wait + convert + LDS writes", "barrier", "issue next loads", "xp copy", "MFMA groups",
                               "epilogue", "barrier", "slices"]),
}


def run():
    g = torch.Generator(device=dev).manual_seed(1)
    if which in ("wgrad", "wgrad16"):
        ci, t, h = (32, 16, 62) if which == "wgrad" else (11, 18, 64)
        x = torch.randn(b, t, h, h, K.bf16_cpad(ci), device=dev, generator=g).to(torch.bfloat16)
        dy = torch.randn(b, t - 2, h - 2, h - 2, 32, device=dev, generator=g).to(torch.bfloat16)
        fn = lambda: K.conv3d_bwd_weight_bf16(x, dy, None, ci, 32, (0, 0, 0))
        return fn, "wgrad"
    if which in ("fwd", "dgrad"):
        w = torch.randn(32, 32, 3, 3, 3, device=dev, generator=g) * 0.05
        if which == "fwd":
            x = torch.randn(b, 16, 62, 62, 32, device=dev, generator=g).to(torch.bfloat16)
            wp = K.conv3d_pack_weight_bf16(w, transpose_flip=False)
            bias = torch.zeros(32, device=dev)
            fn = lambda: K.conv3d_fwd_bf16(x, None, wp, bias, 32, 32, (0, 0, 0), True, False)
        else:
            dy = torch.randn(b, 14, 60, 60, 32, device=dev, generator=g).to(torch.bfloat16)
            gate = torch.randn(b, 16, 62, 62, 32, device=dev, generator=g).relu().to(torch.bfloat16)
            wpt = K.conv3d_pack_weight_bf16(w, transpose_flip=True)
            fn = lambda: K.conv3d_fwd_bf16(dy, None, wpt, None, 32, 32, (2, 2, 2), relu=False, y_ncdhw=False, out_gate=gate)
        return fn, "v3"
    if which == "first":
        w = torch.randn(32, 11, 3, 3, 3, device=dev, generator=g) * 0.05
        x = torch.randn(b, 11, 18, 64, 64, device=dev, generator=g)
        wp = K.conv3d_pack_weight_bf16(w, transpose_flip=False)
        bias = torch.zeros(32, device=dev)
        fn = lambda: K.conv3d_fwd_bf16_f32in(x, wp, bias, 32, (0, 0, 0), True, want_packed=True)
        return fn, "first"
    if which == "flow":
        from predict_pv_yield_amd import optical_flow as of
        from predict_pv_yield_amd.data.synthetic import advected_counts
        raw = torch.from_numpy(advected_counts(batch=b, seed=1234)[0]).to(dev)      # the SURVEY section-8(d) input
        return (lambda: of.advect_future_frames(raw, 6)), "flow"
    raise SystemExit(__doc__)


def flow_report(d):
    """fb_level_u_kernel (8 uniform waves per workgroup; waves 0..3 = group 0: three channels, waves 4..7 = group 1: two
    channels; half of the solve each); the last launch's stamps"""
    d = d[: (d[:, 7] > 0).nonzero()[0].max() + 1]
    if True:
        d = d[: len(d) // 8 * 8].reshape(-1, 8, SLOTS)
        if os.environ.get("PV_DIAG_BARRIER_DETAIL"):      # a library built with -DFBU_BARRIER_DETAIL
            labels = ["wait at Bm", "wait at B0", "wait at B1", "wait at B2", "wait at B3", "wait at B4", "everything else"]
        else:
          labels = ["G: gathers + blend", "F: rest of UpdateMatrices + publish max", "scale, split + write channels (+ R1 fetch issue)",
                  "products (+ split + write of the next channels, mailbox write)", "mailbox read + solve + flow to LDS / memory",
                  "flow from LDS / next unit's flow + R0 requests", "barrier waits"]
        for name, sl in (("group 0", slice(0, 4)), ("group 1", slice(4, 8))):
            w = d[:, sl].reshape(-1, SLOTS)
            w = w[w[:, 7] > 0]
            tot = w[:, :7].sum(1).mean()
            print(f"  {name}: {len(w)} waves, {w[:, 7].mean():.1f} stages per wave, {tot / w[:, 7].mean():.0f} cycles / stage")
            for i, lab in enumerate(labels):
                print(f"    {lab:64s} {(w[:, i] / w[:, 7]).mean():9.0f} cycles / stage  {100 * w[:, i].mean() / tot:5.1f} %")
        return


fn, kind = run()
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
fn()
e1.record()
torch.cuda.synchronize()
sym, labels = LABELS.get(kind, ("pv_diag_read_fb_iter", None))
buf = np.zeros(WAVES * SLOTS, dtype=np.uint64)
lib = get_lib()
f = getattr(lib, sym)
f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert f(buf.ctypes.data, buf.size) == 0
d = buf.reshape(WAVES, SLOTS).astype(np.float64)
if kind == "flow":
    print(f"flow B={b}: stamped pipeline {e0.elapsed_time(e1) * 1e3:.1f} us")
    flow_report(d)
    # fb_prep_polyexp_mfma_kernel (the last launch = level 0): 8 waves per workgroup
    f2 = lib.pv_diag_read_fb_polyexp
    f2.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    buf2 = np.zeros(WAVES * SLOTS, dtype=np.uint64)
    assert f2(buf2.ctypes.data, buf2.size) == 0
    p = buf2.reshape(WAVES, SLOTS).astype(np.float64)
    grp_of_row = (np.arange(len(p)) % 8) >> 2          # waves 0..3 = group 0 (b1, b2, b4 + the pair stores), 4..7 = group 1
    live = p[:, 7] > 0
    for gsel in (0, 1):
        pg = p[live & (grp_of_row == gsel)]
        if len(pg):
            print(f"  PolyExp level 0, group {gsel}: " + ", ".join(f"{lab.split(':')[0][:24]} {(pg[:, i] / pg[:, 7]).mean():.0f}"
                                                                 for i, lab in enumerate(["source", "barrier waits", "filters + X write", "products", "epilogue"])))
    p = p[live]
    if len(p):
        labels = ["source words -> floats in LDS, next image requested", "barrier waits", "3-tap filters, resize, split, X write",
                  "products (+ mailbox write)", "epilogue: coefficients, stores"]
        tot = p[:, :5].sum(1).mean()
        print(f"  PolyExp (matrix cores), level 0: {len(p)} waves, {p[:, 7].mean():.1f} images per wave, {tot / p[:, 7].mean():.0f} cycles / image")
        for i, lab in enumerate(labels):
            print(f"    {lab:64s} {(p[:, i] / p[:, 7]).mean():9.0f} cycles / image  {100 * p[:, i].mean() / tot:5.1f} %")
    raise SystemExit(0)
count_col = len(labels) - 1
live = d[:, count_col] > 0
d = d[live]
print(f"{which} B={b}: stamped launch {e0.elapsed_time(e1) * 1e3:.1f} us, {live.sum()} waves, "
      f"{d[:, count_col].mean():.1f} iterations per wave")
tot = d[:, :count_col].sum(1).mean()
for i, lab in enumerate(labels[:count_col]):
    per_it = (d[:, i] / d[:, count_col]).mean()
    print(f"  {lab:32s} {per_it:9.0f} cycles / iteration  {100 * d[:, i].mean() / tot:5.1f} %")
print(f"  {'sum':32s} {tot / d[:, count_col].mean():9.0f} cycles / iteration")
