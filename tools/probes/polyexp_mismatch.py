import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from predict_pv_yield_amd import hip_ops as K
dev = torch.device("cuda:0")
rng = np.random.default_rng(7)
for case in range(400):
    h, w = int(rng.integers(3, 17)) * 4, int(rng.integers(3, 17)) * 4
    if rng.random() < 0.2:
        w += int(rng.integers(1, 4))
    t = int(rng.integers(2, 7))
    stacks = int(rng.integers(1, 90))
    kw = dict(levels=int(rng.integers(1, 4)), winsize=int(rng.choice([5, 9, 15, 21])), iterations=int(rng.integers(1, 4)),
              poly_n=int(rng.choice([5, 7])), poly_sigma=float(rng.choice([1.1, 1.5])))
    base = rng.integers(0, 256, (stacks, 1, h + 8, w + 8), dtype=np.uint8)
    frames = np.stack([np.roll(base[:, 0], (i, 2 * i), axis=(1, 2))[:, 4:4 + h, 4:4 + w] for i in range(t)], axis=1)
    frames = np.ascontiguousarray((frames.astype(np.int16) + rng.integers(0, 6, frames.shape)).clip(0, 255).astype(np.uint8))
    u8 = torch.from_numpy(frames).to(dev)
    mfma = K.farneback_stack(u8, **kw)
    os.environ["PV_FARNEBACK_POLYEXP_VALU"] = "1"
    valu = K.farneback_stack(u8, **kw)
    del os.environ["PV_FARNEBACK_POLYEXP_VALU"]
    d = (mfma - valu).abs()
    if float(d.max()) > 2e-3 * max(1.0, float(valu.abs().max())):
        per_pair = d.reshape(stacks * (t - 1), -1).max(1).values
        bad = (per_pair > 1e-3).nonzero().flatten().tolist()
        idx = (d == d.max()).nonzero()[0].tolist()
        print("case", case, (h, w, t, stacks), kw, "max", float(d.max()), "bad pairs", len(bad), "of", len(per_pair), bad[:12], "argmax", idx, flush=True)
        for it in (1,):
            kw1 = dict(kw, iterations=1, levels=1)
            m1 = K.farneback_stack(u8, **kw1)
            os.environ["PV_FARNEBACK_POLYEXP_VALU"] = "1"
            v1 = K.farneback_stack(u8, **kw1)
            del os.environ["PV_FARNEBACK_POLYEXP_VALU"]
            print("   levels=1 iterations=1:", float((m1 - v1).abs().max()), "max flow", float(v1.abs().max()))
