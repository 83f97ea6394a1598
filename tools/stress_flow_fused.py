"""Determinism / race check of the level kernel: repeated runs on many pairs must give identical bits (and stay within
2e-5 px of the two-launch form)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from predict_pv_yield_amd import hip_ops as K
from oracle import flow_oracle as fo
from predict_pv_yield_amd.data.synthetic import advected_counts
dev = torch.device("cuda:0")
bad = 0
for (h, w, t, batch, seed) in [(64, 64, 4, 100, 196), (64, 64, 12, 100, 7), (64, 64, 3, 300, 11), (48, 64, 5, 80, 3)]:
    raw, _ = advected_counts(batch=batch, t=t, channels=2, h=h, w=w, seed=seed)
    stacks = np.ascontiguousarray(raw.transpose(0, 2, 1, 3, 4)).reshape(2 * batch, t, h, w)
    u8 = torch.from_numpy(fo.convert_10bpp_to_uint8(stacks, 0)[0]).to(dev)
    os.environ["PV_FARNEBACK_TWO_LAUNCH_ITERATION"] = "1"
    ref = K.farneback_stack(u8)
    del os.environ["PV_FARNEBACK_TWO_LAUNCH_ITERATION"]
    first = K.farneback_stack(u8)
    # (the two-launch form has the vector-ALU PolyExp only, the level kernels run on the matrix-core one: 1e-4 px between them)
    assert float((first - ref).abs().max()) <= 1e-4, float((first - ref).abs().max())
    ref = first
    for rep in range(25):
        got = K.farneback_stack(u8)
        if not torch.equal(got, ref):
            d = (got != ref)
            bad += 1
            print("MISMATCH", (h, w, t, batch), "rep", rep, int(d.sum()), "elements, max abs", float((got - ref).abs().max()),
                  "first idx", d.nonzero()[0].tolist())
    print("case", (h, w, t, batch), "pairs", 2 * batch * (t - 1), "done")
print("bad runs:", bad)
