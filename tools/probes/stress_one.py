import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from predict_pv_yield_amd import hip_ops as K
from oracle import flow_oracle as fo
from predict_pv_yield_amd.data.synthetic import advected_counts
dev = torch.device("cuda:0")
h, w, t, batch, seed = 64, 64, 3, 300, 11
raw, _ = advected_counts(batch=batch, t=t, channels=2, h=h, w=w, seed=seed)
stacks = np.ascontiguousarray(raw.transpose(0, 2, 1, 3, 4)).reshape(2 * batch, t, h, w)
u8 = torch.from_numpy(fo.convert_10bpp_to_uint8(stacks, 0)[0]).to(dev)
first = K.farneback_stack(u8)
bad = 0
for rep in range(30):
    got = K.farneback_stack(u8)
    if not torch.equal(got, first):
        bad += 1
print(os.environ.get("PV_YIELD_LIB", "default"), os.environ.get("PV_X",""), "bad runs:", bad, "of 30")
