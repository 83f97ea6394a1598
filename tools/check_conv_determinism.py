"""Repeated launches of the benched conv kernels (weight gradient with its slab reduce, forward) must give identical bits."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from predict_pv_yield_amd import hip_ops as K
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
bad = 0
for (ci, t, h) in [(32, 16, 62), (32, 12, 58), (11, 18, 64)]:
    x = torch.randn(32, t, h, h, K.bf16_cpad(ci), device=dev, generator=g).to(torch.bfloat16)
    dy = torch.randn(32, t - 2, h - 2, h - 2, 32, device=dev, generator=g).to(torch.bfloat16)
    ref = None
    for rep in range(6):
        dw, db = K.conv3d_bwd_weight_bf16(x, dy, None, ci, 32, (0, 0, 0))
        if ref is None: ref = (dw.clone(), db.clone())
        elif not (torch.equal(dw, ref[0]) and torch.equal(db, ref[1])): bad += 1
    w = torch.randn(32, 32, 3, 3, 3, device=dev, generator=g) * 0.05
    if ci == 32:
        wp = K.conv3d_pack_weight_bf16(w)
        ys = [K.conv3d_fwd_bf16(x, None, wp, None, 32, 32, (0, 0, 0), True, False) for _ in range(4)]
        bad += sum(not torch.equal(ys[0], y) for y in ys[1:])
print("non-deterministic repeats:", bad)
