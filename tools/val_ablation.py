"""Where does the bf16 path's validation penalty come from?  

bench.py's matched-validation experiment reads bf16 - fp32 = about +1e-3 of the normalised yield after 512 Adam steps (32 seeds,
round 5).  This tool runs, on the SAME seeds / initial weights / batches / held-out samples, besides the two arms of that
experiment three hybrids, and reports every arm's paired difference from the fp32 arm:

  bf16              the benched path
  bf16conv_f32fc1   bf16 conv tower (NCDHW last layer), fc1 forward / dx / weight gradient in f32 on the f32 master weights
  f32conv_bf16fc1   f32-accurate conv tower, fc1 on bf16 operands (bf16 activation, bf16 operand copy of the weight, one-pass backward)
  bf16_f32masks     the benched path with the F32 arithmetic's ReLU masks forced on every conv layer and fc1 (tools/relu_masks.py;
                    the masks come from torch's own f32 operators on the device): operand rounding without a single flipped unit

python tools/val_ablation.py [seeds=8] [steps=512]        (about 25 s per seed)"""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from predict_pv_yield_amd import functional as Fn
from predict_pv_yield_amd.models.conv3d import model as model_mod
from predict_pv_yield_amd.models.conv3d.model import Model
from tools.relu_masks import f32_relu_masks_torch, forced_relu_masks

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 512
batch, n_val, tail, tail_stride = 32, 1024, 64, 4
dev = torch.device("cuda:0")
t_frames = 18
K_FC1 = 32 * 10 * 56 * 56

vgen = torch.Generator(device=dev).manual_seed(77)
val_sat, val_pv = bench.learnable_task_on_device(n_val, t_frames, vgen, dev)
y_val = val_pv[:, -6:, 0]


def build(arm):
    if arm == "bf16conv_f32fc1":
        model_mod.FC1_CHANNELS_LAST = False          # reference column order in memory: the f32 fc1 kernels read it as it is
    try:
        m = Model(**bench.MODEL_KW, history_minutes=55, precision="fp32" if arm in ("fp32", "f32conv_bf16fc1") else "bf16").to(dev)
    finally:
        model_mod.FC1_CHANNELS_LAST = True
    m.batch_size = max(m.batch_size, n_val, batch)
    return m


class patched:
    """The hybrids re-route ONE call of the tower inside this context (fc1's Linear), nothing else."""

    def __init__(self, arm):
        self.arm = arm

    def __enter__(self):
        self.lin_bf16, self.lin_f32 = Fn.linear_bf16, Fn.linear_f32
        if self.arm == "bf16conv_f32fc1":
            Fn.linear_bf16 = lambda x, w, b, relu=False, x_is_relu_output=False: self.lin_f32(x.float(), w, b, relu)
        elif self.arm == "f32conv_bf16fc1":
            def lin(x, w, b, relu=False):
                if x.shape[1] == K_FC1:
                    return self.lin_bf16(x.to(torch.bfloat16), w, b, relu, False)
                return self.lin_f32(x, w, b, relu)
            Fn.linear_f32 = lin
        return self

    def __exit__(self, *exc):
        Fn.linear_bf16, Fn.linear_f32 = self.lin_bf16, self.lin_f32
        return False


def reinitialise(m, seed):
    g = torch.Generator(device=dev).manual_seed(518 + 1000 * seed)
    with torch.no_grad():
        for _, mod in m.named_modules():
            w = getattr(mod, "weight", None)
            if w is None or not isinstance(w, torch.nn.Parameter):
                continue
            bound = 1.0 / w[0].numel() ** 0.5
            w.uniform_(-bound, bound, generator=g)
            if getattr(mod, "bias", None) is not None:
                mod.bias.uniform_(-bound, bound, generator=g)


def batches_of(seed):
    g = torch.Generator(device=dev).manual_seed(100 + 1000 * seed)
    for _ in range(n_steps):
        yield bench.learnable_task_on_device(batch, t_frames, g, dev)


def val_of(m, arm):
    with torch.no_grad():
        ys = []
        for i in range(0, n_val, 64):
            b = {"satellite": {"data": val_sat[i:i + 64]}, "pv": {"pv_yield": val_pv[i:i + 64]}}
            if arm == "bf16_f32masks":
                cm, fm = f32_relu_masks_torch(m, val_sat[i:i + 64])
                with forced_relu_masks(cm, fm):
                    ys.append(m(b))
            else:
                ys.append(m(b))
    return float((torch.cat(ys) - y_val).abs().mean())


ARMS = ("fp32", "bf16", "bf16conv_f32fc1", "f32conv_bf16fc1", "bf16_f32masks")
models = {a: build(a) for a in ARMS}
tail_steps = set(range(n_steps, n_steps - tail, -tail_stride))
runs = {a: [] for a in ARMS}
t0 = time.perf_counter()
for seed in range(seeds):
    reinitialise(models["fp32"], seed)
    init = {k: v.clone() for k, v in models["fp32"].state_dict().items()}
    for arm in ARMS:
        m = models[arm]
        m.load_state_dict(init)
        opt = m.configure_optimizers()
        acc = []
        with patched(arm):
            for i, (sat, pv) in enumerate(batches_of(seed)):
                b = {"satellite": {"data": sat}, "pv": {"pv_yield": pv}}
                opt.zero_grad(set_to_none=True)
                if arm == "bf16_f32masks":
                    cm, fm = f32_relu_masks_torch(m, sat)
                    with forced_relu_masks(cm, fm):
                        loss = m.training_step(b, 0)
                        loss.backward()
                else:
                    loss = m.training_step(b, 0)
                    loss.backward()
                opt.step()
                if i + 1 in tail_steps:
                    acc.append(val_of(m, arm))
        runs[arm].append(sum(acc) / len(acc))
        del opt
        bench.settle()
    print(f"seed {seed}: " + "  ".join(f"{a} {runs[a][-1]:.5f}" for a in ARMS), flush=True)

se = lambda v: statistics.stdev(v) / len(v) ** 0.5 if len(v) > 1 else float("nan")
out = {"seeds": seeds, "steps": n_steps, "seconds": round(time.perf_counter() - t0, 1),
       "definition": f"validation NMAE on {n_val} held-out samples, mean over every {tail_stride}th of the last {tail} of {n_steps} Adam steps at "
                     f"B = {batch}; every arm of a seed shares initial weights and batches; differences are paired, against the fp32 arm",
       "arms": {a: {"mean": round(statistics.fmean(runs[a]), 6), "standard_error": round(se(runs[a]), 6),
                    "runs": [round(v, 5) for v in runs[a]]} for a in ARMS},
       "paired_minus_fp32": {}}
for a in ARMS[1:]:
    d = [x - y for x, y in zip(runs[a], runs["fp32"])]
    out["paired_minus_fp32"][a] = {"mean": round(statistics.fmean(d), 6), "standard_error": round(se(d), 6),
                                   "per_seed": [round(v, 5) for v in d]}
print(json.dumps(out, indent=1))
