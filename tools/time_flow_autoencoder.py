"""Times the nb-13 path on the MI355X (SURVEY.md §8a a-10..a-16 at the notebook's own sizes):
  1. super batch: 49 consecutive 704x548 HRV images -> 48 Farnebäck fields (2 coarse levels) -> normalise -> 1176 warps
  2. LitAutoEncoder train step (fwd + MSE + bwd + Adam), B = 64, [B,2,5,128,128] -> [B,1,1,64,64]
and, with --cpu, the torch-CPU oracle's train step on the host cores for the same batch."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from predict_pv_yield_amd.data import flow_examples as fe
from predict_pv_yield_amd.models.conv3d import flow_autoencoder as fa

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=49)
ap.add_argument("--height", type=int, default=704)
ap.add_argument("--width", type=int, default=548)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--cpu", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")

g = torch.Generator().manual_seed(1234)
# smooth random texture translated by ~1 px / frame so the warps stay mostly inside the image
base = torch.nn.functional.avg_pool2d(torch.rand(1, 1, args.height + 128, args.width + 128, generator=g), 9, 1, 4)[0, 0]
base = (base - base.min()) / (base.max() - base.min()) * 1000.0
raw = torch.stack([base[64 + i:64 + i + args.height, 64 - i // 2:64 - i // 2 + args.width] for i in range(args.frames)])
raw = raw.round().to(torch.int16).to(dev)

sb = fe.load_super_batch(raw)
torch.cuda.synchronize()
t0 = time.perf_counter()
n_rep = 3
for _ in range(n_rep):
    sb = fe.load_super_batch(raw)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n_rep
n_pairs, n_pred = args.frames - 1, (args.frames - 1) * args.frames // 2
print(f"super batch {args.frames}x{args.height}x{args.width}: {dt * 1e3:.1f} ms  ({n_pairs} Farneback pairs, {n_pred} warps)"
      f"  -> {dt / n_pairs * 1e3:.2f} ms per pair incl. its warps")
from predict_pv_yield_amd import optical_flow as of
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n_rep):
    fl = of.compute_optical_flow(raw)
torch.cuda.synchronize(); dtf = (time.perf_counter() - t0) / n_rep
print(f"  Farneback only: {dtf * 1e3:.1f} ms = {dtf / n_pairs * 1e3:.3f} ms/pair ({args.height * args.width * n_pairs / dtf / 1e6:.0f} Mpx/s)")

rng = np.random.default_rng(42)
t0 = time.perf_counter()
batch = fe.collate([fe.super_batch_to_example(sb, rng=rng) for _ in range(args.batch)])
torch.cuda.synchronize()
print(f"sampling {args.batch} examples: {(time.perf_counter() - t0) * 1e3:.1f} ms")

torch.manual_seed(0)
model = fa.LitAutoEncoder().to(dev)
opt = model.configure_optimizers()


def step():
    opt.zero_grad()
    loss = model.training_step(batch, 0)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
flop = 3 * 2 * 1.097e9 * args.batch
print(f"LitAutoEncoder train step B={args.batch}: {dt * 1e3:.2f} ms -> {args.batch / dt:.0f} samples/s, "
      f"{flop / dt / 1e12:.1f} TFLOP/s f32 (loss {float(loss):.4f})")

if args.cpu:
    from oracle import conv3d_oracle as co
    cpu_model = co.OracleLitAutoEncoder()
    cpu_model.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    cb = {k: v.cpu() for k, v in batch.items()}
    n = max(1, min(16, args.batch))
    a = (cb[fa.HISTORICAL_SAT_IMAGES][:n], cb[fa.OPTICAL_FLOW_PREDICTIONS][:n], cb[fa.FORECAST_HORIZON][:n],
         cb[fa.TARGET_SAT_IMAGE][:n])
    cpu_model.train_steps(*a, n_steps=1)
    t0 = time.perf_counter()
    cpu_model.train_steps(*a, n_steps=2)
    dtc = (time.perf_counter() - t0) / 2
    print(f"torch-CPU oracle train step B={n} on {torch.get_num_threads()} threads: {dtc * 1e3:.0f} ms -> {n / dtc:.1f} samples/s")
