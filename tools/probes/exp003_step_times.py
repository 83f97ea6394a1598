"""Per-step wall times of the experiments/003 train step (60 steps after 5 warm-up): looks for periodic stalls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from predict_pv_yield_amd.models.perceiver.exp003 import LitModel, make_fake_exp003_batch
dev = torch.device("cuda:0")
batch = {k: v.to(dev) for k, v in make_fake_exp003_batch(8, 128, torch.Generator().manual_seed(1)).items()}
torch.manual_seed(0)
model = LitModel(operand_dtype="bf16").to(dev)
opt = model.configure_optimizers()
def step():
    opt.zero_grad(set_to_none=True)
    model.training_step(batch, 0).backward()
    opt.step()
for _ in range(5): step()
ts = []
for i in range(60):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join(f"{t:.1f}" for t in ts))
import gc
print("gc counts", gc.get_count(), "gc stats collections", [s["collections"] for s in gc.get_stats()])
