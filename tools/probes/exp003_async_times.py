"""experiments/003 train steps enqueued without host synchronisation (as Trainer.fit runs them): per-step device time from
events and the host's enqueue time per step -- a stall of the device shows as a long event interval, one of the host as a long
enqueue."""
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
for _ in range(20): step()
torch.cuda.synchronize()
n = 80
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
host = []
ev[0].record()
for i in range(n):
    t0 = time.perf_counter()
    step()
    ev[i + 1].record()
    host.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
devt = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print("device ms per step:", " ".join(f"{t:.1f}" for t in devt))
print("host enqueue ms   :", " ".join(f"{t:.1f}" for t in host))
