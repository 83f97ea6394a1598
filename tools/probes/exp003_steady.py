import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from predict_pv_yield_amd.models.perceiver import perceiver_core
from predict_pv_yield_amd.models.perceiver.exp003 import LitModel, make_fake_exp003_batch
perceiver_core.SPLIT_CONTEXT = sys.argv[1] == "1"
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
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 10 * 1e3)
print(f"SPLIT_CONTEXT={perceiver_core.SPLIT_CONTEXT}: " + " ".join(f"{t:.2f}" for t in ts) + " ms/step; peak memory %.2f GiB" % (torch.cuda.max_memory_allocated() / 2**30))
