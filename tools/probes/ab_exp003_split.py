import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from predict_pv_yield_amd.models.perceiver import perceiver_core
from predict_pv_yield_amd import perceiver_functional as PF
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
def timed(n=6):
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    for stored in (True, False):  # here: norm_context inside the attention node
        perceiver_core.SPLIT_CONTEXT = stored
        print(f"context kept as two tensors = {stored}: {timed():.2f} ms/step", flush=True)
