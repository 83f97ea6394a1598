"""Same-process A/B of the experiments/003 train step for a module-level switch:
   python tools/probes/ab_exp003.py perceiver_core.SPLIT_CONTEXT        (models/perceiver/perceiver_core.py)
   python tools/probes/ab_exp003.py PF.ONE_PASS_CONTEXT_BACKWARD        (perceiver_functional.py)
Alternates switch = True / False three times, two warm-up + six timed steps each (the first timing of a process also pays the
workspace allocations: read the later lines)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from predict_pv_yield_amd import perceiver_functional as PF
from predict_pv_yield_amd.models.perceiver import perceiver_core
from predict_pv_yield_amd.models.perceiver.exp003 import LitModel, make_fake_exp003_batch

where, attr = sys.argv[1].split(".")
mod = {"PF": PF, "perceiver_core": perceiver_core}[where]
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
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(3):
    for value in (True, False):
        setattr(mod, attr, value)
        print(f"{sys.argv[1]} = {value}: {timed():.2f} ms/step", flush=True)
setattr(mod, attr, True)
