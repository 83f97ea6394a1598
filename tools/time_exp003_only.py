"""Train-step time of exp003.LitModel alone (bf16 attention operands): target for rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predict_pv_yield_amd.models.perceiver.exp003 import LitModel, make_fake_exp003_batch
dev = torch.device("cuda:0")
b3 = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
batch = {k: v.to(dev) for k, v in make_fake_exp003_batch(b3, 128, torch.Generator().manual_seed(1)).items()}
torch.manual_seed(0)
model = LitModel(operand_dtype="bf16").to(dev)
opt = model.configure_optimizers()


def step():
    opt.zero_grad(set_to_none=True)
    model.training_step(batch, 0).backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
d = (time.perf_counter() - t0) / steps
print(f"exp003.LitModel B={b3} x 19 images 128x128x12, bf16 attention operands: {d * 1e3:.1f} ms/step, {b3 / d:.1f} samples/s")
