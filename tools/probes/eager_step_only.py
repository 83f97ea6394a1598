"""The headline step run eagerly 60 times (counterpart of graph_replay_only.py for a same-box kernel table)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from predict_pv_yield_amd.models.conv3d.model import Model
dev = torch.device("cuda:0")
torch.manual_seed(518)
model = Model(**bench.MODEL_KW, history_minutes=55, precision="bf16").to(dev)
model.batch_size = 32
g = torch.Generator(device=dev).manual_seed(518)
batch = {"satellite": {"data": torch.randn(32, 11, 18, 64, 64, generator=g, device=dev)},
         "pv": {"pv_yield": torch.rand(32, 18, 128, generator=g, device=dev)}}
opt = model.configure_optimizers()


def step():
    opt.zero_grad(set_to_none=True)
    model.training_step(batch, 0).backward()
    opt.step()


for _ in range(8):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(60):
    step()
torch.cuda.synchronize()
print(f"eager {(time.perf_counter() - t0) / 60 * 1e3:.3f} ms per step")
