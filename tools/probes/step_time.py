"""ms per headline train step (B = 32, T = 18, bf16) in this process: for same-box comparisons of two builds of the library
(PV_YIELD_LIB=... selects one) or of a module switch (PV_AB_SET=functional.USE_RELU_MASKS=1).  Median of 8 rounds of 40 steps."""
import importlib
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from predict_pv_yield_amd.models.conv3d.model import Model

for item in filter(None, os.environ.get("PV_AB_SET", "").split(",")):
    name, val = item.rsplit("=", 1)
    mod, attr = name.rsplit(".", 1)
    setattr(importlib.import_module("predict_pv_yield_amd." + mod), attr, bool(int(val)))
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
torch.manual_seed(518)
model = Model(include_pv_yield=False, include_nwp=False, forecast_minutes=30, history_minutes=55, number_sat_channels=11).to(dev)
model.batch_size = max(32, b)
opt = model.configure_optimizers()
g = torch.Generator(device=dev).manual_seed(1)
batch = {"satellite": {"data": torch.randn(b, 11, 18, 64, 64, generator=g, device=dev)},
         "pv": {"pv_yield": torch.rand(b, 18, 128, generator=g, device=dev)}}


def step():
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(batch, 0)
    loss.backward()
    opt.step()
    return loss


for _ in range(20):
    step()
ts = []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        last = step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 40 * 1e3)
print(f"{os.environ.get('PV_YIELD_LIB', 'default lib').split('/')[-1]:24s} {os.environ.get('PV_AB_SET', ''):36s} "
      f"median {statistics.median(ts):.4f} ms  min {min(ts):.4f}  loss {float(last):.6f}")
