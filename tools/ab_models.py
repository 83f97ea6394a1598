"""Same-process A/B of the headline train step (B = 32, T = 18, bf16) between two MODELS built under different module-level
switches read at construction time (tools/ab_step.py flips switches read at call time):
   python tools/ab_models.py models.conv3d.model.FC1_CHANNELS_LAST"""
import importlib
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from predict_pv_yield_amd.models.conv3d.model import Model

switch = sys.argv[1] if len(sys.argv) > 1 else "models.conv3d.model.FC1_CHANNELS_LAST"
b = int(sys.argv[2]) if len(sys.argv) > 2 else 32
mod_name, attr = switch.rsplit(".", 1)
mod = importlib.import_module("predict_pv_yield_amd." + mod_name)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
batch = {"satellite": {"data": torch.randn(b, 11, 18, 64, 64, generator=g, device=dev)},
         "pv": {"pv_yield": torch.rand(b, 18, 128, generator=g, device=dev)}}
arms = {}
for v in (True, False):
    setattr(mod, attr, v)
    torch.manual_seed(518)
    model = Model(include_pv_yield=False, include_nwp=False, forecast_minutes=30, history_minutes=55, number_sat_channels=11).to(dev)
    model.batch_size = max(32, b)
    arms[v] = (model, model.configure_optimizers())


def run(v, n):
    model, opt = arms[v]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        opt.zero_grad(set_to_none=True)
        loss = model.training_step(batch, 0)
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, float(loss)


for v in arms:
    run(v, 5)
res = {v: [] for v in arms}
for r in range(6):
    for v in arms:
        run(v, 2)
        ms, loss = run(v, 20)
        res[v].append(ms)
for v in arms:
    print(f"{switch} = {v}: median {statistics.median(res[v]):.4f} ms/step   all {[round(x, 4) for x in res[v]]}   last loss {run(v, 1)[1]:.5f}")
