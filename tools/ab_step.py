"""Same-process A/B of the headline train step (B = 32, T = 18, bf16): alternates module-level switches and reports
ms/step per setting (median of rounds), so that box-to-box and warm-up drift cancel.
   python tools/ab_step.py functional.USE_RELU_MASKS
   python tools/ab_step.py env:PV_FC1_NT          (an environment switch the library reads at every launch: set / unset)"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib

import torch

from predict_pv_yield_amd.models.conv3d.model import Model

switch = sys.argv[1] if len(sys.argv) > 1 else "functional.USE_RELU_MASKS"
if switch.startswith("env:"):
    class _Env:      # setattr(mod, NAME, True / False) sets / removes the environment variable
        def __setattr__(self, name, value):
            if value:
                os.environ[name] = os.environ.get("AB_ENV_VALUE", "1")
            else:
                os.environ.pop(name, None)
    mod, attr = _Env(), switch[4:]
else:
    mod_name, attr = switch.rsplit(".", 1)
    mod = importlib.import_module("predict_pv_yield_amd." + mod_name)
b = int(sys.argv[2]) if len(sys.argv) > 2 else 32
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
    model.training_step(batch, 0).backward()
    opt.step()


def run(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for v in (True, False):
    setattr(mod, attr, v)
    run(5)
res = {True: [], False: []}
for r in range(6):
    for v in (True, False):
        setattr(mod, attr, v)
        run(2)
        res[v].append(run(20))
for v in (True, False):
    print(f"{switch} = {v}: median {statistics.median(res[v]):.4f} ms/step   all {[round(x, 4) for x in res[v]]}")
