"""Same-process A/B of the PerceiverModel train step: toggles a module attribute (e.g. perceiver_core.FUSE_RESIDUALS,
perceiver_functional.ACCUMULATE_TIED_GRADS) between rounds and prints the step time of each setting.
   python tools/ab_perceiver.py perceiver_core.FUSE_RESIDUALS"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predict_pv_yield_amd.data.fake import FakeDataConfiguration, make_fake_batch
from predict_pv_yield_amd.models.perceiver.perceiver import PerceiverModel

mod_name, attr = sys.argv[1].rsplit(".", 1)
mod = importlib.import_module("predict_pv_yield_amd.models.perceiver." + mod_name if mod_name == "perceiver_core"
                              else "predict_pv_yield_amd." + mod_name)
dev = torch.device("cuda:0")
kw = dict(history_minutes=60, forecast_minutes=30, batch_size=8, num_latents=128, latent_dim=64, embedding_dem=16, output_variable="gsp_yield")
torch.manual_seed(0)
model = PerceiverModel(**kw, operand_dtype="bf16").to(dev)
cfg = FakeDataConfiguration(batch_size=8, history_minutes=60, forecast_minutes=30, satellite_image_size_pixels=64, nwp_image_size_pixels=64)
batch = make_fake_batch(cfg, torch.Generator().manual_seed(1)).to(dev)
opt = model.configure_optimizers()


def steps(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        opt.zero_grad(set_to_none=True)
        model.training_step(batch, 0).backward()
        opt.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


acc = {True: [], False: []}
for rnd in range(4):
    for val in (True, False):
        setattr(mod, attr, val)
        steps(2)
        acc[val].append(steps(5))
for val in (True, False):
    print(f"{sys.argv[1]} = {val}: {sum(acc[val]) / len(acc[val]):.2f} ms/step  (rounds {[round(v, 2) for v in acc[val]]})")
