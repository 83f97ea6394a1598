"""experiments/003 (default) or PerceiverModel (`perceiver` as the first argument) train step as a HIP graph
(graphs.GraphedTrainStep): losses against the eager step, ms per step of both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from predict_pv_yield_amd.graphs import GraphedTrainStep
from predict_pv_yield_amd.optim import HipAdam
from predict_pv_yield_amd.models.perceiver.exp003 import LitModel, make_fake_exp003_batch
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "exp003"
if kind == "exp003":
    batches = [{k: v.to(dev) for k, v in make_fake_exp003_batch(8, 128, torch.Generator().manual_seed(s)).items()} for s in range(4)]
else:
    from predict_pv_yield_amd.data.fake import FakeDataConfiguration, make_fake_batch
    from predict_pv_yield_amd.models.perceiver.perceiver import PerceiverModel
    cfg = FakeDataConfiguration(batch_size=8, history_minutes=60, forecast_minutes=30, satellite_image_size_pixels=64,
                                nwp_image_size_pixels=64)
    batches = [make_fake_batch(cfg, torch.Generator().manual_seed(s)).to(dev) for s in range(4)]

def make(capturable):
    torch.manual_seed(0)
    if kind == "exp003":
        model = LitModel(operand_dtype="bf16").to(dev)
    else:
        model = PerceiverModel(history_minutes=60, forecast_minutes=30, batch_size=8, num_latents=128, latent_dim=64,
                               embedding_dem=16, output_variable="gsp_yield", operand_dtype=kind.split(":")[1] if ":" in kind else "f32").to(dev)
    ref_opt = model.configure_optimizers()
    g = ref_opt.param_groups[0]
    opt = HipAdam(model.parameters(), lr=g["lr"], betas=g["betas"], eps=g["eps"],
                  capturable=capturable) if capturable else ref_opt
    return model, opt

model_e, opt_e = make(False)
def eager(b):
    opt_e.zero_grad(set_to_none=True)
    loss = model_e.training_step(b, 0)
    loss.backward()
    opt_e.step()
    return loss.detach()
model_g, opt_g = make(True)
step = GraphedTrainStep(model_g, opt_g, batches[0], warmup=3)
for _ in range(3):
    eager(batches[0])          # the same three warm-up steps on the eager side
le, lg = [], []
for i in range(8):
    le.append(float(eager(batches[i % 4])))
    lg.append(float(step(batches[i % 4])))
print("eager losses :", " ".join(f"{v:.6f}" for v in le))
print("graph losses :", " ".join(f"{v:.6f}" for v in lg))
def timed(f, n=20):
    for _ in range(3): f(batches[0])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): f(batches[i % 4])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    print(f"eager {timed(eager):.2f} ms/step   graph {timed(step):.2f} ms/step")
