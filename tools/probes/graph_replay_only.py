"""The headline step captured as a HIP graph and replayed 60 times (target of tools/probes/graph_vs_eager_kernels.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from predict_pv_yield_amd.graphs import GraphedTrainStep
from predict_pv_yield_amd.models.conv3d.model import Model
from predict_pv_yield_amd.optim import HipAdam
dev = torch.device("cuda:0")
torch.manual_seed(518)
model = Model(**bench.MODEL_KW, history_minutes=55, precision="bf16").to(dev)
model.batch_size = 32
g = torch.Generator(device=dev).manual_seed(518)
batch = {"satellite": {"data": torch.randn(32, 11, 18, 64, 64, generator=g, device=dev)},
         "pv": {"pv_yield": torch.rand(32, 18, 128, generator=g, device=dev)}}
opt = HipAdam(model.parameters(), lr=5e-4, capturable=True)
step = GraphedTrainStep(model, opt, batch, warmup=3)
for _ in range(5):
    step.graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(60):
    step.graph.replay()
torch.cuda.synchronize()
print(f"graph replay {(time.perf_counter() - t0) / 60 * 1e3:.3f} ms per step")
step.close()
