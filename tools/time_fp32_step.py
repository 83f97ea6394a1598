"""Times the headline train step with precision="fp32" (exact-f32 kernels) and prints the per-launch table."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from predict_pv_yield_amd.models.conv3d.model import Model

b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
torch.manual_seed(518)
model = Model(**bench.MODEL_KW, history_minutes=55, precision="fp32").to(dev)
model.batch_size = max(model.batch_size, b)
opt = model.configure_optimizers()
g = torch.Generator(device=dev).manual_seed(518)
batch = {"satellite": {"data": torch.randn(b, 11, 18, 64, 64, generator=g, device=dev)},
         "pv": {"pv_yield": torch.rand(b, 18, 128, generator=g, device=dev)}}


def step():
    opt.zero_grad(set_to_none=True)
    model.training_step(batch, 0).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    step()
torch.cuda.synchronize()
d = (time.perf_counter() - t0) / n
print(f"fp32 step B={b}: {d * 1e3:.2f} ms  {b / d:.0f} samples/s")
import types
from predict_pv_yield_amd import hip_ops as K
names = [n_ for n_ in dir(K) if isinstance(getattr(K, n_), types.FunctionType) and not n_.startswith("_")
         and getattr(K, n_).__module__ == K.__name__ and n_ not in ("conv_geom", "conv_dims", "bf16_cpad")]
with bench.LaunchTimer(names) as lt:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
rows = sorted(lt.summary().items(), key=lambda kv: -kv[1][0] * kv[1][1])
for (name, shape, extra), (sec, n_) in rows[:18]:
    print(f"  {name:36s} {str(shape):34s} {sec * 1e6:9.1f} us x {n_ // 3}")
