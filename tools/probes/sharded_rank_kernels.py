"""Kernel table of ONE rank's step of bench.py --gpus 8 --global-batch 512 (sharded fc1 update), emulated on one GPU without any
exchange (bench.measure_sharded_rank_compute's set-up).   python tools/probes/sharded_rank_kernels.py [world] [global_batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unittest import mock
import torch
import bench
from predict_pv_yield_amd import distributed as D
from predict_pv_yield_amd.models.conv3d.model import Model

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
gb = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
b = gb // world
torch.manual_seed(518)
model = Model(**bench.MODEL_KW, history_minutes=55, precision="bf16").to(dev)
model.batch_size = max(model.batch_size, b)
opt = model.configure_optimizers()
opt.grad_scale = 1.0 / world
rows = model.fc1.weight.shape[0]
shard = (0, rows // world)
g = torch.Generator(device=dev).manual_seed(518)
batch = {"satellite": {"data": torch.randn(b, 11, 18, 64, 64, generator=g, device=dev)}, "pv": {"pv_yield": torch.rand(b, 18, 128, generator=g, device=dev)}}
OPS = bench.TIMED_OPS + ("linear_wgrad_bf16out", "adam_step_bf16grad", "linear_fwd_f32", "linear_bwd_f32", "forecast_losses")
with mock.patch.object(D, "row_shard", lambda n_rows, rank=None, world=None: shard), \
        mock.patch.object(D, "all_gather_rows", lambda full, async_op=True: None):
    opt.set_large_grad_mode("sharded")
    for p in opt.large_params():
        p._pv_on_grad = lambda gbuf, param=None: setattr(param, "_pv_grad_shard", gbuf[shard[0]:shard[1]])

    def step():
        opt.zero_grad(set_to_none=True)
        model.training_step(batch, 0).backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    print(f"one rank of {world}, per-GPU batch {b}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per step")
    with bench.LaunchTimer(OPS) as lt:
        for _ in range(5):
            step()
    tot = 0.0
    for (name, shape, extra), (secs, n) in sorted(lt.summary().items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        tot += secs * n / 5
        print("%8.1f us x%d %s %s %s" % (secs * 1e6, n // 5, name, extra, list(shape)))
    print("sum of timed launches %.1f us" % (tot * 1e6))
