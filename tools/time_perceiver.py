"""Times the PerceiverModel train step (fwd + NMAE + bwd + Adam) at the reference's configuration
(configs/model/perceiver.yaml: history 60 / forecast 30 minutes -> T = 19 frames = 19 weight-tied layers, 64 x 64 x 11
satellite crops, 128 latents x 64) and, with --cpu, the torch-CPU restatement on a reduced batch."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predict_pv_yield_amd.data.fake import FakeDataConfiguration, make_fake_batch
from predict_pv_yield_amd.models.perceiver.perceiver import PerceiverModel

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--cpu", action="store_true")
ap.add_argument("--operands", default="bf16", choices=["f32", "bf16"])
args = ap.parse_args()
dev = torch.device("cuda:0")
kw = dict(history_minutes=60, forecast_minutes=30, batch_size=args.batch, num_latents=128, latent_dim=64, embedding_dem=16,
          output_variable="gsp_yield")
torch.manual_seed(0)
model = PerceiverModel(**kw, operand_dtype=args.operands).to(dev)
cfg = FakeDataConfiguration(batch_size=args.batch, history_minutes=60, forecast_minutes=30, satellite_image_size_pixels=64,
                            nwp_image_size_pixels=64)
batch = make_fake_batch(cfg, torch.Generator().manual_seed(1)).to(dev)
opt = model.configure_optimizers()


def step():
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(batch, 0)
    loss.backward()
    opt.step()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
t = 19
frames = args.batch * t
flop_fwd = frames * (2 * 2 * 128 * 4096 * 64 * t + 2 * 4096 * 37 * 128 * 2        # cross-attention products, 2 kv projections
                     + t * (2 * 128 * 64 * (64 + 64) + 2 * 128 * 64 * 2048 + 4 * 2 * 128 * 128 * 64 * 2   # q/out, latent qkv+out, latent attention
                            + 2 * 2 * (128 * 64 * 512 + 128 * 256 * 64)))                               # two GEGLU feed-forwards
print(f"PerceiverModel train step B={args.batch} (T={t}, {frames} frames x {t} layers): {dt * 1e3:.1f} ms -> "
      f"{args.batch / dt:.1f} samples/s, ~{3 * flop_fwd / dt / 1e12:.1f} TFLOP/s f32 (loss {float(loss.detach()):.4f}); "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
if args.cpu:
    from oracle import perceiver_oracle as po
    n = min(2, args.batch)
    oracle = po.OraclePerceiverModel(**dict(kw, batch_size=n))
    oracle.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    cb = batch.to("cpu")
    a = (cb.satellite.data[:n], cb.nwp.data[:n], cb.gsp.gsp_yield[:n], cb.gsp.gsp_id[:n])
    o = torch.optim.Adam(oracle.parameters(), lr=5e-4)
    t0 = time.perf_counter()
    o.zero_grad()
    y = oracle(*a)
    (y - cb.gsp.gsp_yield[:n, -oracle.forecast_len:, 0]).abs().mean().backward()
    o.step()
    dtc = time.perf_counter() - t0
    print(f"torch-CPU restatement, B={n}, {torch.get_num_threads()} threads: {dtc * 1e3:.0f} ms -> {n / dtc:.2f} samples/s")
