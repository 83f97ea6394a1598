"""Config 3 as a two-stage software pipeline: the advection of batch i+1 (Farneback + remap, stream A) runs while batch i
is trained (conv3d step, stream B).  Prints sequential vs overlapped time per batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predict_pv_yield_amd import optical_flow as of
from predict_pv_yield_amd.models.conv3d.model import Model

dev = torch.device("cuda:0")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = 12
g = torch.Generator(device=dev).manual_seed(1)
raws = [(torch.rand(b, 12, 11, 64, 64, generator=g, device=dev) * 1023).to(torch.int16) for _ in range(2)]
torch.manual_seed(518)
model = Model(include_pv_yield=False, include_nwp=False, forecast_minutes=30, number_of_conv3d_layers=4, conv3d_channels=32,
              image_size_pixels=64, number_sat_channels=11, fc1_output_features=128, fc2_output_features=128,
              fc3_output_features=64, output_variable="pv_yield", history_minutes=55, precision="bf16").to(dev)
model.batch_size = max(model.batch_size, b)
opt = model.configure_optimizers()
pv = torch.rand(b, 18, 128, device=dev)


def train(x):
    opt.zero_grad(set_to_none=True)
    model.training_step({"satellite": {"data": x}, "pv": {"pv_yield": pv}}, 0).backward()
    opt.step()


# sequential
for _ in range(2):
    train(of.advect_future_frames(raws[0], 6))
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    train(of.advect_future_frames(raws[i & 1], 6))
torch.cuda.synchronize()
seq = (time.perf_counter() - t0) / n

# pipelined: advection one batch ahead on its own stream
s_flow = torch.cuda.Stream()
main = torch.cuda.current_stream()
with torch.cuda.stream(s_flow):
    x_next = of.advect_future_frames(raws[0], 6)
ready = torch.cuda.Event()
ready.record(s_flow)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    main.wait_event(ready)
    x_cur = x_next
    with torch.cuda.stream(s_flow):
        x_next = of.advect_future_frames(raws[(i + 1) & 1], 6)
        ready = torch.cuda.Event()
        ready.record(s_flow)
    x_cur.record_stream(main)
    train(x_cur)
torch.cuda.synchronize()
pipe = (time.perf_counter() - t0) / n
print(f"B={b}: sequential {seq * 1e3:.3f} ms/batch ({b / seq:.0f} samples/s), advection overlapped with the train step {pipe * 1e3:.3f} ms/batch ({b / pipe:.0f} samples/s)")


def epoch(k):
    for bt in of.AdvectingLoader(({"satellite": {"data": raws[i & 1]}, "pv": {"pv_yield": pv}} for i in range(k)), 6):
        opt.zero_grad(set_to_none=True)
        model.training_step(bt, 0).backward()
        opt.step()


model.future_frames = "optical_flow"
for warm, k in ((3, 10), (0, 10), (0, 40)):
    if warm:
        epoch(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    epoch(k)
    torch.cuda.synchronize()
    d = (time.perf_counter() - t0) / k
    print(f"AdvectingLoader, {k} batches after {warm} warm-up: {d * 1e3:.3f} ms/batch ({b / d:.0f} samples/s); "
          f"reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB")
