"""The engine clock a kernel really runs at: a one-wave watcher (pv_clock_watch) on a side stream samples the shader-cycle counter
against the constant 100 MHz counter while ONE kind of kernel loops on the main stream.  sysfs (bench.py's
device_clocks_during_timed_steps) averages over ~10 ms and cannot see a 70 us kernel.
   python tools/probes/sclk_under_kernels.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from predict_pv_yield_amd import hip_ops as K

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
b = 32
side = torch.cuda.Stream()


def watch(fn, label, seconds=0.03, flop=None, nbytes=None):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    n = 40000
    buf = K.clock_watch_launch(n, 40, side)      # ~40 x 64 cycles between samples: the watcher lives ~45 ms
    time.sleep(0.002)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        fn()
        reps += 1
    e1.record()
    torch.cuda.synchronize()
    s = buf.cpu().numpy().astype(np.float64)
    t_us = (s[:, 1] - s[0, 1]) / 100.0            # the watcher's own time axis (100 MHz counter)
    dt, dr = np.diff(s[:, 0]), np.diff(s[:, 1])
    mhz = np.where(dr > 0, dt / np.maximum(dr, 1) * 100.0, np.nan)
    mid = (t_us[1:] > 8000) & (t_us[1:] < 2000 + seconds * 1e6 - 5000)      # well inside the loop (it starts ~2 ms into the watcher's life)
    busy = mhz[mid & ~np.isnan(mhz)]
    tail = mhz[(t_us[1:] > 2000 + seconds * 1e6 + 3000) & ~np.isnan(mhz)]
    us = e0.elapsed_time(e1) * 1e3 / max(reps, 1)
    extra = ""
    if flop:
        extra += (f"  {flop / us / 1e6:7.0f} TFLOP/s = {flop / us / 1e6 / 2500 * 100:4.1f} % of 2.5 PF, "
                  f"{flop / us / 1e6 / (2500 * np.median(busy) / 2400) * 100:4.1f} % of the peak AT THIS CLOCK")
    if nbytes:
        extra += f"  {nbytes / us / 1e6:5.2f} TB/s"
    print(f"{label:44s} {us:8.1f} us/launch  sclk median {np.median(busy):6.0f} MHz  (p10 {np.percentile(busy, 10):5.0f}, p90 {np.percentile(busy, 90):5.0f}, "
          f"{len(busy)} samples; after the loop {np.median(tail) if len(tail) else float('nan'):5.0f}; watcher lived {t_us[-1] / 1e3:.1f} ms){extra}")


# idle
watch(lambda: None, "nothing (idle device)")
# calibration loops
sink = torch.zeros(4096, device=dev)
from predict_pv_yield_amd._lib import get_lib, ptr, current_stream_ptr, check
lib = get_lib()
watch(lambda: check(lib.pv_calibrate_mfma_bf16(ptr(sink), 512, 4000, current_stream_ptr())), "bare MFMA loop (calibration)",
      flop=512 * 4 * 4000 * 8 * 16 * 16 * 32 * 2)
src = torch.empty(256 << 20, dtype=torch.float32, device=dev)
dst = torch.empty_like(src)
watch(lambda: check(lib.pv_calibrate_copy_f32(ptr(src), ptr(dst), src.numel(), current_stream_ptr())), "plain copy (calibration)",
      nbytes=2 * src.numel() * 4)
del src, dst
# conv kernels of the headline step (layer 2: 32 -> 32 channels, [32,16,62,62] -> [32,14,60,60])
w = torch.randn(32, 32, 3, 3, 3, device=dev, generator=g) * 0.05
x = torch.randn(b, 16, 62, 62, 32, device=dev, generator=g).relu().to(torch.bfloat16)
dy = torch.randn(b, 14, 60, 60, 32, device=dev, generator=g).to(torch.bfloat16)
wp, wpt = K.conv3d_pack_weight_bf16(w, transpose_flip=False), K.conv3d_pack_weight_bf16(w, transpose_flip=True)
bias = torch.zeros(32, device=dev)
fl = 2 * 27 * 32 * 32 * b * 14 * 60 * 60
watch(lambda: K.conv3d_fwd_bf16(x, None, wp, bias, 32, 32, (0, 0, 0), True, False), "conv forward 32->32 (input-stationary)", flop=fl)
watch(lambda: K.conv3d_fwd_bf16(dy, None, wpt, None, 32, 32, (2, 2, 2), relu=False, y_ncdhw=False, out_gate=x), "conv dgrad + bf16 gate", flop=fl)
watch(lambda: K.conv3d_bwd_weight_bf16(x, dy, None, 32, 32, (0, 0, 0)), "conv weight gradient (+ slab reduce)", flop=fl)
# fc1 one-pass backward
k = 1003520
xf = torch.randn(b, k, device=dev, generator=g).relu().to(torch.bfloat16)
wf = torch.randn(128, k, device=dev, generator=g) * 0.01
m1, v1 = torch.zeros_like(wf), torch.zeros_like(wf)
sh = K.cast_f32_to_bf16(wf)
gy = torch.randn(b, 128, device=dev, generator=g) * 1e-3
yy = torch.rand(b, 128, device=dev, generator=g)
st = [0]


def fc1():
    st[0] += 1
    K.linear_wgrad_dx_adam_bf16(xf, gy, yy, wf, m1, v1, sh, st[0], need_dx=True, need_db=False, gate_dx_by_x=True)


watch(fc1, "fc1 backward in one pass (HBM-bound)", nbytes=128 * k * 26 + 2 * b * k * 2)
# the advection pipeline (config 3, SURVEY section-8(d) input): 55 % of it is the level-0 iteration kernel
del xf, wf, m1, v1, sh
from predict_pv_yield_amd import optical_flow as of
from predict_pv_yield_amd.data.synthetic import advected_counts
raw = torch.from_numpy(advected_counts(batch=b, seed=1234)[0]).to(dev)
watch(lambda: of.advect_future_frames(raw, 6), "advection pipeline (121 Farneback pairs / sample)")
