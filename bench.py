#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X hot path (BASELINE.json metric, configs[1]).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one train step of the Conv3D PV-yield model (forward + NMAE loss + backward + Adam [+ gradient
all-reduce over RCCL when N > 1]) on one batch of synthetic PV-site crop stacks [B, 11, 18, 64, 64]
(12 observed + 6 forecast frames) already resident in HBM.  Per-GPU batch is fixed
(weak scaling).  Rank 0 prints ONE JSON line with the whole-job samples/s plus
  roofline     : dominant kernel (conv3d implicit-GEMM MFMA kernels), algorithmic FLOPs / measured launch time
                 (HIP events on the launching stream) against the dense bf16 MFMA peak;
  cpu_baseline : the torch-CPU oracle (identical arithmetic to the reference's Lightning path) timed on this
                 host's cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK = 2.5e15     # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK = 8.0e12


def conv_layer_shapes(t, hw, c_in, c, layers):
    """[(c_in, t_in, h_in, t_out, h_out)] for valid 3x3x3 convs."""
    out, ci, ti, hi = [], c_in, t, hw
    for _ in range(layers):
        out.append((ci, ti, hi, ti - 2, hi - 2))
        ci, ti, hi = c, ti - 2, hi - 2
    return out


def conv_flops(batch, c_in, c_out, t_out, h_out):
    return 2.0 * batch * c_out * c_in * 27 * t_out * h_out * h_out


def time_kernel(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()  # recorded on torch's current stream == the stream the C ABI launches on
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def measure_conv_roofline(batch, hist_frames, dev):
    """Times every conv launch of one train step in isolation (same shapes, same kernels) and returns the
    roofline object of the dominant kernel."""
    from predict_pv_yield_amd import hip_ops as K
    t = hist_frames
    shapes = conv_layer_shapes(t, 64, 11, 32, 4)
    per_kernel = {}
    for li, (ci, ti, hi, to, ho) in enumerate(shapes):
        cpad = K.bf16_cpad(ci)
        x = torch.randn(batch, ti, hi, hi, cpad, device=dev).to(torch.bfloat16)
        w = torch.randn(32, ci, 3, 3, 3, device=dev) * 0.05
        bias = torch.zeros(32, device=dev)
        wp = K.conv3d_pack_weight_bf16(w)
        y = K.conv3d_fwd_bf16(x, None, wp, bias, ci, 32, (0, 0, 0), True, False)
        dy = torch.randn_like(y)
        fl = conv_flops(batch, ci, 32, to, ho)
        name_f = f"conv3d_fwd_bf16_kernel<{cpad}>"
        if li == 0:   # the step's first layer reads the f32 NCDHW input itself and leaves the bf16 image for its wgrad
            xf = torch.randn(batch, ci, ti, hi, hi, device=dev)
            name_f = f"conv3d_fwd_bf16_kernel<{cpad}> (f32 NCDHW input + bf16 NDHWC copy out)"
            d = time_kernel(lambda: K.conv3d_fwd_bf16_f32in(xf, wp, bias, 32, (0, 0, 0), True, want_packed=True))
            del xf
        else:
            d = time_kernel(lambda: K.conv3d_fwd_bf16(x, None, wp, bias, ci, 32, (0, 0, 0), True, False))
        per_kernel.setdefault(name_f, [0.0, 0.0, 0])
        per_kernel[name_f][0] += d; per_kernel[name_f][1] += fl; per_kernel[name_f][2] += 1
        d = time_kernel(lambda: K.conv3d_bwd_weight_bf16(x, dy, None, ci, 32, (0, 0, 0)))  # dy arrives pre-gated
        name_w = f"conv3d_wgrad_bf16_kernel<{cpad}>"
        per_kernel.setdefault(name_w, [0.0, 0.0, 0])
        per_kernel[name_w][0] += d; per_kernel[name_w][1] += fl; per_kernel[name_w][2] += 1
        if li > 0:  # dgrad = the forward kernel on dy (32 channels) with mirrored weights
            wpt = K.conv3d_pack_weight_bf16(w, transpose_flip=True)
            d = time_kernel(lambda: K.conv3d_fwd_bf16(dy, None, wpt, None, 32, ci, (2, 2, 2), False, False, out_gate=x))
            per_kernel["conv3d_fwd_bf16_kernel<32>"][0] += d
            per_kernel["conv3d_fwd_bf16_kernel<32>"][1] += fl
            per_kernel["conv3d_fwd_bf16_kernel<32>"][2] += 1
        del x, y, dy
    dom = max(per_kernel, key=lambda k: per_kernel[k][0])
    secs, flops, launches = per_kernel[dom]
    achieved = flops / secs / 1e12
    detail = {k: {"ms_per_step": round(v[0] * 1e3, 4), "tflops": round(v[1] / v[0] / 1e12, 2), "launches": v[2]}
              for k, v in per_kernel.items()}
    return {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": MFMA_BF16_PEAK / 1e12,
            "unit": "TFLOP/s", "frac": round(achieved * 1e12 / MFMA_BF16_PEAK, 4), "traffic": None,
            "avg_launch_ms": round(secs / launches * 1e3, 4), "launches_per_step": launches,
            "algorithmic_gflop_per_step": round(flops / 1e9, 2), "kernels": detail}


TRAFFIC_PROFILE = os.path.join("profiles", "r01", "pmc_hbm_traffic_bench_B32_v12.json")


def committed_hbm_traffic(cpad: int, batch: int):
    """HBM bytes per launch of the dominant conv kernel family, from the COMMITTED PMC passes (tools/pmc_traffic.py:
    separate --pmc FETCH_SIZE / WRITE_SIZE runs of this bench, gfx950-corrected).  PMC counters cannot be collected from
    inside the timed run; the figure is per launch at B=32 and only reported for that batch."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), TRAFFIC_PROFILE)
    if batch != 32 or cpad != 32 or not os.path.exists(path):
        return None
    ks = json.load(open(path))["kernels"]
    fam = [v for k, v in ks.items() if k.startswith("pv::conv3d_fwd_bf16_v3_kernel") or
           k.startswith("pv::conv3d_fwd_bf16_v2_kernel") or k.startswith("pv::conv3d_fwd_bf16_kernel<32")]
    n = sum(v["launches"] for v in fam)
    return round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in fam) / n) if n else None


def measure_hbm_kernels(model, opt, batch_size, t_frames, dev):
    """HBM-bound side of the step: the fused fc1 wgrad+Adam pass (one stream over p, m, v + bf16 shadow) and the
    config-3 advection stages (remap x6 of 11 channels: 16 algorithmic bytes per output pixel)."""
    from predict_pv_yield_amd import hip_ops as K
    from predict_pv_yield_amd import optical_flow as of
    out = {}
    p = model.fc1.weight
    st = opt.state.get(p)
    if st:
        n, k = p.shape
        x = torch.randn(batch_size, k, device=dev).to(torch.bfloat16)
        dy = torch.randn(batch_size, n, device=dev) * 1e-6
        from predict_pv_yield_amd.functional import bf16_shadow_of
        sh = bf16_shadow_of(p)
        pc, mc, vc = p.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone()
        d = time_kernel(lambda: K.linear_wgrad_adam_bf16(x, dy, None, pc, mc, vc, sh, 10), iters=5, warm=1)
        byt = n * k * (3 * 4 * 2 + 2) + batch_size * k * 2   # p,m,v read+write, shadow write, x read once
        # calibration on the same (warm) device: a plain device-to-device copy of one f32 copy of the matrix, the
        # 1:1 read/write mix this pass is made of
        dc = time_kernel(lambda: mc.copy_(pc), iters=5, warm=1)
        out["fc1_wgrad_adam"] = {"bound": "hbm", "ms": round(d * 1e3, 4), "algorithmic_GB": round(byt / 1e9, 3),
                                 "achieved": round(byt / d / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                 "frac": round(byt / d / HBM_PEAK, 4),
                                 "device_copy_GBps_same_run": round(2 * 4 * n * k / dc / 1e9, 1)}
        del x, dy, pc, mc, vc
    b = batch_size
    src = torch.randn(b * 11, 64, 64, device=dev)
    fl = torch.randn(b * 11, 64, 64, 2, device=dev)
    d = time_kernel(lambda: K.remap_bilinear(src, fl, 6, 1.0, 1, 0.0), iters=20, warm=3)
    byt = b * 11 * 6 * 4096 * 16
    out["remap_x6"] = {"bound": "hbm", "ms": round(d * 1e3, 4), "achieved": round(byt / d / 1e9, 1),
                       "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(byt / d / HBM_PEAK, 4),
                       "algorithmic_bytes_per_output_pixel": 16}
    raw = torch.randint(0, 1021, (b, 12, 11, 64, 64), dtype=torch.int16, device=dev)
    d = time_kernel(lambda: of.advect_future_frames(raw, 6), iters=3, warm=1)
    out["config3_advection_pipeline"] = {"ms": round(d * 1e3, 3), "samples_per_s": round(b / d, 1),
                                         "farneback_pairs_per_s": round(b * 121 / d, 0),
                                         "workload": f"raw [B={b},12,11,64,64] int16 -> u8 -> 121 Farneback pairs/sample -> "
                                                     "weighted mean -> normalise -> 6 advected frames"}
    return out


def measure_other_models(dev):
    """Train-step throughput of the other models built on the same kernels (not the headline metric; outside the timed
    region): the optical-flow notebook's 3-D CNN at its own batch size, and the PerceiverModel at the reference's
    configs/model/perceiver.yaml (T = 19, 64 x 64, batch 8)."""
    out = {}

    def time_steps(step, n, warm):
        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    from predict_pv_yield_amd.models.conv3d import flow_autoencoder as fa
    torch.manual_seed(0)
    g = torch.Generator(device=dev).manual_seed(1)
    b = 64
    batch = {fa.HISTORICAL_SAT_IMAGES: torch.randn(b, 4, 128, 128, generator=g, device=dev),
             fa.OPTICAL_FLOW_PREDICTIONS: torch.randn(b, 128, 128, generator=g, device=dev),
             fa.FORECAST_HORIZON: torch.randn(b, generator=g, device=dev),
             fa.TARGET_SAT_IMAGE: torch.randn(b, 64, 64, generator=g, device=dev)}
    ae = fa.LitAutoEncoder().to(dev)
    opt = ae.configure_optimizers()

    def ae_step():
        opt.zero_grad(set_to_none=True)
        ae.training_step(batch, 0).backward()
        opt.step()

    d = time_steps(ae_step, 5, 2)
    out["flow_autoencoder"] = {"workload": "LitAutoEncoder train step (notebook 13), [64,2,5,128,128] -> [64,1,1,64,64], f32",
                               "ms_per_step": round(d * 1e3, 2), "samples_per_s": round(b / d, 1),
                               "tflops_f32": round(3 * 2 * 1.097e9 * b / d / 1e12, 1)}
    del ae, opt, batch

    from predict_pv_yield_amd.data.fake import FakeDataConfiguration, make_fake_batch
    from predict_pv_yield_amd.models.perceiver.perceiver import PerceiverModel
    b = 8
    pm = PerceiverModel(history_minutes=60, forecast_minutes=30, batch_size=b, num_latents=128, latent_dim=64,
                        embedding_dem=16, output_variable="gsp_yield").to(dev)
    cfg = FakeDataConfiguration(batch_size=b, history_minutes=60, forecast_minutes=30, satellite_image_size_pixels=64,
                                nwp_image_size_pixels=64)
    pbatch = make_fake_batch(cfg, torch.Generator().manual_seed(2)).to(dev)
    popt = pm.configure_optimizers()

    def p_step():
        popt.zero_grad(set_to_none=True)
        pm.training_step(pbatch, 0).backward()
        popt.step()

    d = time_steps(p_step, 3, 1)
    out["perceiver"] = {"workload": "PerceiverModel train step (configs/model/perceiver.yaml): B=8, T=19 frames = 19 weight-tied "
                                    "layers, 64x64x11, 128 latents x 64, f32 MFMA",
                        "ms_per_step": round(d * 1e3, 1), "samples_per_s": round(b / d, 1)}
    return out


def cpu_baseline(model_kwargs, t_frames, budget_s=24.0):
    """torch-CPU oracle train step (fwd + NMAE + bwd + Adam), B = 8, on this host's cores.  oneDNN's Conv3d does not
    always scale to every hardware thread, so two thread counts share the budget and the faster one is reported."""
    from oracle import conv3d_oracle as co
    torch.manual_seed(518)
    kw = {k: v for k, v in model_kwargs.items() if k not in ("precision", "future_frames")}
    m = co.OracleConv3dModel(**kw)
    b = 8
    g = torch.Generator().manual_seed(518)
    sat = torch.randn(b, 11, t_frames, 64, 64, generator=g)
    pv = torch.rand(b, t_frames, 128, generator=g)
    opt = co.make_optimizer(m)
    all_threads = torch.get_num_threads()
    best = None
    for threads in sorted({all_threads, max(1, all_threads // 4)}, reverse=True):
        torch.set_num_threads(threads)
        co.train_steps(m, sat, pv, 1, opt)  # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            co.train_steps(m, sat, pv, 1, opt)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s / 2 or n >= 12:
                break
        rate = n * b / el
        if best is None or rate > best[0]:
            best = (rate, threads, n)
    torch.set_num_threads(all_threads)
    return {"value": round(best[0], 2), "unit": "samples/s", "cores": best[1], "kind": "port",
            "sample": f"{best[2]} train steps at B={b}, T={t_frames}, fp32, torch-CPU oracle (oracle/conv3d_oracle.py); "
                      f"best of {all_threads} and {max(1, all_threads // 4)} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch of PV-site crop stacks")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: fix the GLOBAL batch (e.g. 512) and give each of the N GPUs global/N samples; "
                         "default 0 = weak scaling with --batch samples per GPU")
    ap.add_argument("--history-minutes", type=int, default=55, help="55 -> T=18 (12 observed + 6 forecast frames)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--overlap-update", action="store_true",
                    help="N = 1: launch fc1's fused wgrad+Adam from backward on a side stream (under the conv backward)")
    ap.add_argument("--f32-grads", action="store_true", help="N > 1: all-reduce fc1's gradient in f32 instead of bf16")
    ap.add_argument("--grad-sync", choices=["sharded", "allreduce"], default="sharded",
                    help="N > 1, bf16 gradients: 'sharded' = reduce-scatter + per-rank Adam over its rows of fc1 + all-gather "
                         "of the bf16 operand copy (default); 'allreduce' = every rank steps the whole matrix")
    args = ap.parse_args()

    from predict_pv_yield_amd import distributed as D
    from predict_pv_yield_amd.models.conv3d.model import Model

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the product has no CPU path)")
    distributed = D.init_from_env()
    world = torch.distributed.get_world_size() if distributed else 1
    rank = torch.distributed.get_rank() if distributed else 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    dev = torch.device("cuda", D.local_device_index())
    torch.cuda.set_device(dev)

    model_kwargs = dict(include_pv_yield=False, include_nwp=False, forecast_minutes=30,
                        history_minutes=args.history_minutes, number_of_conv3d_layers=4, conv3d_channels=32,
                        image_size_pixels=64, number_sat_channels=11, fc1_output_features=128,
                        fc2_output_features=128, fc3_output_features=64, output_variable="pv_yield",
                        precision=args.precision)
    torch.manual_seed(518)  # configs/experiment/conv3d.yaml:16
    model = Model(**model_kwargs).to(dev)
    t_frames = model.history_len_5 + model.forecast_len_5 + 1
    if distributed:
        D.broadcast_parameters(model)
    opt = model.configure_optimizers()
    opt.grad_scale = 1.0 / world
    if args.overlap_update and not distributed:
        opt.overlap_large_update = True
        opt.set_large_grad_mode("fused")
    if distributed:
        # the all-reduce needs a materialised fc1 gradient: bf16 on the wire (half the xGMI bytes), f32 with --f32-grads
        opt.set_large_grad_mode("autograd" if args.f32_grads else ("sharded" if args.grad_sync == "sharded" else "bf16"))

    g = torch.Generator(device=dev).manual_seed(518 + rank)
    b = args.batch
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit("--global-batch must be a multiple of the number of GPUs")
        b = args.global_batch // world
    model.batch_size = max(model.batch_size, b)  # BaseModel.batch_size slices the target (base_model.py:95)
    batch = {"satellite": {"data": torch.randn(b, 11, t_frames, 64, 64, generator=g, device=dev)},
             "pv": {"pv_yield": torch.rand(b, t_frames, 128, generator=g, device=dev)}}

    sync = D.OverlappedGradSync(model) if distributed else None

    def step():
        opt.zero_grad(set_to_none=True)
        loss = model.training_step(batch, 0)
        loss.backward()      # under N > 1 fc1's gradient all-reduce starts from a hook inside backward
        if sync is not None:
            sync.finish()
        opt.step()
        return loss

    first = None
    for i in range(args.warmup):
        l = step()
        if i == 0:
            first = float(l.detach())
    if distributed:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    if distributed:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
    last = float(last.detach())

    if rank == 0:
        value = world * b * args.steps / elapsed
        out = {
            "metric": "PV-site samples/sec (train step), conv3d 12->6 frames",
            "value": round(value, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"conv3d train step (fwd + NMAE + bwd + Adam): sat [B,11,{t_frames},64,64] N(0,1), "
                                   f"4x Conv3d(3x3x3, 32ch) + fc 128/128/64, {sum(p.numel() for p in model.parameters())/1e6:.1f} M params",
                       "per_gpu_batch": b, "global_batch": b * world, "t_frames": t_frames,
                       "parallelism": f"dp{world}" if world > 1 else "single"},
            "train_nmae_first_step": round(first, 6) if first is not None else None,
            "train_nmae_last_step": round(last, 6),
        }
        if not args.no_roofline and world == 1:
            out["roofline"] = measure_conv_roofline(b, t_frames, dev)
            out["roofline"]["traffic"] = committed_hbm_traffic(32, b)
            out["roofline"]["traffic_profile"] = TRAFFIC_PROFILE + " (separate --pmc FETCH_SIZE / WRITE_SIZE passes; bytes per launch)"
            out["hbm_bound_kernels"] = measure_hbm_kernels(model, opt, b, t_frames, dev)
            del model, opt, batch
            torch.cuda.empty_cache()
            out["other_models"] = measure_other_models(dev)
        else:
            out["roofline"] = None
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(model_kwargs, t_frames)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if distributed:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
