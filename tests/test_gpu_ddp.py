"""Data-parallel step on the GPU with world_size 2 (both ranks share the one GPU of the test box; gloo carries the
collectives): the sharded update of the big layer (reduce-scatter -> per-rank Adam over its rows -> all-gather of the
bf16 operand copy) must give exactly the parameters of the bf16 all-reduce path, and both must track a single-process
run on the whole batch."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_two_ranks(mode, out_path, steps=3, world=2, global_batch=4, fuse_min_numel=1):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PV_DIST_BACKEND="gloo", PV_SINGLE_DEVICE="1", PV_TEST_GLOBAL_BATCH=str(global_batch),
                   PV_TEST_FUSE_MIN_NUMEL=str(fuse_min_numel))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "ddp_two_rank_worker.py"), mode, out_path,
                                       str(steps)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return torch.load(out_path)


def test_sharded_update_equals_allreduce_update(device, tmp_path):
    a = _run_two_ranks("bf16", str(tmp_path / "allreduce.pt"))
    b = _run_two_ranks("sharded", str(tmp_path / "sharded.pt"))
    assert a["mode"] == "bf16" and b["mode"] == "sharded"
    assert a["losses"] == b["losses"]
    assert torch.equal(a["y"], b["y"])
    for k in a["state"]:
        assert torch.equal(a["state"][k], b["state"][k]), k          # consolidated f32 parameters, every row
    assert torch.equal(a["exp_avg_fc1"], b["exp_avg_fc1"])

    # single process, whole batch, same bf16 gradient format: the data-parallel runs follow it closely
    from predict_pv_yield_amd.models.conv3d.model import Model
    from predict_pv_yield_amd.optim import HipAdam
    from tests.ddp_two_rank_worker import SMALL
    old = HipAdam.FUSE_MIN_NUMEL
    HipAdam.FUSE_MIN_NUMEL = 1
    try:
        torch.manual_seed(518)
        model = Model(**SMALL, precision="bf16").to(device)
        opt = model.configure_optimizers()
        opt.set_large_grad_mode("bf16")
        g = torch.Generator().manual_seed(7)
        sat, pv = torch.randn(4, 11, 25, 16, 16, generator=g), torch.rand(4, 25, 128, generator=g)
        batch = {"satellite": {"data": sat.to(device)}, "pv": {"pv_yield": pv.to(device)}}
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            model.training_step(batch, 0).backward()
            opt.step()
    finally:
        HipAdam.FUSE_MIN_NUMEL = old
    for k, v in model.state_dict().items():
        # 3 Adam steps of lr 5e-4.  Adam moves a weight by up to lr per step whatever the gradient's size, so a bound on
        # the LARGEST difference (<= 3 lr) cannot fail; what carries information is how many weights disagree by a whole
        # step (isolated near-zero gradients whose sign differs: the two half-batch gradients are rounded to bf16
        # separately before they are summed) and the mean distance in units of lr
        d = (v.cpu() - b["state"][k]).abs()
        whole_step = (d > 5e-4).float().mean().item()
        assert whole_step <= 0.01 and d.mean().item() <= 0.2 * 5e-4, (k, whole_step, d.mean().item())


def _run_one_rank_rccl(mode, out_path, steps=3, fuse_min_numel=1):
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               PV_DIST_SINGLE_RANK="1", PV_DIST_TIMEOUT_S="120", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PV_TEST_FUSE_MIN_NUMEL=str(fuse_min_numel))
    env.pop("PV_DIST_BACKEND", None)
    env.pop("PV_SINGLE_DEVICE", None)
    p = subprocess.run([sys.executable, os.path.join(HERE, "ddp_two_rank_worker.py"), mode, out_path, str(steps)], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode()[-3000:]
    return torch.load(out_path)


def test_rccl_branches_run_on_one_rank(device, tmp_path):
    """backend "nccl" (= RCCL), world_size 1, initialised before any GPU call: reduce_scatter_tensor /
    all_gather_into_tensor / all_reduce of bf16 tensors and the flat f32 bucket really execute (on 8 GPUs this is the code
    that runs; `dist.get_backend() == "nccl"` selects it).  With one rank every collective is the identity, so both modes
    must reproduce the single-process bf16-gradient run bit for bit."""
    a = _run_one_rank_rccl("sharded", str(tmp_path / "sharded.pt"))
    b = _run_one_rank_rccl("bf16", str(tmp_path / "bf16.pt"))
    assert a["backend"] == "nccl" and a["world"] == 1 and a["mode"] == "sharded" and b["mode"] == "bf16"
    from predict_pv_yield_amd.models.conv3d.model import Model
    from predict_pv_yield_amd.optim import HipAdam
    from tests.ddp_two_rank_worker import SMALL
    old = HipAdam.FUSE_MIN_NUMEL
    HipAdam.FUSE_MIN_NUMEL = 1
    try:
        torch.manual_seed(518)
        model = Model(**SMALL, precision="bf16").to(device)
        opt = model.configure_optimizers()
        opt.set_large_grad_mode("bf16")
        g = torch.Generator().manual_seed(7)
        sat, pv = torch.randn(4, 11, 25, 16, 16, generator=g), torch.rand(4, 25, 128, generator=g)
        batch = {"satellite": {"data": sat.to(device)}, "pv": {"pv_yield": pv.to(device)}}
        losses = []
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            loss = model.training_step(batch, 0)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
    finally:
        HipAdam.FUSE_MIN_NUMEL = old
    for run in (a, b):
        assert run["losses"] == losses
        for k, v in model.state_dict().items():
            assert torch.equal(v.cpu(), run["state"][k]), (run["mode"], k)


def test_eight_ranks_on_one_gpu_sharded_equals_allreduce_and_follows_one_process(device, tmp_path):
    """The 8-GPU job's shape on the one GPU of the test box: 8 ranks (gloo, PV_SINGLE_DEVICE=1), a global batch of 8 samples
    split one per rank (distributed.shard_range, the --global-batch rule of bench.py), fc1's 16 rows sharded two per rank.
    The row-sharded exchange must give the bits of the bf16 all-reduce, and both must follow a single process that trains
    on the whole batch within the bounds of the two-rank test (eight bf16 addends instead of two: same bounds hold)."""
    a = _run_two_ranks("bf16", str(tmp_path / "allreduce8.pt"), world=8, global_batch=8, fuse_min_numel=100000)
    b = _run_two_ranks("sharded", str(tmp_path / "sharded8.pt"), world=8, global_batch=8, fuse_min_numel=100000)
    assert a["world"] == 8 and b["world"] == 8 and a["mode"] == "bf16" and b["mode"] == "sharded"
    assert a["losses"] == b["losses"]
    for k in a["state"]:
        assert torch.equal(a["state"][k], b["state"][k]), k
    from predict_pv_yield_amd.models.conv3d.model import Model
    from predict_pv_yield_amd.optim import HipAdam
    from tests.ddp_two_rank_worker import SMALL
    old = HipAdam.FUSE_MIN_NUMEL
    HipAdam.FUSE_MIN_NUMEL = 100000
    try:
        torch.manual_seed(518)
        model = Model(**SMALL, precision="bf16").to(device)
        opt = model.configure_optimizers()
        opt.set_large_grad_mode("bf16")
        g = torch.Generator().manual_seed(7)
        sat, pv = torch.randn(8, 11, 25, 16, 16, generator=g), torch.rand(8, 25, 128, generator=g)
        batch = {"satellite": {"data": sat.to(device)}, "pv": {"pv_yield": pv.to(device)}}
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            model.training_step(batch, 0).backward()
            opt.step()
    finally:
        HipAdam.FUSE_MIN_NUMEL = old
    for k, v in model.state_dict().items():
        d = (v.cpu() - b["state"][k]).abs()
        whole_step = (d > 5e-4).float().mean().item()
        assert whole_step <= 0.02 and d.mean().item() <= 0.3 * 5e-4, (k, whole_step, d.mean().item())


def test_bench_global_batch_path_with_eight_ranks_on_one_gpu(device, tmp_path):
    """bench.py --gpus 8 --global-batch 64 as the driver launches it (one process per rank, RANK / WORLD_SIZE / MASTER_* in
    the environment), on the one GPU of the test box with gloo: the strong-scaling branch (shard_range of the global batch,
    negotiate_grad_sync, the default exchange -- K-sharded since round 6 -- of the full 128 M-parameter fc1, max-over-ranks
    timing) runs to its JSON line before an 8-GPU node ever sees it."""
    import json
    port = _free_port()
    root = os.path.dirname(HERE)
    procs = []
    for rank in range(8):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="8", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PV_DIST_BACKEND="gloo", PV_SINGLE_DEVICE="1", PV_DIST_TIMEOUT_S="900")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--global-batch", "64",
                                       "--steps", "2", "--warmup", "1", "--no-roofline", "--no-cpu-baseline",
                                       "--only-requested-mode"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=root))
    outs = [p.communicate(timeout=1500) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-3000:]
    json_lines = lambda o: [ln for ln in o.decode().splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(json_lines(outs[0][0])) == 1
    line = json.loads(json_lines(outs[0][0])[0])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["config"]["global_batch"] == 64
    assert line["config"]["per_gpu_batch"] == 8 and "dp8 (ksharded)" in line["config"]["parallelism"]
    assert line["value"] > 0 and line["train_nmae_last_step"] == line["train_nmae_last_step"]      # finite
    assert all(not json_lines(o) for o, _ in outs[1:]), "only rank 0 prints the line"


def test_bench_refuses_a_demoted_exchange_unless_allowed(device):
    """a scaling number measured on a simpler gradient exchange than the requested one must not pass for
    it.  bench.py --gpus 2 (two gloo ranks on the one GPU) with the sharded mode's trial step made to fail
    (PV_DIST_FAIL_MODES=sharded): every rank exits non-zero and no JSON line appears; with --allow-demotion the run completes
    on the bf16 all-reduce and says so in config.parallelism / config.collectives."""
    import json
    root = os.path.dirname(HERE)

    def run(extra):
        port = _free_port()
        procs = []
        for rank in range(2):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), PV_DIST_BACKEND="gloo", PV_SINGLE_DEVICE="1", PV_DIST_TIMEOUT_S="900",
                       PV_DIST_FAIL_MODES="sharded")
            procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "4", "--steps", "2",
                                           "--warmup", "1", "--no-roofline", "--no-cpu-baseline", "--grad-sync", "sharded"] + extra, env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=root))
        outs = [p.communicate(timeout=1200) for p in procs]
        return procs, outs

    json_lines = lambda o: [ln for ln in o.decode().splitlines() if ln.startswith("{") and '"metric"' in ln]
    procs, outs = run([])
    assert all(p.returncode != 0 for p in procs), [p.returncode for p in procs]
    assert all(not json_lines(o) for o, _ in outs)
    assert any("'sharded' was requested but is not in force" in e.decode() for _, e in outs)
    procs, outs = run(["--allow-demotion"])
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-3000:]
    line = json.loads(json_lines(outs[0][0])[0])
    assert "dp2 (bf16)" in line["config"]["parallelism"]
    c = line["config"]["collectives"]
    assert c["requested_mode"] == "sharded" and c["mode_in_force"] == "bf16" and c["world_size"] == 2 and c["backend"] == "gloo"


def _single_process_reference(device, mode, n_samples, fuse_min_numel, steps=3):
    from predict_pv_yield_amd.models.conv3d.model import Model
    from predict_pv_yield_amd.optim import HipAdam
    from tests.ddp_two_rank_worker import SMALL
    old = HipAdam.FUSE_MIN_NUMEL
    HipAdam.FUSE_MIN_NUMEL = fuse_min_numel
    try:
        torch.manual_seed(518)
        model = Model(**SMALL, precision="bf16").to(device)
        opt = model.configure_optimizers()
        opt.set_large_grad_mode(mode)
        g = torch.Generator().manual_seed(7)
        sat, pv = torch.randn(n_samples, 11, 25, 16, 16, generator=g), torch.rand(n_samples, 25, 128, generator=g)
        batch = {"satellite": {"data": sat.to(device)}, "pv": {"pv_yield": pv.to(device)}}
        losses = []
        for _ in range(steps):
            opt.zero_grad(set_to_none=True)
            loss = model.training_step(batch, 0)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
    finally:
        HipAdam.FUSE_MIN_NUMEL = old
    return model, losses


@pytest.mark.parametrize("world,global_batch", [(2, 4), (8, 8)])
def test_ksharded_fc1_follows_the_f32_allreduce_and_one_process(device, tmp_path, world, global_batch):
    """fc1's COLUMNS dealt over the ranks (large_grad_mode "ksharded": activations are exchanged by two
    all-to-alls, no gradient or weight of fc1 crosses a link).  Its update is the f32 all-reduce's up to summation order (the
    partial products of the column shards are added in rank order; the weight gradient of a shard is formed over the whole
    global batch in one pass instead of as a sum of per-rank gradients), so: the first step's loss is the all-reduce run's to
    1e-6 relative, three Adam steps leave the consolidated parameters within the distance the sharded test allows its
    single-process reference, and a whole-batch single process (exact f32 gradient, one-pass fc1 backward) is as close."""
    a = _run_two_ranks("autograd", str(tmp_path / "f32.pt"), world=world, global_batch=global_batch, fuse_min_numel=100000)
    b = _run_two_ranks("ksharded", str(tmp_path / "ksharded.pt"), world=world, global_batch=global_batch, fuse_min_numel=100000)
    assert a["mode"] == "autograd" and b["mode"] == "ksharded" and b["world"] == world
    assert abs(a["losses"][0] - b["losses"][0]) <= 1e-6 * abs(a["losses"][0])
    for la, lb in zip(a["losses"], b["losses"]):
        assert abs(la - lb) <= 2e-3 * abs(la), (a["losses"], b["losses"])
    model, losses = _single_process_reference(device, "fused", global_batch, 100000)
    # (rank 0's loss is the mean over ITS samples only: the single process is compared through the parameters)
    for ref_state, what in ((a["state"], "f32 all-reduce"), ({k: v.cpu() for k, v in model.state_dict().items()}, "one process")):
        for k, v in ref_state.items():
            d = (v - b["state"][k]).abs()
            whole_step = (d > 5e-4).float().mean().item()
            assert whole_step <= 0.02 and d.mean().item() <= 0.3 * 5e-4, (what, k, whole_step, d.mean().item())
    assert b["state"]["fc1.weight"].shape == a["state"]["fc1.weight"].shape          # consolidated: every column is there
    assert torch.isfinite(b["exp_avg_fc1"]).all() and b["exp_avg_fc1"].abs().sum() > 0


def test_ksharded_rccl_branches_on_one_rank(device, tmp_path):
    """backend "nccl" (= RCCL), one rank: all_to_all_single / all_gather_into_tensor of the K-sharded exchange really run; with
    one rank they are the identity and the shard is the whole matrix, so the run follows the single-process one-pass fc1 backward:
    the same forward kernel (first loss bit for bit), then the weight gradient from the register-tiled kernel inside Adam's pass
    instead of the matrix-core tile of the one-pass kernel (another f32 summation order: 1e-6 on the losses)."""
    a = _run_one_rank_rccl("ksharded", str(tmp_path / "ksharded1.pt"), fuse_min_numel=100000)
    assert a["backend"] == "nccl" and a["world"] == 1 and a["mode"] == "ksharded"
    model, losses = _single_process_reference(device, "fused", 4, 100000)
    assert a["losses"][0] == losses[0]
    for la, lb in zip(a["losses"], losses):
        assert abs(la - lb) <= 2e-6 * abs(lb), (a["losses"], losses)
    for k, v in model.state_dict().items():
        d = (v.cpu() - a["state"][k]).abs()
        assert (d > 5e-4).float().mean().item() <= 0.005 and d.mean().item() <= 0.05 * 5e-4, (k, d.mean().item())


def test_bench_times_every_exchange_mode_in_one_invocation(device):
    """bench.py --gpus N (N > 1) times the requested exchange of fc1 AND the others in the same invocation --
    two gloo ranks on the one GPU here, the full 128 M-parameter fc1, default flags (--grad-sync auto: K-sharded first): the
    line's `value` is the mode in force's and says so, `grad_sync_modes` carries the row-sharded form, the bf16 all-reduce and the
    f32 all-reduce with their own ms_per_step, collectives and bytes."""
    import json
    root = os.path.dirname(HERE)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PV_DIST_BACKEND="gloo", PV_SINGLE_DEVICE="1", PV_DIST_TIMEOUT_S="900")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "4", "--steps", "1",
                                       "--warmup", "1", "--no-cpu-baseline"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=root))
    outs = [p.communicate(timeout=1500) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-3000:]
    lines = [ln for ln in outs[0][0].decode().splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1
    line = json.loads(lines[0])
    modes = line["grad_sync_modes"]
    assert set(modes) == {"sharded", "ksharded", "bf16", "autograd"}, modes
    assert "requested" in modes["ksharded"]["status"] and modes["ksharded"]["value"] == line["value"]
    assert "dp2 (ksharded)" in line["config"]["parallelism"]
    c = line["config"]["collectives"]
    assert c["requested_mode"].startswith("auto") and c["mode_in_force"] == "ksharded"
    for m in ("sharded", "bf16", "autograd"):
        assert modes[m]["status"] == "ok" and modes[m]["value"] > 0 and modes[m]["collectives"]["mode_in_force"] == m, (m, modes[m])
    # the line's roofline object under N > 1: the conv forward / dgrad family of the mode in force, bracketed on rank 0 while both
    # ranks took the same six extra steps
    r = line["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0 < r["frac"] < 1 and r["launches_per_step"] == 6, r
    assert "all_conv_frac" in r and line["cpu_baseline"] is None
    k = 128 * 1003520
    assert modes["sharded"]["exchange_bytes_per_rank_and_step"]["reduce_scatter_gradient_bf16"] == k        # (W - 1) / W of 2 bytes each
    assert modes["ksharded"]["exchange_bytes_per_rank_and_step"]["all_to_all_activations_bf16"] == 4 * 1003520


def test_bench_keeps_the_requested_modes_line_when_another_mode_hangs(device):
    """The leg that times the OTHER exchanges has never met RCCL with N > 1; a rank that never enters one of its collectives
    must not cost the run its line: with the last rank held back in front of the bf16 all-reduce (PV_BENCH_HANG_IN_MODE) the
    leg's timer prints the requested mode's line with what was recorded so far and every rank exits 0."""
    import json
    root = os.path.dirname(HERE)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PV_DIST_BACKEND="gloo", PV_SINGLE_DEVICE="1", PV_DIST_TIMEOUT_S="900",
                   PV_BENCH_HANG_IN_MODE="bf16")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "4", "--steps", "1",
                                       "--warmup", "1", "--no-roofline", "--no-cpu-baseline", "--other-modes-budget", "60"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=root))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-3000:]
    lines = [ln for ln in outs[0][0].decode().splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1 and not [ln for ln in outs[1][0].decode().splitlines() if '"metric"' in ln]
    line = json.loads(lines[0])
    modes = line["grad_sync_modes"]
    assert line["value"] > 0 and "requested" in modes["ksharded"]["status"] and modes["ksharded"]["value"] == line["value"]
    assert modes["sharded"]["status"] == "ok", modes            # timed before the hang
    assert "no answer within" in modes["bf16"]["status"] and "autograd" not in modes, modes


@pytest.mark.parametrize("m,n,k,gate", [(256, 128, 128 * 37, True), (72, 16, 128 * 5 + 64, False), (512, 128, 128 * 9, True), (32, 64, 256, False)])
def test_kshard_one_pass_backward_kernel_against_the_separate_kernels(device, m, n, k, gate):
    """pv_linear_wgrad_dx_adam_tall_bf16 (the K-sharded fc1's whole backward on this rank's column shard: gradient over all m rows
    of the global batch + Adam + dx in one pass, row blocks of 32) against the kernels it replaces: pv_linear_bwd_bf16 (dx, gated
    by x > 0) and pv_linear_wgrad_adam_bf16 on g scaled by 1 / world.  Since round 6 the gradient tile is formed on the matrix
    cores from g split EXACTLY into three bf16 terms: the same products, f32 sums in another order -- the moments agree to 1e-6 of
    their scale (and against a float64 gradient), parameters to an ulp or two, the bf16 operand copy but for rare one-ulp
    flips; dx (another tiling of the same bf16 hi + lo products) within one bf16 ulp, with identical zeros where x gates it;
    ragged m, a k that is no multiple of the 128-column tile, n below 128."""
    from predict_pv_yield_amd import hip_ops as K
    g = torch.Generator(device=device).manual_seed(m + n + k)
    x = torch.randn(m, k, generator=g, device=device).relu().to(torch.bfloat16)           # a ReLU output: zeros gate dx
    dy = torch.randn(m, n, generator=g, device=device) * 1e-2
    w0 = torch.randn(n, k, generator=g, device=device) * 0.05
    m0, v0 = torch.randn(n, k, generator=g, device=device) * 1e-3, torch.rand(n, k, generator=g, device=device) * 1e-6
    scale, step = 0.125, 3
    # reference: the separate kernels
    wa, ma, va = w0.clone(), m0.clone(), v0.clone()
    sha = K.cast_f32_to_bf16(wa)
    dx_ref, _, _ = K.linear_bwd_bf16(x, sha, dy, None, need_dx=True, need_dw=False, gate_dx_by_x=gate, need_db=False)
    K.linear_wgrad_adam_bf16(x, K.scale_f32(dy, scale), None, wa, ma, va, sha, step)
    # one pass
    wb, mb_, vb = w0.clone(), m0.clone(), v0.clone()
    shb = K.cast_f32_to_bf16(wb)
    dx = K.linear_wgrad_dx_adam_tall_bf16(x, dy, wb, mb_, vb, shb, step, grad_scale=scale, need_dx=True, gate_dx_by_x=gate)
    torch.cuda.synchronize()
    assert torch.equal(dx == 0, dx_ref == 0)
    assert float((dx.float() - dx_ref.float()).abs().max()) <= 2.0 ** -7 * float(dx_ref.float().abs().max())
    # the first moment IS the gradient: m1 = m0 + 0.1 (g - m0); against float64 products of the same operands
    g64 = (dy.double() * scale).t() @ x.double()
    m_ref = m0.double() + 0.1 * (g64 - m0.double())
    g_scale = float(g64.abs().max())
    assert float((mb_.double() - m_ref).abs().max()) <= 2e-6 * max(g_scale, 1e-3), float((mb_.double() - m_ref).abs().max())
    assert float((ma - mb_).abs().max()) <= 2e-6 * max(g_scale, 1e-3)
    assert float((va - vb).abs().max()) <= 1e-5 * float(va.abs().max())
    assert float((wa - wb).abs().max()) <= 1e-7 and float((wa == wb).float().mean()) > 0.9, float((wa - wb).abs().max())
    sh_diff = (sha.float() - shb.float()).abs()
    assert float((sh_diff > 0).float().mean()) < 1e-3 and float(sh_diff.max()) <= 2.0 ** -7 * float(sha.float().abs().max())
    assert not torch.equal(wb, w0)
