"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl" on ROCm),
gloo on CPU for tests.  The path shards by sample: flow / warp / normalise need no
collective; the only exchange step is the gradient all-reduce (the reference gets it from Lightning DDP,
experiments/003_...py:288-294) plus the scalar-metric all-reduce of `sync_dist=True` (base_model.py:117).

Gradients are all-reduced as ONE flat bucket per dtype: fc1's gradient is 99.9 % of the bytes, so
bucketing finer than "everything" only adds launches; xGMI is point-to-point, so one large collective
lets RCCL use all 7 links at once.  The sum is left un-averaged: HipAdam folds 1/world_size into the
update (grad_scale), saving a pass over 0.5 GB.
"""
import datetime
import os
import sys
import threading
from typing import Dict

import torch
import torch.distributed as dist


def is_distributed() -> bool:
    """True when gradients / metrics must be exchanged.  PV_DIST_SINGLE_RANK=1 also counts an initialised ONE-rank group:
    every collective of the N > 1 path then really runs (RCCL reduce_scatter_tensor / all_gather_into_tensor / all_reduce
    on one GPU) -- how tests/test_gpu_ddp.py executes the "nccl" branches on the single-GPU test box."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("PV_DIST_SINGLE_RANK") == "1"


def collective_timeout_s() -> float:
    """Upper bound on any one collective (PV_DIST_TIMEOUT_S, default 300 s instead of torch's 10 / 30 minutes).  A failed
    RCCL exchange usually shows as a HANG, not as an exception: with this bound the process-group watchdog aborts the rank
    (non-zero exit, torchrun then takes the job down) instead of leaving the job stuck."""
    return float(os.environ.get("PV_DIST_TIMEOUT_S", "300"))


def init_from_env(backend: str = None, force: bool = False) -> bool:
    """Initialise from torchrun's RANK / WORLD_SIZE / MASTER_* environment; returns True if world_size > 1.
    `force` initialises a one-rank group too (the RCCL code paths can then be exercised on a single GPU)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not force:
        return False
    if not dist.is_initialized():
        if backend is None:
            # PV_DIST_BACKEND=gloo: debugging aid (e.g. two ranks sharing one GPU, where RCCL refuses to start)
            backend = os.environ.get("PV_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_device_index())
        dist.init_process_group(backend=backend, rank=int(os.environ.get("RANK", "0")), world_size=world,
                                timeout=datetime.timedelta(seconds=collective_timeout_s()))
    return world > 1


def local_device_index() -> int:
    """cuda index of this rank: LOCAL_RANK, or 0 for every rank when PV_SINGLE_DEVICE=1 (debugging on a 1-GPU box)."""
    if os.environ.get("PV_SINGLE_DEVICE") == "1":
        return 0
    return int(os.environ.get("LOCAL_RANK", "0"))


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """DDP's initial parameter broadcast from rank 0."""
    if not is_distributed():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)
        if hasattr(t, "_pv_bf16_shadow"):
            del t._pv_bf16_shadow  # stale after the broadcast; rebuilt lazily


def all_reduce_gradients(module: torch.nn.Module, average: bool = False) -> None:
    """Sum (or average) gradients over ranks through one flat bucket per dtype/device."""
    if not is_distributed():
        return
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    if not grads:
        return
    buckets = {}
    for g in grads:
        buckets.setdefault((g.dtype, g.device), []).append(g)
    for (_, _), gs in buckets.items():
        big = max(gs, key=lambda g: g.numel())
        small = [g for g in gs if g is not big]
        # the dominant tensor (fc1.weight.grad, ~0.5 GB) is reduced in place; the rest share one flat buffer
        work = dist.all_reduce(big, op=dist.ReduceOp.SUM, async_op=True)
        if small:
            flat = torch.cat([g.reshape(-1) for g in small])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            off = 0
            for g in small:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
        work.wait()
    if average:
        w = dist.get_world_size()
        for g in grads:
            g.div_(w)


def row_shard(n_rows: int, rank: int = None, world: int = None):
    """Equal row shard [r0, r1) of a matrix for the sharded update of the big layer; None if rows % world != 0."""
    if rank is None:
        rank = dist.get_rank() if is_distributed() else 0
    if world is None:
        world = dist.get_world_size() if is_distributed() else 1
    if n_rows % world != 0:
        return None
    per = n_rows // world
    return rank * per, (rank + 1) * per


def shard_indices(n_items: int, rank: int = None, world: int = None):
    """Item indices of this rank's loader shard, DistributedSampler's rule without shuffling (what Lightning's
    replace_sampler_ddp installs, configs/trainer/all_params.yaml:43): the index list is PADDED by wrapping around to
    ceil(n_items / world) * world entries and rank r takes entries r, r + world, ...  Every rank gets the same count (the
    loops issue collectives per step / per logged metric, so no rank may run short) and no item is dropped; a split with
    fewer items than ranks (the shipped experiments validate on n_val_data = 2 batches) repeats items instead of failing."""
    if rank is None:
        rank = dist.get_rank() if is_distributed() else 0
    if world is None:
        world = dist.get_world_size() if is_distributed() else 1
    if n_items <= 0:
        return []
    per = -(-n_items // world)
    return [(rank + k * world) % n_items for k in range(per)]


def reduce_scatter_rows(full: torch.Tensor):
    """Sum `full` [N, K] over ranks and keep only this rank's row shard: (shard [N/W, K], async work).  RCCL:
    one reduce_scatter (each rank receives 1/W of the bytes of an all-reduce's second half); gloo (tests): all-reduce +
    slice, same values."""
    r0, r1 = row_shard(full.shape[0])
    if dist.get_backend() == "nccl":
        shard = torch.empty((r1 - r0,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
        work = dist.reduce_scatter_tensor(shard, full, op=dist.ReduceOp.SUM, async_op=True)
        return shard, work
    work = dist.all_reduce(full, op=dist.ReduceOp.SUM, async_op=True)
    return full[r0:r1], work


_GATHER_STAGING = {}


def _gather_staging(full: torch.Tensor, rows: int) -> torch.Tensor:
    """One persistent staging buffer per (matrix, shard shape): the all-gather's input must not alias its output, but
    it does not have to be a fresh allocation every step (it was: 1/W of the matrix cloned per call)."""
    key = (full.data_ptr(), rows, tuple(full.shape[1:]), full.dtype, str(full.device))
    buf = _GATHER_STAGING.get(key)
    if buf is None:
        if len(_GATHER_STAGING) > 16:
            _GATHER_STAGING.clear()
        buf = torch.empty((rows,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
        _GATHER_STAGING[key] = buf
    return buf


def all_gather_rows(full: torch.Tensor, async_op: bool = True):
    """Every rank contributes its own row shard of `full` [N, K] (already written in place) and receives the others."""
    r0, r1 = row_shard(full.shape[0])
    mine = _gather_staging(full, r1 - r0)
    mine.copy_(full[r0:r1])          # 1/W of the matrix; keeps the collective's input distinct from its output
    if dist.get_backend() == "nccl":
        return dist.all_gather_into_tensor(full, mine, async_op=async_op)
    world = dist.get_world_size()
    per = r1 - r0
    chunks = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(chunks, mine, async_op=False)
    for r, c in enumerate(chunks):
        full[r * per:(r + 1) * per].copy_(c)
    return None


# ---- K-sharded fc1 (HipAdam large_grad_mode "ksharded"; VERDICT r5 item 3, SURVEY section 8e "alternative noted") ---------------
# The conv tower stays data-parallel; fc1's COLUMNS (its 1 003 520 input features) are dealt over the ranks, rank r owning
# columns [r K/W, (r+1) K/W) of the weight, its moments and its bf16 operand copy for ALL samples of the global batch.  Per step
# a rank exchanges activations instead of weights: its samples' last-layer activation goes out by one all-to-all ([B, K/W]
# slices, forward), the input gradient comes back by another (backward), plus a reduce-scatter of the [W B, N] partial outputs
# and an all-gather of the [B, N] output gradients -- 2 x 56 MB per rank and step at B = 32, W = 8 instead of the 2 x 257 MB of
# the row-sharded exchange, no weight or gradient of fc1 ever crosses a link, and fc1's traffic (3.47 GB of p / m / v per step
# on one GPU) divides by W.  The reference's DDP all-reduce (experiments/003_...py:292-293) has no counterpart for this layer.
def column_shard(n_cols: int, rank: int = None, world: int = None, multiple: int = 8):
    """Equal column shard [k0, k1) of a matrix; None if the columns do not divide into shards of a multiple of `multiple`."""
    if rank is None:
        rank = dist.get_rank() if is_distributed() else 0
    if world is None:
        world = dist.get_world_size() if is_distributed() else 1
    if n_cols % world != 0 or (n_cols // world) % multiple != 0:
        return None
    per = n_cols // world
    return rank * per, (rank + 1) * per


def _swap01(t: torch.Tensor) -> torch.Tensor:
    """[n0, n1, seg] -> [n1, n0, seg], contiguous: the library's segment copy on the device when the segments allow it (16-byte
    multiples), else torch's strided copy."""
    if t.is_cuda and t.is_contiguous() and (t.shape[2] * t.element_size()) % 16 == 0:
        from . import hip_ops as K
        return K.swap01_segments(t)
    return t.transpose(0, 1).contiguous()


def all_to_all_columns(x_local: torch.Tensor) -> torch.Tensor:
    """x_local [B, K] (this rank's samples, all columns) -> [W B, K/W] (ALL samples in rank order, this rank's columns).
    RCCL: one all_to_all_single of W chunks [B, K/W] (the chunk-major staging copy is the only extra pass); gloo (tests):
    all-gather + slice, same bytes in the same places."""
    world = dist.get_world_size()
    b, k = x_local.shape
    kr = k // world
    if dist.get_backend() == "nccl":
        send = _swap01(x_local.contiguous().view(b, world, kr))                  # [W, B, K/W]: chunk s = my samples, rank s's columns
        recv = torch.empty_like(send)                                            # chunk s = rank s's samples, my columns
        dist.all_to_all_single(recv, send)
        return recv.view(world * b, kr)
    r = dist.get_rank()
    full = [torch.empty_like(x_local) for _ in range(world)]
    dist.all_gather(full, x_local.contiguous())
    return torch.cat([t[:, r * kr:(r + 1) * kr] for t in full], dim=0).contiguous()


def all_to_all_rows_back(dx_cols: torch.Tensor) -> torch.Tensor:
    """The inverse exchange: dx_cols [W B, K/W] (all samples, this rank's columns) -> [B, K] (this rank's samples, all columns)."""
    world = dist.get_world_size()
    wb, kr = dx_cols.shape
    b = wb // world
    if dist.get_backend() == "nccl":
        send = dx_cols.contiguous().view(world, b, kr)                           # chunk s = rank s's samples, my columns
        recv = torch.empty_like(send)                                            # chunk s = my samples, rank s's columns
        dist.all_to_all_single(recv, send)
        return _swap01(recv).view(b, world * kr)
    r = dist.get_rank()
    full = [torch.empty_like(dx_cols) for _ in range(world)]
    dist.all_gather(full, dx_cols.contiguous())
    return torch.cat([t[r * b:(r + 1) * b] for t in full], dim=1).contiguous()


def reduce_scatter_sample_rows(partial: torch.Tensor) -> torch.Tensor:
    """partial [W B, N] (every rank's partial sums for ALL samples) -> the summed rows of this rank's samples [B, N].
    The sum runs in RANK ORDER on every backend (all-gather of the W slices this rank needs, then a fixed-order add): the
    K-sharded forward must not depend on a collective's internal reduction order -- 2 x the bytes of a reduce-scatter on a
    [W B, 128] f32 matrix (128 KB at B = 32, W = 8) buys bit-reproducibility across backends and runs."""
    world = dist.get_world_size()
    wb, n = partial.shape
    b = wb // world
    if dist.get_backend() == "nccl":
        send = partial.contiguous().view(world, b, n)
        recv = torch.empty_like(send)                    # chunk s = rank s's partial sums for MY samples
        dist.all_to_all_single(recv, send)
    else:
        r = dist.get_rank()
        full = [torch.empty_like(partial) for _ in range(world)]
        dist.all_gather(full, partial.contiguous())
        recv = torch.stack([t[r * b:(r + 1) * b] for t in full])
    if recv.is_cuda:
        from . import hip_ops as K      # pv_colsum_f32 over the W slices: partial sums added in rank order
        return K.colsum(recv.view(world, b * n)).view(b, n)
    out = recv[0].clone()               # (CPU tensors: the gloo tests of the exchange itself)
    for s_ in range(1, world):
        out += recv[s_]
    return out


def all_gather_sample_rows(local: torch.Tensor) -> torch.Tensor:
    """local [B, N] (this rank's samples) -> [W B, N] (all samples, rank order)."""
    world = dist.get_world_size()
    local = local.contiguous()
    if dist.get_backend() == "nccl":
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local)
        return out
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local)
    return torch.cat(parts, dim=0)


def all_gather_columns(shard: torch.Tensor, full: torch.Tensor) -> None:
    """Every rank's column shard [N, K/W] into `full` [N, K] (checkpoints: HipAdam.consolidate_sharded)."""
    world = dist.get_world_size()
    kr = shard.shape[1]
    parts = [torch.empty_like(shard) for _ in range(world)]
    dist.all_gather(parts, shard.contiguous())
    for r, t in enumerate(parts):
        full[:, r * kr:(r + 1) * kr].copy_(t)


class OverlappedGradSync:
    """DDP-style overlap: the all-reduce of a LARGE parameter's gradient is launched from an autograd hook the
    moment that gradient is final, on RCCL's own stream, and is waited for only before the optimiser step.
    fc1.weight.grad (99.9 % of the bytes) is produced FIRST in backward (the head runs before the conv stack), so its
    ~0.5 GB exchange hides under the conv dgrad/wgrad kernels.  The many small gradients go as one flat bucket at
    the end.  Sums are left un-averaged (HipAdam.grad_scale = 1/world folds the mean into the update)."""

    def __init__(self, module: torch.nn.Module, large_numel: int = 1 << 22):
        self.module = module
        self.large = [p for p in module.parameters() if p.requires_grad and p.numel() >= large_numel]
        self.small = [p for p in module.parameters() if p.requires_grad and p.numel() < large_numel]
        self._pending = []
        self._handles = []
        self._flat = None          # the small-gradient bucket, laid out once (finish())
        self._flat_key = None
        if is_distributed():
            for p in self.large:
                self._handles.append(p.register_post_accumulate_grad_hook(self._launch))
                # side channel of functional.LinearBF16 (HipAdam large_grad_mode "bf16"): the bf16 gradient tensor is
                # handed over the moment the wgrad kernel is enqueued
                p._pv_on_grad = self._launch_tensor

    def _launch(self, p: torch.Tensor) -> None:
        if p.grad is not None:  # None when the gradient travels through the bf16 side channel instead
            self._pending.append(dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, async_op=True))

    def _launch_tensor(self, g: torch.Tensor, param: torch.Tensor = None) -> None:
        if param is not None and getattr(param, "_pv_grad_mode", None) == "sharded":
            # ZeRO-1 for the big layer: this rank only needs the summed gradient of the rows it owns
            shard, work = reduce_scatter_rows(g)
            param._pv_grad_shard = shard
            self._pending.append(work)
            return
        self._pending.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self) -> None:
        """Call after backward, before optimizer.step()."""
        if not is_distributed():
            return
        params = [p for p in self.small if p.grad is not None]
        if params:
            ref = params[0].grad
            for p in [q for q in params if q.grad.dtype != ref.dtype or q.grad.device != ref.device]:
                dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)       # odd ones out travel alone
            params = [q for q in params if q.grad.dtype == ref.dtype and q.grad.device == ref.device]
            # one flat bucket, segments aligned to 16 bytes; after the all-reduce every .grad BECOMES its segment of the
            # bucket (a view), so nothing is copied back and the multi-tensor Adam reads the bucket directly
            key = (tuple(id(p) for p in params), ref.dtype, ref.device)
            if self._flat is None or self._flat_key != key:
                # laid out once: the bucket and its per-parameter views persist across steps (the padding between
                # segments is zeroed here and never written again)
                offs, total = [], 0
                for p in params:
                    offs.append(total)
                    total += (p.grad.numel() + 3) // 4 * 4
                self._flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
                self._views = [self._flat[o:o + p.grad.numel()].view_as(p.grad) for o, p in zip(offs, params)]
                self._flat_key = key
            flat, views = self._flat, self._views
            src = [p.grad for p in params]
            stale = [(v, g) for v, g in zip(views, src) if g.data_ptr() != v.data_ptr()]
            if stale:       # a gradient that already IS its bucket segment (accumulated in place) needs no copy
                torch._foreach_copy_([v for v, _ in stale], [g for _, g in stale])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            for p, v in zip(params, views):
                p.grad = v
        for w in self._pending:
            w.wait()
        self._pending = []

    def remove(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles = []
        for p in self.large:
            if hasattr(p, "_pv_on_grad"):
                del p._pv_on_grad


_FALLBACKS = {"ksharded": ("ksharded", "sharded", "bf16", "autograd"), "sharded": ("sharded", "bf16", "autograd"),
              "bf16": ("bf16", "autograd"), "autograd": ("autograd",)}


def negotiate_grad_sync(model: torch.nn.Module, optimizer, batch, mode: str, allow_demotion: bool = True) -> str:
    """Collective.  Tries the requested gradient-exchange mode of the big layer with ONE untimed train step in THIS
    process -- "sharded" (reduce-scatter -> row-sharded Adam -> all-gather of the bf16 operand copy), "bf16" (one bf16
    all-reduce), "autograd" (f32 all-reduce of .grad) -- and, if the step raises on any rank, moves every rank to the
    next simpler mode together (the outcome is agreed with an all-reduce, parameters are re-broadcast).  Returns the
    mode in force.  Exits non-zero with a clear message if none works; never re-executes the process (the GPU is
    initialised: an exec would take the node down).
    allow_demotion=False (bench.py without --allow-demotion): ANY outcome other than the requested mode -- a failed trial
    step, or an optimiser that silently chose a simpler mode (rows that do not divide over the ranks) -- ends every rank
    with a non-zero status instead: a scaling number measured on another exchange than the one asked for is not that number.
    PV_DIST_FAIL_MODES=<mode>[,<mode>] makes the trial step of those modes raise (test hook for exactly this rule)."""
    if mode not in _FALLBACKS:
        raise ValueError(mode)
    if not is_distributed():
        optimizer.set_large_grad_mode(mode)
        return optimizer.large_grad_mode
    dev = next(model.parameters()).device
    flag_dev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
    errors = []
    def all_ok(ok: int) -> bool:
        agreed = torch.tensor([ok], dtype=torch.int32, device=flag_dev)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        return int(agreed.item()) == 1

    def trial_watchdog(candidate: str):
        """A trial step that HANGS (the usual face of an RCCL failure) cannot be demoted: no rank can tell the others.
        After PV_TRIAL_TIMEOUT_S (default 120 s) this rank prints why and exits with status 3 -- os._exit, never an exec
        (the GPU is initialised) -- and torchrun ends the job."""
        limit = float(os.environ.get("PV_TRIAL_TIMEOUT_S", "120"))
        done = threading.Event()

        def watch():
            if not done.wait(limit):
                print(f"[predict_pv_yield_amd] rank {dist.get_rank()}: the trial step of gradient exchange mode "
                      f"'{candidate}' did not finish within {limit:.0f} s (hung collective?); exiting with status 3. "
                      f"Re-run with a simpler mode (Trainer(large_grad_mode='autograd')) or a longer PV_TRIAL_TIMEOUT_S.",
                      file=sys.stderr, flush=True)
                os._exit(3)
        threading.Thread(target=watch, daemon=True).start()
        return done

    for candidate in _FALLBACKS[mode]:
        ok, sync = 1, None
        # phase 1, local: switching the optimiser over.  Agreed on BEFORE any collective of the trial step is issued, so a
        # rank that cannot even set the mode does not leave the others inside a gradient exchange it never joins
        try:
            optimizer.set_large_grad_mode(candidate)
        except Exception as e:      # noqa: BLE001
            ok = 0
            errors.append(f"{candidate}: {type(e).__name__}: {e}")
        if all_ok(ok):
            # phase 2: one whole train step with the real collectives
            finished = trial_watchdog(candidate)
            try:
                if candidate in os.environ.get("PV_DIST_FAIL_MODES", "").split(","):
                    raise RuntimeError(f"PV_DIST_FAIL_MODES: injected failure of mode '{candidate}'")
                sync = OverlappedGradSync(model)
                optimizer.zero_grad(set_to_none=True)
                model.training_step(batch, 0).backward()
                sync.finish()
                optimizer.step()
                if dev.type == "cuda":
                    torch.cuda.synchronize(dev)
            except Exception as e:      # noqa: BLE001 -- any failure of the trial step demotes the mode
                ok = 0
                errors.append(f"{candidate}: {type(e).__name__}: {e}")
            finally:
                if sync is not None:
                    sync.remove()
            ok = 1 if all_ok(ok) else 0
            finished.set()
        else:
            ok = 0
        if ok and (allow_demotion or optimizer.large_grad_mode == mode):
            return optimizer.large_grad_mode
        if not allow_demotion:
            why = (errors[-1] if errors else "on another rank") if not ok else \
                f"the optimiser chose '{optimizer.large_grad_mode}' (rows of the large matrix do not divide over the ranks)"
            raise SystemExit(f"predict_pv_yield_amd: gradient exchange mode '{mode}' was requested but is not in force ({why}); "
                             f"refusing to continue on a simpler mode (pass --allow-demotion / allow_demotion=True to accept it)")
        if dist.get_rank() == 0:
            print(f"[predict_pv_yield_amd] gradient exchange mode '{candidate}' failed on at least one rank"
                  f" ({errors[-1] if errors else 'on another rank'}); trying the next simpler mode", flush=True)
        # drop whatever the failed attempt left behind, then make the replicas identical again
        for p in optimizer.large_params():
            p._pv_pending = p._pv_grad_bf16 = p._pv_grad_shard = None
            p._pv_kshard_pending = None
            p._pv_shadow_work = None
            if hasattr(p, "_pv_bf16_shadow"):
                del p._pv_bf16_shadow
        optimizer._sharded_dirty = False
        broadcast_parameters(model)
    raise SystemExit("predict_pv_yield_amd: no gradient-exchange mode works on this job (" + "; ".join(errors) + ")")


def all_reduce_mean_scalars(values: Dict[str, float], device=None) -> Dict[str, float]:
    """All logged scalars of one log_dict call travel as ONE vector (the reference sends one tiny all-reduce each)."""
    if not is_distributed():
        return values
    keys = sorted(values)
    dev = device if (device is not None and dist.get_backend() == "nccl") else torch.device("cpu")
    t = torch.tensor([values[k] for k in keys], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    t /= dist.get_world_size()
    return {k: float(v) for k, v in zip(keys, t.tolist())}


def shard_range(n_items: int, rank: int = None, world: int = None):
    """Contiguous shard [lo, hi) of a GLOBAL batch of independent samples for this rank (bench.py --global-batch, the
    strong-scaling mode; tests/ddp_two_rank_worker.py): no collective needed.  EVERY rank gets the same count,
    n_items // world (a tail n_items % world is dropped, DistributedSampler's drop_last): the train loop issues
    collectives every step, so a rank with fewer samples per step than the others is fine but one with none is not --
    fewer items than ranks is an error, not an empty shard."""
    if rank is None:
        rank = dist.get_rank() if is_distributed() else 0
    if world is None:
        world = dist.get_world_size() if is_distributed() else 1
    per = n_items // world
    if per == 0:
        raise ValueError(f"shard_range: {n_items} item(s) cannot be split over {world} ranks (every rank needs >= 1)")
    return rank * per, (rank + 1) * per
