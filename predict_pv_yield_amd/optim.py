"""HipAdam — torch.optim.Adam(lr=5e-4) (predict_pv_yield/models/base_model.py:255-257) stepped by the fused
HIP kernel pv_adam_step_f32 (one pass over p, g, m, v; torch's single-tensor Adam order of operations).

Contract of the default single-process mode ("fused", FUSE_DX_INTO_UPDATE): fc1's weight, its bf16 operand copy and both
moments are updated INSIDE loss.backward() (the one-pass fc1 backward); optimizer.step() then steps the other tensors.
A backward() without a following step() therefore still changes fc1, a second backward() before step() raises, and
gradient clipping / accumulation need HipAdam(fuse_large_linear=False).  INTEGRATION.md states this first.

It is a torch.optim.Optimizer so `configure_optimizers()` keeps the Lightning contract and state_dict()
round-trips (`exp_avg`, `exp_avg_sq`, `step`: the same keys torch.optim.Adam uses, so checkpoints
interchange).  Parameters that carry a `_pv_bf16_shadow` (fc1 in the bf16 path) get the shadow rewritten in
the same pass.  `grad_scale` folds the 1/world_size of a summing gradient all-reduce into the update.
"""
import torch

from . import hip_ops as K


# Single process, "fused" mode: fc1's whole backward (dx, db, weight gradient, Adam) runs as ONE pass over the matrix,
# launched from inside backward (functional.LinearBF16) -- the parameter is therefore updated when backward() returns and
# step() only counts it.  A second backward before step() would differentiate against already-updated weights: it raises.
# tools/ab_step.py flips this switch to time the two arrangements in one process.
FUSE_DX_INTO_UPDATE = True

# The one-pass fc1 backward keeps exp_avg / exp_avg_sq of its matrix TILE BY TILE ([K/128][N][128], hip_ops.moments_to_tiled):
# a workgroup's share of each moment array is one contiguous 64 KB block instead of 128 segments 4 MB apart (the row-major
# form left the HBM-bound pass at the mercy of how the arrays' rows fell onto the DRAM banks: 690-780 us by device and
# placement; tiled 680-730 us).  The layout is private to that pass: state_dict(), a change of the gradient mode and every
# other update path see row-major tensors again (HipAdam.moments / _moments_rows).
TILE_LARGE_MOMENTS = True


def _k_to_reference(t, c):
    from .models.conv3d._fc1_layout import to_reference
    return to_reference(t, c)


def _k_to_channels_last(t, c):
    from .models.conv3d._fc1_layout import to_channels_last
    return to_channels_last(t, c)


class HipAdam(torch.optim.Optimizer):
    # 2-D parameters at least this large whose gradient comes from functional.LinearBF16 (fc1) can bypass autograd's
    # f32 .grad:  "fused" -- wgrad + Adam in one pass, the gradient is never materialised (single process);
    #             "bf16"  -- gradient written as bf16 and all-reduced in bf16 (data parallel);
    #             "autograd" -- plain f32 .grad.
    #             "sharded" -- bf16 gradient, reduce-scatter over rows, Adam on this rank's rows, all-gather of the operand copy;
    #             "ksharded" -- fc1's COLUMNS dealt over the ranks: activations are exchanged, no weight or gradient is.
    FUSE_MIN_NUMEL = 1 << 22

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, fuse_large_linear=True, overlap_large_update=False,
                 capturable=False):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("HipAdam: invalid hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.grad_scale = 1.0
        # capturable: the step counter and the bias-correction scalars live in device memory and are advanced by a
        # one-thread kernel, so that forward + backward + step can be captured in ONE HIP graph and replayed
        # (graphs.GraphedTrainStep); the by-value form freezes the step at capture.  One parameter group, every parameter
        # stepping together.
        self.capturable = bool(capturable)
        self._layout_frozen = False
        self._dev_scalars = None
        self._dev_step = None
        self._dev_fresh = False
        # fused mode only (opt-in): launch the big layer's wgrad+Adam pass from inside backward on a side stream; step()
        # then only waits for it.  Measured on MI355X (B=32): no gain -- the conv kernels own a CU's whole register file
        # and LDS, so the streaming pass cannot co-reside with them and the two still take turns
        self.overlap_large_update = overlap_large_update
        self._side_stream = None
        self._inflight = []
        self.set_large_grad_mode("fused" if fuse_large_linear else "autograd")

    def _advance_device_scalars(self, device) -> torch.Tensor:
        """The device-side scalars of the CURRENT optimiser step (advanced once per step, by whoever needs them first: the
        fused fc1 backward inside backward(), or step())."""
        self._ensure_device_state(device)
        if not self._dev_fresh:
            g = self.param_groups[0]
            K.adam_scalars_advance(self._dev_scalars, self._dev_step, lr=g["lr"], betas=g["betas"], eps=g["eps"])
            self._dev_fresh = True
        return self._dev_scalars

    def _ensure_device_state(self, device) -> None:
        """Creates the device-side counter from the host-side step counts.  Called BEFORE any host counter of the current
        step is incremented (top of the fused fc1 backward, top of step())."""
        if not self.capturable or self._dev_scalars is not None:
            return
        if len(self.param_groups) != 1:
            raise RuntimeError("HipAdam(capturable=True) takes one parameter group")
        steps = [int(st["step"].item()) for st in self.state.values() if "step" in st]
        self._dev_scalars = torch.zeros(8, dtype=torch.float32, device=device)
        self._dev_step = torch.full((1,), max(steps) if steps else 0, dtype=torch.int32, device=device)

    def device_step(self) -> int:
        """Steps taken so far as counted on the device (capturable mode; synchronises)."""
        return int(self._dev_step.item()) if self._dev_step is not None else 0

    def large_params(self):
        return [p for g in self.param_groups for p in g["params"] if p.dim() == 2 and p.numel() >= self.FUSE_MIN_NUMEL]

    def set_large_grad_mode(self, mode: str) -> None:
        if mode not in ("fused", "bf16", "sharded", "ksharded", "autograd"):
            raise ValueError(mode)
        if getattr(self, "_layout_frozen", False) and mode != getattr(self, "large_grad_mode", mode):
            raise RuntimeError("HipAdam.set_large_grad_mode(): a captured HIP graph holds this optimiser's state tensors and the "
                               "gradient mode it was captured with -- release the graph first (GraphedTrainStep.close())")
        if getattr(self, "large_grad_mode", None) in ("sharded", "ksharded") and mode != self.large_grad_mode:
            # (collective, a no-op unless sharded steps were taken) the row / column shards hold the truth: back into the full
            # tensors before another mode reads them
            self.consolidate_sharded()
            for p in self.large_params():
                p._pv_kshard = None
        if mode == "ksharded":
            from . import distributed as D
            if self.capturable or not D.is_distributed() or any(
                    D.column_shard(p.shape[1], multiple=K.MOMENT_TILE) is None for p in self.large_params()):
                mode = "sharded"    # columns do not divide over the ranks (or nothing to shard over): the row-sharded exchange
        if mode == "sharded":
            from . import distributed as D
            if any(D.row_shard(p.shape[0]) is None for p in self.large_params()):
                mode = "bf16"       # rows do not divide over the ranks: plain all-reduce
        self.large_grad_mode = mode
        self._moments_rows()
        for p in self.large_params():
            if mode == "ksharded" and getattr(p, "_pv_kshard", None) is None:
                self._make_column_shard(p)
            p._pv_grad_mode = mode
            p._pv_kshard_pending = None
            p._pv_kshard_backward = self._make_kshard_backward(p) if mode == "ksharded" else None
            p._pv_pending = None
            p._pv_pending_f32 = None
            p._pv_takes_f32_pending = mode == "fused" and not self.capturable
            p._pv_grad_bf16 = None
            p._pv_grad_shard = None
            p._pv_eager_update = self._make_eager_update(p) if (mode == "fused" and self.overlap_large_update) else None
            p._pv_fused_backward = self._make_fused_backward(p) if (mode == "fused" and not self.overlap_large_update) else None
            p._pv_applied = False

    def _make_column_shard(self, p) -> None:
        """large_grad_mode "ksharded": this rank's COLUMNS of p as contiguous tensors of their own -- f32 master, bf16 operand
        copy, both moments -- cut from the full parameter / optimiser state (replicated at this point: after the initial
        broadcast, a checkpoint load or consolidate_sharded()).  From here on the full tensors are stale until
        consolidate_sharded(); the kernels only ever see the shard, an ordinary dense [N, K / W] matrix."""
        from . import distributed as D
        k0, k1 = D.column_shard(p.shape[1], multiple=K.MOMENT_TILE)
        st = self.state.get(p) or {}
        with torch.no_grad():
            w = p.detach()[:, k0:k1].contiguous()
            ks = {"k0": k0, "k1": k1, "w": w, "shadow": K.cast_f32_to_bf16(w),
                  "exp_avg": (st["exp_avg"][:, k0:k1].contiguous() if "exp_avg" in st else torch.zeros_like(w)),
                  "exp_avg_sq": (st["exp_avg_sq"][:, k0:k1].contiguous() if "exp_avg_sq" in st else torch.zeros_like(w)),
                  "step": int(st["step"].item()) if "step" in st else 0}
        p._pv_kshard = ks

    def consolidate_sharded(self) -> None:
        """After sharded steps every rank holds current f32 values (parameter, exp_avg, exp_avg_sq) only for the rows
        ("sharded") or columns ("ksharded") it owns.  All-gather them so that state_dict() / checkpoints are complete on
        every rank.  Collective: call on all ranks."""
        if not getattr(self, "_sharded_dirty", False):
            return
        self._moments_rows()
        from . import distributed as D
        if D.is_distributed():
            for p in self.large_params():
                ks = getattr(p, "_pv_kshard", None)
                if getattr(p, "_pv_grad_mode", None) == "ksharded" and ks is not None:
                    st = self._init_state(p)
                    with torch.no_grad():
                        D.all_gather_columns(ks["w"], p.data)
                        D.all_gather_columns(ks["exp_avg"], st["exp_avg"])
                        D.all_gather_columns(ks["exp_avg_sq"], st["exp_avg_sq"])
                        st["step"].fill_(float(ks["step"]))
                    if hasattr(p, "_pv_bf16_shadow"):
                        p._pv_bf16_shadow = None      # (the full operand copy, if any mode built one earlier, is stale)
                    continue
                if getattr(p, "_pv_grad_mode", None) != "sharded":
                    continue
                st = self.state.get(p, {})
                for t in (p.data, st.get("exp_avg"), st.get("exp_avg_sq")):
                    if t is not None:
                        D.all_gather_rows(t, async_op=False)
        self._sharded_dirty = False

    # ---- layout of the large matrix's moments --------------------------------------------------------------------------
    # The flag travels ON the state tensor (`exp_avg._pv_tiled`), so copies of the optimiser (deepcopy) and replaced state
    # (load_state_dict) cannot disagree with what the bytes actually hold.
    def _is_tiled(self, p) -> bool:
        st = self.state.get(p)
        return bool(st) and getattr(st.get("exp_avg"), "_pv_tiled", False)

    def _moments_tiled(self, p) -> bool:
        """Puts exp_avg / exp_avg_sq of p into the tile layout (if the shape allows); True when they are tiled afterwards."""
        if self._layout_frozen:
            # a captured HIP graph holds the state tensors' addresses and the kernel argument that names their layout: whatever
            # the eager steps before the capture left (tiled, normally) stays
            return self._is_tiled(p)
        if not TILE_LARGE_MOMENTS or p.dim() != 2 or p.shape[1] % K.MOMENT_TILE:
            self._moments_rows(p)
            return False
        if self._is_tiled(p):
            return True
        st = self._init_state(p)
        c = self._k_channels(p)
        for key in ("exp_avg", "exp_avg_sq"):
            st[key] = K.moments_to_tiled(st[key])
            st[key]._pv_tiled = True
        if c:
            st["exp_avg"]._pv_k_channels = c
        return True

    def freeze_layout(self, frozen: bool = True) -> None:
        """graphs.GraphedTrainStep: from the capture until the graph is released the state tensors of this optimiser are never
        replaced (a replay writes to the addresses it recorded).  state_dict() / moments() then hand out row-major COPIES of
        tiled moments instead of converting them in place; anything that would have to convert in place raises."""
        # a counter: two GraphedTrainSteps on one optimiser each freeze once; the layout thaws when the last one is released
        self._freeze_count = max(0, getattr(self, "_freeze_count", 0) + (1 if frozen else -1))
        self._layout_frozen = self._freeze_count > 0

    def _moments_rows(self, p=None) -> None:
        """Back to row-major (torch's layout) for p, or for every parameter: before anything but the one-pass backward
        reads or writes the moments."""
        if self._layout_frozen:
            if any(self._is_tiled(q) for q in ([p] if p is not None else self._params_in_order())):
                raise RuntimeError("HipAdam: a captured HIP graph (graphs.GraphedTrainStep) holds this optimiser's state tensors; "
                                   "the operation would replace the tiled moments of the large matrix -- release the graph first "
                                   "(GraphedTrainStep.close())")
            return
        for q in ([p] if p is not None else [q for g in self.param_groups for q in g["params"]]):
            if self._is_tiled(q):
                st = self.state[q]
                c = self._k_channels(q)
                for key in ("exp_avg", "exp_avg_sq"):
                    st[key] = K.moments_to_rows(st[key])
                if c:
                    st["exp_avg"]._pv_k_channels = c

    def moments(self, p):
        """(exp_avg, exp_avg_sq) of p in torch's row-major layout and the reference's column order (copies when the stored
        layout is tiled, or when p's columns are stored channels-last: models/conv3d/_fc1_layout.py)."""
        st = self.state[p]
        if self._is_tiled(p):
            m, v = K.moments_to_rows(st["exp_avg"]), K.moments_to_rows(st["exp_avg_sq"])
        else:
            m, v = st["exp_avg"], st["exp_avg_sq"]
        c = self._k_channels(p)
        return (_k_to_reference(m, c), _k_to_reference(v, c)) if c else (m, v)

    def _k_channels(self, p) -> int:
        """C when p's columns are stored channels-last (models/conv3d/_fc1_layout.py), else 0.  The mark lives on the Parameter;
        a copy of it travels on exp_avg (like `_pv_tiled`): copy.deepcopy(optimizer) creates fresh Parameter objects without
        Python attributes, but copies a plain tensor's."""
        c = int(getattr(p, "_pv_k_channels", 0) or 0)
        st = self.state.get(p)
        ea = st.get("exp_avg") if st else None
        if ea is not None:
            if c:
                ea._pv_k_channels = c
            else:
                c = int(getattr(ea, "_pv_k_channels", 0) or 0)
        return c

    def _params_in_order(self):
        return [p for g in self.param_groups for p in g["params"]]      # the index torch's state_dict() gives each parameter

    def state_dict(self):
        if not self._layout_frozen:
            self._moments_rows()
        if self.capturable and self._dev_step is not None:
            # graph replays advance only the DEVICE counter: it is the truth, the host-side `step` entries are brought up to
            # it so that a checkpoint resumes with the right bias corrections
            n = float(self.device_step())
            for st in self.state.values():
                if "step" in st:
                    st["step"].fill_(n)
        sd = super().state_dict()
        # a parameter whose columns are stored channels-last leaves with its moments in the reference's column order (the
        # packed state holds the live dicts: replaced by copies, never edited)
        for i, p in enumerate(self._params_in_order()):
            c = self._k_channels(p)
            tiled = self._is_tiled(p)        # (only under a frozen layout: the live tensors stay tiled, the dict gets copies)
            if (c or tiled) and i in sd["state"]:
                st = dict(sd["state"][i])
                for key in ("exp_avg", "exp_avg_sq"):
                    if key in st:
                        t = K.moments_to_rows(st[key]) if tiled else st[key]
                        st[key] = _k_to_reference(t, c) if c else t
                sd["state"][i] = st
        return sd

    def load_state_dict(self, state_dict):
        if self._layout_frozen:
            # super().load_state_dict() REPLACES the state tensors; a captured graph would keep replaying on the old addresses
            # (with moments_tiled = 1 baked into its kernel arguments) and silently ignore the loaded state.  While a graph
            # lives, the loaded values are copied INTO the tensors it holds instead, in the layout they have.
            self._load_state_in_place(state_dict)
            return
        super().load_state_dict(state_dict)
        for p in self._params_in_order():
            c = self._k_channels(p)
            st = self.state.get(p)
            if c and st:
                for key in ("exp_avg", "exp_avg_sq"):
                    if key in st:
                        st[key] = _k_to_channels_last(st[key], c).contiguous()
                st["exp_avg"]._pv_k_channels = c
        if self.capturable:
            steps = [int(st["step"].item()) for st in self.state.values() if "step" in st]
            if self._dev_step is not None:
                # in place: a captured graph holds this tensor's address
                self._dev_step.fill_(max(steps) if steps else 0)
            self._dev_fresh = False

    def _load_state_in_place(self, state_dict) -> None:
        params = self._params_in_order()
        packed = state_dict["state"]
        if len(state_dict["param_groups"]) != len(self.param_groups) or \
                sum(len(g["params"]) for g in state_dict["param_groups"]) != len(params):
            raise ValueError("HipAdam.load_state_dict(): the state dict's parameter groups do not match this optimiser's")
        with torch.no_grad():
            for i, p in enumerate(params):
                src = packed.get(i)
                if src is None:
                    continue
                st = self._init_state(p)
                c, tiled = self._k_channels(p), self._is_tiled(p)
                for key in ("exp_avg", "exp_avg_sq"):
                    t = src[key].to(device=st[key].device, dtype=st[key].dtype)
                    if c:
                        t = _k_to_channels_last(t, c).contiguous()
                    if tiled:
                        t = K.moments_to_tiled(t)
                    st[key].copy_(t)
                st["step"].fill_(float(src["step"]))
        for g, sg in zip(self.param_groups, state_dict["param_groups"]):
            g.update({k: v for k, v in sg.items() if k != "params"})
        if self.capturable:
            steps = [int(st["step"].item()) for st in self.state.values() if "step" in st]
            if self._dev_step is not None:
                self._dev_step.fill_(max(steps) if steps else 0)
            self._dev_fresh = False

    def _group_of(self, p):
        for g in self.param_groups:
            if any(q is p for q in g["params"]):
                return g
        raise KeyError("parameter not owned by this optimiser")

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            self._k_channels(p)      # copies the parameter's layout mark onto exp_avg
        return st

    def _make_eager_update(self, p):
        def eager(x, dy, y):
            from .functional import bf16_shadow_of
            group = self._group_of(p)
            with torch.no_grad():
                st = self._init_state(p)
                self._moments_rows(p)
                shadow = bf16_shadow_of(p)
                st["step"] += 1
                main = torch.cuda.current_stream(p.device)
                if self._side_stream is None:
                    self._side_stream = torch.cuda.Stream(device=p.device)
                ready = main.record_event()            # dx (which reads the pre-update weights) is queued before this
                self._side_stream.wait_event(ready)
                with torch.cuda.stream(self._side_stream):
                    K.linear_wgrad_adam_bf16(x, dy, y, p, st["exp_avg"], st["exp_avg_sq"], shadow, int(st["step"].item()),
                                             lr=group["lr"], betas=group["betas"], eps=group["eps"])
                    done = self._side_stream.record_event()
                self._inflight.append((done, (x, dy, y)))   # keep the operands alive until step() has waited
        return eager

    def _make_fused_backward(self, p):
        def fused(x, dy, y, need_dx, need_db, gate_dx=False):
            """(dx, db) of the layer, with the Adam update of p applied in the same pass; None if the shape is not covered
            (the caller then takes the two-kernel path) or the switch is off."""
            from .functional import bf16_shadow_of
            if not FUSE_DX_INTO_UPDATE or not need_dx or not K.fused_dx_update_supported(x.shape[0], p.shape[0], p.shape[1],
                                                                                         self.capturable):
                return None
            if p._pv_applied:
                raise RuntimeError("HipAdam: backward() ran twice without optimizer.step() in between; the fused fc1 pass has "
                                   "already applied the first update (gradient accumulation needs HipAdam(fuse_large_linear="
                                   "False))")
            group = self._group_of(p)
            with torch.no_grad():
                self._ensure_device_state(p.device)
                st = self._init_state(p)
                tiled = self._moments_tiled(p)
                st["step"] += 1
                if x.shape[0] > 32:
                    # more than 32 rows (a per-GPU batch of 64, the one-GPU form of a global batch of 512): the row-block form of the
                    # same pass -- the gated output gradient as operand fragments, gradient tile on the matrix cores
                    g = K.relu_gate_f32(dy, y) if y is not None else dy
                    db = K.colsum(g) if need_db else None
                    dx = K.linear_wgrad_dx_adam_tall_bf16(x, g, p, st["exp_avg"], st["exp_avg_sq"], bf16_shadow_of(p),
                                                          int(st["step"].item()), lr=group["lr"], betas=group["betas"],
                                                          eps=group["eps"], grad_scale=1.0, need_dx=True, gate_dx_by_x=gate_dx,
                                                          moments_tiled=tiled)
                    out = (dx, db) if need_db else dx
                elif self.capturable:
                    out = K.linear_wgrad_dx_adam_dev_bf16(x, dy, y, p, st["exp_avg"], st["exp_avg_sq"], bf16_shadow_of(p),
                                                          self._advance_device_scalars(p.device), need_db=need_db,
                                                          gate_dx_by_x=gate_dx, moments_tiled=tiled)
                else:
                    out = K.linear_wgrad_dx_adam_bf16(x, dy, y, p, st["exp_avg"], st["exp_avg_sq"], bf16_shadow_of(p),
                                                      int(st["step"].item()), lr=group["lr"], betas=group["betas"],
                                                      eps=group["eps"], need_dx=True, need_db=need_db, gate_dx_by_x=gate_dx,
                                                      moments_tiled=tiled)
            p._pv_applied = True
            return out if need_db else (out, None)
        return fused

    def _make_kshard_backward(self, p):
        def one_pass(x_cols, g_all, need_dx, gate_dx):
            """K-sharded fc1: dx_cols, with the Adam update of this rank's column shard applied in the same pass
            (pv_linear_wgrad_dx_adam_tall_bf16); None if the shape is not covered (the caller then parks the pair for step())."""
            ks = p._pv_kshard
            if not need_dx or not K.kshard_one_pass_supported(ks["w"].shape[0], ks["w"].shape[1]):
                return None
            if p._pv_applied:
                raise RuntimeError("HipAdam: backward() ran twice without optimizer.step() in between; the K-sharded fc1 pass has "
                                   "already applied the first update (gradient accumulation needs another large_grad_mode)")
            group = self._group_of(p)
            with torch.no_grad():
                ks["step"] += 1
                dx = K.linear_wgrad_dx_adam_tall_bf16(x_cols, g_all, ks["w"], ks["exp_avg"], ks["exp_avg_sq"], ks["shadow"], ks["step"],
                                                      lr=group["lr"], betas=group["betas"], eps=group["eps"],
                                                      grad_scale=self.grad_scale, need_dx=True, gate_dx_by_x=gate_dx)
            p._pv_applied = True
            self._sharded_dirty = True
            return dx
        return one_pass

    def zero_grad(self, set_to_none: bool = True):
        """Also drops what a backward() parked on the large matrices for a step() that never came (a skipped step would otherwise
        keep the activations alive and apply a stale gradient later)."""
        for p in self.large_params():
            p._pv_pending = None
            p._pv_pending_f32 = None
            p._pv_grad_bf16 = None
            p._pv_grad_shard = None
            p._pv_kshard_pending = None
        return super().zero_grad(set_to_none=set_to_none)

    def _wait_inflight(self):
        if self._inflight:
            main = torch.cuda.current_stream()
            for done, _ in self._inflight:
                main.wait_event(done)
            self._inflight = []

    def set_fuse_large_linear(self, enabled: bool) -> None:
        self.set_large_grad_mode("fused" if enabled else "autograd")

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._wait_inflight()
        if self.capturable:
            first = next((p for g in self.param_groups for p in g["params"] if p.is_cuda), None)
            if first is not None:
                self._ensure_device_state(first.device)
        for p in self.large_params():
            p._pv_applied = False          # the update of this step was applied from inside backward
        for group in self.param_groups:
            plain = {}      # step count -> [(param, grad, exp_avg, exp_avg_sq, bf16 shadow)]: one multi-tensor launch each
            stepped = []
            for p in group["params"]:
                pending = getattr(p, "_pv_pending", None)
                pending32 = getattr(p, "_pv_pending_f32", None)
                gb = getattr(p, "_pv_grad_bf16", None)
                gs = getattr(p, "_pv_grad_shard", None)
                kpend = getattr(p, "_pv_kshard_pending", None)
                if kpend is not None and p.grad is None:
                    # K-sharded fc1: the gradient of THIS RANK'S COLUMNS over the whole global batch, formed and applied in
                    # one pass over the shard's p / m / v (no gradient tensor, no exchange: functional.LinearBF16KSharded
                    # parked the exchanged activations and the all-gathered output gradients)
                    x_cols, g_full = kpend
                    p._pv_kshard_pending = None
                    ks = p._pv_kshard
                    ks["step"] += 1
                    if self.grad_scale != 1.0:
                        g_full = K.scale_f32(g_full, self.grad_scale)
                    K.linear_wgrad_adam_bf16(x_cols, g_full, None, ks["w"], ks["exp_avg"], ks["exp_avg_sq"], ks["shadow"],
                                             ks["step"], lr=group["lr"], betas=group["betas"], eps=group["eps"])
                    self._sharded_dirty = True
                    continue
                if p.grad is None and pending is None and pending32 is None and gb is None and gs is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("HipAdam steps parameters on the MI355X only (no CPU path)")
                st = self._init_state(p)
                self._moments_rows(p)       # (a no-op unless an earlier step of p went through the one-pass backward)
                st["step"] += 1
                if self.capturable and (pending is not None or gs is not None or gb is not None):
                    raise NotImplementedError("HipAdam(capturable=True) covers the single-process paths (fc1's one-pass "
                                              "backward + the multi-tensor step); not the deferred / exchanged fc1 gradients")
                if pending32 is not None and p.grad is None:
                    # the f32 model's fc1 (functional.LinearF32): gradient from the f32 activations + Adam in one pass
                    x, g = pending32
                    p._pv_pending_f32 = None
                    K.linear_wgrad_adam_f32(x, g, None, p, st["exp_avg"], st["exp_avg_sq"], int(st["step"].item()),
                                            lr=group["lr"], betas=group["betas"], eps=group["eps"])
                    if getattr(p, "_pv_bf16_shadow", None) is not None:
                        p._pv_bf16_shadow = None      # (a bf16 operand copy from an earlier bf16 forward is stale now)
                    continue
                if pending is not None and p.grad is None:
                    from .functional import bf16_shadow_of
                    x, dy, y = pending
                    p._pv_pending = None
                    K.linear_wgrad_adam_bf16(x, dy, y, p, st["exp_avg"], st["exp_avg_sq"], bf16_shadow_of(p),
                                             int(st["step"].item()), lr=group["lr"], betas=group["betas"], eps=group["eps"])
                    continue
                if gs is not None and p.grad is None:
                    # sharded update (data parallel): Adam over this rank's rows only, then every rank's bf16 operand
                    # rows are all-gathered; the gather is waited for by the next forward (functional.bf16_shadow_of),
                    # so it runs under the next step's conv forward.  f32 rows owned by other ranks go stale until
                    # consolidate_sharded().
                    from . import distributed as D
                    from .functional import bf16_shadow_of
                    p._pv_grad_shard = None
                    r0, r1 = D.row_shard(p.shape[0])
                    shadow = bf16_shadow_of(p)
                    K.adam_step_bf16grad(p[r0:r1], gs, st["exp_avg"][r0:r1], st["exp_avg_sq"][r0:r1],
                                         int(st["step"].item()), lr=group["lr"], betas=group["betas"], eps=group["eps"],
                                         bf16_shadow=shadow[r0:r1], grad_scale=self.grad_scale)
                    p._pv_shadow_work = D.all_gather_rows(shadow)
                    self._sharded_dirty = True
                    continue
                if gb is not None and p.grad is None:
                    from .functional import bf16_shadow_of
                    p._pv_grad_bf16 = None
                    K.adam_step_bf16grad(p, gb, st["exp_avg"], st["exp_avg_sq"], int(st["step"].item()), lr=group["lr"],
                                         betas=group["betas"], eps=group["eps"], bf16_shadow=bf16_shadow_of(p),
                                         grad_scale=self.grad_scale)
                    continue
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                stepped.append(p)
                p._pv_opt_gen = getattr(p, "_pv_opt_gen", 0) + 1      # (caches keyed on the weight's value: functional.split2_conv_weight)
                plain.setdefault(int(st["step"].item()), []).append(
                    (p, g.float(), st["exp_avg"], st["exp_avg_sq"], getattr(p, "_pv_bf16_shadow", None)))
            if self.capturable and plain:
                scalars = self._advance_device_scalars(stepped[0].device)
                K.adam_step_multi_dev([it for items in plain.values() for it in items], scalars, grad_scale=self.grad_scale)
            else:
                for step, items in plain.items():
                    K.adam_step_multi(items, step, lr=group["lr"], betas=group["betas"], eps=group["eps"],
                                      grad_scale=self.grad_scale)
            if stepped:
                from .functional import refresh_packed_conv_weights
                refresh_packed_conv_weights(stepped)
        self._dev_fresh = False        # the next backward / step advances the device-side scalars again
        return loss
