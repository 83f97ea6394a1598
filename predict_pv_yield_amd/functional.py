"""torch.autograd.Function wrappers over the HIP kernels: the operator boundary of SURVEY.md §8b.

Reference operators replaced (all third-party torch in the reference):
  F.relu(nn.Conv3d(...)(x))   predict_pv_yield/models/conv3d/model.py:117-120
  F.relu(nn.Linear(...)(x))   model.py:125-126,151-152
  (y_hat - y).abs().mean()    predict_pv_yield/models/base_model.py:99
Two numeric modes:
  "fp32": reference layout NCDHW, f32-accurate kernels (rtol-1e-4 parity with torch CPU): forward / dgrad on the f32 matrix
          cores (exact f32 products), the weight gradients of the 32-channel layers and fc1's products as split half-float /
          bf16 terms on the 16-bit matrix cores (f32-accurate to ~1e-6, not bit-exact f32).  PV_EXACT_F32=1 in the environment
          puts every one of those products back on the f32 kernels (exact_f32() below; pv_gemm reads the same variable);
  "bf16": NDHWC bf16 activations, MFMA kernels with f32 accumulation (throughput path).
"""
import ctypes

import torch

from . import hip_ops as K
from ._lib import check, current_stream_ptr, get_lib, ptr


# ---------------------------------------------------------------------------------------------
# fp32 path (Conv3D: Conv3dGeneralF32 below -- one set of f32 kernels for every kernel extent / stride / padding)
# ---------------------------------------------------------------------------------------------
LINEAR_F32_GEMM_K = 1 << 16     # the same threshold as pv_linear_fwd_f32 (dense_f32.hip GEMM_K)
F32_PENDING_MAX_ROWS = 1024     # pv_linear_wgrad_adam_f32 keeps m x 16 gradient values in 64 KB of LDS


def exact_f32() -> bool:
    """PV_EXACT_F32=1: the precision="fp32" path keeps every product on the f32 kernels (conv weight gradient on the f32 matrix
    cores, fc1 on the f32 tile kernels / pv_gemm's f32 matrix instruction) instead of the split 16-bit forms."""
    import os
    return bool(os.environ.get("PV_EXACT_F32"))


class LinearF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        x = x.contiguous()
        if (not exact_f32() and x.dtype == torch.float32 and weight.is_contiguous()
                and K.linear_f32_skinny_covers(x.shape[0], weight.shape[0], x.shape[1])):
            # fc1-sized: one stream over the weight, exact f32 products on the f32 matrix instruction
            y = K.linear_fwd_f32_skinny(x, weight, bias.contiguous() if bias is not None else None, relu)
        else:
            y = K.linear_fwd_f32(x, weight.contiguous(), bias.contiguous() if bias is not None else None, relu)
        ctx.save_for_backward(x, weight, y if relu else None)
        ctx.has_bias = bias is not None
        ctx.weight_param = weight
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        # (the GEMM's limits: its column tiles ride grid.y -- 65 535 x 64 input features -- and rows x features in 31 bits)
        if (x.shape[1] >= LINEAR_F32_GEMM_K and dy.numel() % 4 == 0 and not exact_f32() and x.shape[1] <= 65535 * 64
                and x.shape[0] * x.shape[1] < 2 ** 31):
            # fc1-sized: the two products on the matrix cores at f32 accuracy (pv_gemm_f32), 650 us against 790 us
            g = K.relu_gate_f32(dy, y) if y is not None else dy.contiguous()
            dx = None
            if ctx.needs_input_grad[0]:
                if weight.is_contiguous() and K.linear_f32_skinny_covers(g.shape[0], g.shape[1], weight.shape[1]):
                    dx = K.linear_dx_f32_skinny(g.contiguous(), weight)
                else:
                    dx = K.gemm(g, weight)
            owner = ctx.weight_param
            earlier = getattr(owner, "_pv_pending_f32", None)
            if (getattr(owner, "_pv_grad_mode", "autograd") == "fused" and getattr(owner, "_pv_takes_f32_pending", False)
                    and x.shape[1] % 8 == 0 and x.shape[0] <= F32_PENDING_MAX_ROWS and x.is_contiguous() and earlier is None
                    and owner.grad is None):
                # HipAdam owns this parameter (single process): it forms the gradient inside its pass over p / m / v
                # (pv_linear_wgrad_adam_f32): the 0.5 GB gradient is neither written nor read back
                owner._pv_pending_f32 = (x, g)
                dw = None
            else:
                if earlier is not None:
                    # a second backward before step() (gradient accumulation, several losses): the pair parked by the first one
                    # becomes an ordinary gradient now, and this call's joins it through autograd's accumulation
                    x0, g0 = earlier
                    owner._pv_pending_f32 = None
                    dw0 = K.gemm(g0.t(), x0)
                    owner.grad = dw0 if owner.grad is None else owner.grad + dw0
                dw = K.gemm(g.t(), x)
            return dx, dw, (K.colsum(g) if ctx.has_bias else None), None
        dx, dw, db = K.linear_bwd_f32(x, weight.contiguous(), dy.contiguous(), y, need_dx=ctx.needs_input_grad[0])
        return dx, dw, (db if ctx.has_bias else None), None


def linear_f32(x, weight, bias, relu=False):
    return LinearF32.apply(x, weight, bias, relu)


# ---------------------------------------------------------------------------------------------
# bf16 path
# ---------------------------------------------------------------------------------------------
# The conv towers hand 1-bit relu masks from each layer's forward to the next layer's dgrad (see Conv3dReLUBF16): the dgrad
# epilogue reads one byte per lane instead of 16 of the bf16 activation -- 0.31 GB per step at B = 32, a tenth of what the conv
# launches move.  Round 2 measured the mask path as a LOSS (the forwards wrote their masks in a pass of their own, 7-8 us
# each, and the gate read hit the 256 MB Infinity Cache anyway); since round 6 the input-stationary forward and the
# first-layer kernel write the mask with the tile (one byte store per lane and tile row in the side slots the gated form uses
# for its gate loads) and the gated dgrad reads it in-kernel: same-box A/B (tools/ab_step.py, tools/probes/step_ab_libs.sh)
# -3 to -4 us per step at B = 32 and B = 64 -- the time barely moves because these launches are not bound by their bytes
# (profiles/r06/NOTES.md section 5), the traffic does.  Bit-identical either way (tests/test_gpu_conv.py).
USE_RELU_MASKS = True


class PackInputBF16(torch.autograd.Function):
    """x[B,C,T,H,W] f32 -> NDHWC bf16 (channel-padded).  The satellite input needs no gradient."""

    @staticmethod
    def forward(ctx, x):
        out = K.pack_ncdhw_f32_to_ndhwc_bf16(x.contiguous())
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, g):
        return None


def packed_conv_weight(weight: torch.Tensor, transpose_flip: bool) -> torch.Tensor:
    """MFMA-fragment (bf16) image of a conv weight for the forward (transpose_flip=False) or dgrad operator, cached on
    the parameter.  HipAdam re-packs every cached image in ONE launch right after it updates the weights; any other
    in-place change bumps `weight._version` and invalidates the cache here."""
    cache = getattr(weight, "_pv_packed", None)
    if cache is None or cache.get("version") != weight._version or cache.get("device") != weight.device:
        cache = {"version": weight._version, "device": weight.device}
        weight._pv_packed = cache
    wp = cache.get(transpose_flip)
    if wp is None:
        wp = K.conv3d_pack_weight_bf16(weight.detach().contiguous(), transpose_flip=transpose_flip)
        cache[transpose_flip] = wp
    return wp


def split2_conv_weight(weight: torch.Tensor):
    """(fragments, scale state) of a conv weight's two-term half-float split (pv_conv3d_pack_weight_split2_f16), cached on the
    parameter: valid while neither torch (`_version`) nor HipAdam (`_pv_opt_gen`, bumped by every step that touches the
    parameter -- its kernels write through raw pointers) has changed the weight.  A train step re-packs once, as before; a
    no-grad scoring loop (validation: 16 forwards per 1 024 samples) no longer repeats the pack, its maximum and row-sum passes
    and their allocations on every call."""
    key = (weight._version, getattr(weight, "_pv_opt_gen", 0), weight.device)
    cache = getattr(weight, "_pv_split2", None)
    # (while a HIP graph is being captured the pack must be part of it: a replay updates the weights without passing here)
    if cache is None or cache[0] != key or (weight.is_cuda and torch.cuda.is_current_stream_capturing()):
        cache = (key, K.conv3d_pack_weight_split2_f16(weight.detach().contiguous()))
        weight._pv_split2 = cache
    return cache[1]


def refresh_packed_conv_weights(params) -> None:
    """Re-pack, in one launch, every cached fragment image of the given (just updated) conv weights."""
    jobs = []
    for p in params:
        cache = getattr(p, "_pv_packed", None)
        if cache is None or cache.get("version") != p._version or cache.get("device") != p.device or not p.is_contiguous():
            continue
        jobs += [(p.detach(), cache[flip], flip) for flip in (False, True) if flip in cache]
    if jobs:
        K.conv3d_pack_weights_multi(jobs)


class Conv3dReLUBF16(torch.autograd.Function):
    """xp [B,T,H,W,CPAD] bf16 -> y bf16 NDHWC [B,To,Ho,Wo,32] (or NCDHW [B,Co,To,Ho,Wo] when y_ncdhw), plus -- when
    want_relu_mask -- the 1-bit relu mask of y (int32 [B,To,Ho,Wo]) that the NEXT layer hands back as `x_relu_mask`:
    its dgrad epilogue gates dx with 4 bytes per voxel instead of re-reading the bf16 activation."""

    @staticmethod
    def forward(ctx, xp, weight, bias, c_in, padding, relu, y_ncdhw, x_is_relu_output, dy_pregated, x_relu_mask,
                want_relu_mask):
        wp = packed_conv_weight(weight, False)
        c_out = weight.shape[0]
        want_relu_mask = bool(want_relu_mask and relu and not y_ncdhw)
        out = K.conv3d_fwd_bf16(xp, None, wp, bias.contiguous() if bias is not None else None, c_in, c_out, padding, relu,
                                y_ncdhw, want_relu_mask=want_relu_mask)
        y, mask = out if want_relu_mask else (out, None)
        ctx.save_for_backward(xp, weight, y if relu else None, x_relu_mask)
        ctx.cfg = (c_in, c_out, padding, relu, y_ncdhw, bias is not None, x_is_relu_output, dy_pregated)
        if mask is None:
            mask = torch.empty(0, dtype=torch.int32, device=y.device)
        ctx.mark_non_differentiable(mask)
        ctx.set_materialize_grads(False)      # no zero-filled "gradient" for the mask output
        return y, mask

    @staticmethod
    def backward(ctx, dy, _dmask):
        xp, weight, y, x_relu_mask = ctx.saved_tensors
        c_in, c_out, padding, relu, y_ncdhw, has_bias, x_is_relu_output, dy_pregated = ctx.cfg
        dy = dy.contiguous()
        if y_ncdhw:
            # fc1 hands back the gradient in flatten (NCDHW) order: gate + transpose in one pass (transpose only when the
            # fc1 kernel already applied this layer's ReLU derivative)
            dy = K.repack_gate_ncdhw_to_ndhwc_bf16(dy, None if _take_pregated(dy) else y)
            gate = None
        elif dy_pregated == "ask":
            gate = None if _take_pregated(dy) else y   # the last layer under a channels-last fc1: gated by fc1's one-pass backward?
        elif dy_pregated:
            gate = None   # the consumer's dgrad epilogue already applied this layer's ReLU derivative
        else:
            gate = y
        dw, db = K.conv3d_bwd_weight_bf16(xp, dy, gate, c_in, c_out, padding)
        dx = None
        if ctx.needs_input_grad[0]:
            # dgrad = the forward kernel on dy with padding 2-p and mirrored, channel-swapped weights
            wpt = packed_conv_weight(weight, True)
            pad_b = tuple(2 - p for p in padding)
            # x_is_relu_output: x = relu(...) of the producer, so (x > 0) IS its ReLU derivative; applying it in this
            # kernel's epilogue hands the producer an already-gated gradient (no gate reads in its wgrad/dgrad)
            out_gate = xp if (x_is_relu_output and xp.shape[-1] == 32) else None
            gate_mask = x_relu_mask if (out_gate is not None and x_relu_mask is not None and x_relu_mask.numel()) else None
            dx = K.conv3d_fwd_bf16(dy, gate, wpt, None, c_out, c_in, pad_b, relu=False, y_ncdhw=False, out_gate=out_gate,
                                   out_gate_mask=gate_mask)
            if K.bf16_cpad(c_in) != 32:
                dx = dx[..., : K.bf16_cpad(c_in)].contiguous()
        return dx, dw, (db if has_bias else None), None, None, None, None, None, None, None, None


class Conv3dFirstLayerBF16(torch.autograd.Function):
    """First conv layer straight from the reference's f32 NCDHW input (<= 16 channels): y bf16 NDHWC, and -- when a
    weight gradient will be wanted -- the NDHWC bf16 image of the input as a by-product of the same pass (no separate
    pack kernel).  Bit-identical to PackInputBF16 + Conv3dReLUBF16."""

    @staticmethod
    def forward(ctx, x, weight, bias, padding, relu, dy_pregated, want_relu_mask):
        wp = packed_conv_weight(weight, False)
        c_out, c_in = weight.shape[0], weight.shape[1]
        need_bwd = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]   # grad mode is off inside forward()
        want_relu_mask = bool(want_relu_mask and relu and need_bwd)
        out = K.conv3d_fwd_bf16_f32in(x.contiguous(), wp, bias.contiguous() if bias is not None else None, c_out, padding,
                                      relu, want_packed=need_bwd, want_relu_mask=want_relu_mask)
        y, xp = out[0], out[1]
        mask = out[2] if want_relu_mask else torch.empty(0, dtype=torch.int32, device=y.device)
        ctx.save_for_backward(xp, weight, y if relu else None)
        ctx.cfg = (c_in, c_out, padding, bias is not None, dy_pregated)
        ctx.mark_non_differentiable(mask)
        ctx.set_materialize_grads(False)
        return y, mask

    @staticmethod
    def backward(ctx, dy, _dmask):
        xp, weight, y = ctx.saved_tensors
        c_in, c_out, padding, has_bias, dy_pregated = ctx.cfg
        gate = None if dy_pregated else y
        dw, db = K.conv3d_bwd_weight_bf16(xp, dy.contiguous(), gate, c_in, c_out, padding)
        return None, dw, (db if has_bias else None), None, None, None, None


def conv3d_first_layer_bf16(x, weight, bias, padding=(0, 0, 0), relu=True, dy_pregated=False, want_relu_mask=False):
    """Returns y, or (y, relu mask of y) when want_relu_mask."""
    y, mask = Conv3dFirstLayerBF16.apply(x, weight, bias, tuple(padding), relu, dy_pregated, want_relu_mask)
    return (y, mask) if want_relu_mask else y


def bf16_shadow_of(weight: torch.Tensor) -> torch.Tensor:
    """bf16 copy of a big f32 parameter, kept current by HipAdam (pv_adam_step_f32 writes it)."""
    work = getattr(weight, "_pv_shadow_work", None)
    if work is not None:      # all-gather of the row shards other ranks updated (HipAdam "sharded"): stream-level wait
        work.wait()
        weight._pv_shadow_work = None
    shadow = getattr(weight, "_pv_bf16_shadow", None)
    version = getattr(weight, "_pv_bf16_shadow_version", None)
    if shadow is None or shadow.device != weight.device or version != weight._version:
        shadow = torch.empty(weight.shape, dtype=torch.bfloat16, device=weight.device)
        check(get_lib().pv_cast_f32_to_bf16(ptr(weight.detach()), ptr(shadow), weight.numel(), current_stream_ptr()),
              "pv_cast_f32_to_bf16")
        weight._pv_bf16_shadow = shadow
        weight._pv_bf16_shadow_version = weight._version
    return shadow


# dx tensors that already carry their producer's ReLU derivative (keyed by storage address, valid inside one backward pass):
# the fc1 single-pass kernel can apply (x > 0) while it stores dx, and the last conv layer's backward then repacks without
# reading its activation again
_PREGATED_DX = {}


def _mark_pregated(t: torch.Tensor) -> None:
    task = torch._C._current_graph_task_id()
    if any(v != task for v in _PREGATED_DX.values()):
        _PREGATED_DX.clear()          # marks of an earlier backward pass that nobody consumed
    _PREGATED_DX[t.data_ptr()] = task


def _take_pregated(t: torch.Tensor) -> bool:
    task = _PREGATED_DX.pop(t.data_ptr(), None)
    return task is not None and task == torch._C._current_graph_task_id() and task >= 0


class LinearBF16(torch.autograd.Function):
    """fc1: x bf16 [B,K] . bf16(weight)[N,K]^T, f32 accumulate; dw/db f32, dx bf16.
    x_is_relu_output: x = relu(...) of the producing layer, so (x > 0) is that layer's ReLU derivative; the fused backward
    then applies it to dx itself and marks the tensor (see _PREGATED_DX)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, x_is_relu_output=False):
        ctx.x_is_relu_output = bool(x_is_relu_output)
        x = x.contiguous()
        wb = bf16_shadow_of(weight)
        y = K.linear_fwd_bf16(x, wb, bias.contiguous() if bias is not None else None, relu)
        ctx.save_for_backward(x, wb, y if relu else None)
        ctx.has_bias = bias is not None
        ctx.weight_param = weight
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wb, y = ctx.saved_tensors
        dy = dy.contiguous()
        weight = ctx.weight_param
        mode = getattr(weight, "_pv_grad_mode", "autograd")
        if mode == "fused":
            fused = getattr(weight, "_pv_fused_backward", None)
            out = (fused(x, dy, y, ctx.needs_input_grad[0], ctx.has_bias, gate_dx=ctx.x_is_relu_output)
                   if fused is not None else None)
            if out is not None:
                # HipAdam owns this parameter (single process): dx, db, the weight gradient and its Adam update came out of
                # ONE pass over the matrix; nothing is left for step()
                if ctx.x_is_relu_output:
                    _mark_pregated(out[0])
                return out[0], None, out[1], None, None
        # (x a ReLU output: dx leaves gated, as from the one-pass kernel, so the producing conv layer's backward takes the
        # pre-gated kernels -- in the data-parallel modes too)
        dx, dw, db = K.linear_bwd_bf16(x, wb, dy, y, need_dx=ctx.needs_input_grad[0], need_dw=(mode == "autograd"),
                                       gate_dx_by_x=ctx.x_is_relu_output)
        if ctx.x_is_relu_output and dx is not None:
            _mark_pregated(dx)
        if mode == "fused":
            # HipAdam owns this parameter (single process): hand it (x, dy, relu mask); the weight gradient is formed
            # inside the optimiser's pass over p/m/v and never written to memory
            eager = getattr(weight, "_pv_eager_update", None)
            if eager is not None:
                eager(x, dy, y)     # HBM-bound update starts now, on a side stream, under the MFMA-bound conv backward
            else:
                weight._pv_pending = (x, dy, y)
        elif mode in ("bf16", "sharded"):
            # data parallel: the gradient is written once in bf16 and handed to the gradient-sync callback right
            # away, so its exchange (99.9 % of the bytes of the step) runs under the conv backward that follows.
            # "bf16": all-reduce, every rank steps the whole matrix; "sharded": reduce-scatter, every rank steps its
            # own rows and the bf16 operand copy is all-gathered afterwards (HipAdam)
            gb = K.linear_wgrad_bf16out(x, dy, y, weight.shape[0])
            cb = getattr(weight, "_pv_on_grad", None)
            if mode == "bf16":
                weight._pv_grad_bf16 = gb
                if cb is not None:
                    cb(gb)
            elif cb is not None:
                cb(gb, weight)              # sets weight._pv_grad_shard
            else:
                weight._pv_grad_bf16 = gb   # single process: nothing to scatter
        return dx, dw, (db if ctx.has_bias else None), None, None


class LinearBF16KSharded(torch.autograd.Function):
    """fc1 with its COLUMNS dealt over the data-parallel ranks (HipAdam large_grad_mode "ksharded", distributed.py): the
    ranks exchange the last conv layer's ACTIVATIONS (one all-to-all each way) instead of fc1's gradient or weights.
      forward   x_cols = all_to_all(x) [W B, K/W]; partial = x_cols . W[:, shard]^T (the same streaming kernel, on the shard);
                y = relu(sum over ranks in rank order of the partial rows of this rank's samples + bias)
      backward  g = dy * (y > 0); g_all = all_gather(g) [W B, N]; dx_cols = g_all . W[:, shard] (gated by x > 0 when x is a
                ReLU output) -> all_to_all back [B, K]; (x_cols, g_all) are parked on the parameter: HipAdam.step() forms this
                rank's columns of the weight gradient over the WHOLE global batch inside its Adam pass over the shard.
    The reference's counterpart is DDP's all-reduce of fc1.weight.grad (experiments/003_...py:292-293): same update, up to
    f32 summation order, without moving 0.5 GB per rank and step."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, x_is_relu_output=False):
        from . import distributed as D
        ks = weight._pv_kshard
        x_cols = D.all_to_all_columns(x.contiguous())
        partial = K.linear_fwd_bf16(x_cols, ks["shadow"], None, False)
        y = K.scale_bias_relu_f32(D.reduce_scatter_sample_rows(partial), bias.contiguous() if bias is not None else None, 1.0, relu)
        ctx.save_for_backward(x_cols, y if relu else None)
        ctx.cfg = (bias is not None, bool(x_is_relu_output), x.shape[0])
        ctx.weight_param = weight
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import distributed as D
        x_cols, y = ctx.saved_tensors
        has_bias, x_is_relu_output, b_local = ctx.cfg
        weight = ctx.weight_param
        ks = weight._pv_kshard
        g = K.relu_gate_f32(dy.contiguous(), y) if y is not None else dy.contiguous()
        g_all = D.all_gather_sample_rows(g)
        one_pass = getattr(weight, "_pv_kshard_backward", None)
        dx_cols = one_pass(x_cols, g_all, ctx.needs_input_grad[0], x_is_relu_output) if one_pass is not None else None
        if dx_cols is not None:
            # dx of every sample for this rank's columns AND the Adam update of the shard came out of ONE pass over it
            dx = D.all_to_all_rows_back(dx_cols)
            if x_is_relu_output:
                _mark_pregated(dx)
            return dx, None, (K.colsum(g) if has_bias else None), None, None
        dx = None
        if ctx.needs_input_grad[0]:
            dx_cols, _, _ = K.linear_bwd_bf16(x_cols, ks["shadow"], g_all, None, need_dx=True, need_dw=False,
                                              gate_dx_by_x=x_is_relu_output, need_db=False)
            dx = D.all_to_all_rows_back(dx_cols)
            if x_is_relu_output:
                _mark_pregated(dx)
        if getattr(weight, "_pv_kshard_pending", None) is not None:
            raise RuntimeError("K-sharded fc1: backward() ran twice without optimizer.step() in between (gradient accumulation "
                               "needs another large_grad_mode)")
        weight._pv_kshard_pending = (x_cols, g_all)
        return dx, None, (K.colsum(g) if has_bias else None), None, None


def conv3d_relu_bf16(xp, weight, bias, c_in, padding=(0, 0, 0), relu=True, y_ncdhw=False, x_is_relu_output=False,
                     dy_pregated=False, x_relu_mask=None, want_relu_mask=False):
    """x_is_relu_output: xp is the ReLU output of the previous conv3d_relu_bf16 (its dgrad then gates dx itself);
    dy_pregated: the consumer of this layer's output is such a layer, so the incoming gradient is already gated;
    x_relu_mask: the 1-bit relu mask of xp the previous layer produced (want_relu_mask=True there).
    Returns y, or (y, relu mask of y) when want_relu_mask."""
    y, mask = Conv3dReLUBF16.apply(xp, weight, bias, c_in, tuple(padding), relu, y_ncdhw, x_is_relu_output, dy_pregated,
                                   x_relu_mask, want_relu_mask)
    return (y, mask) if want_relu_mask else y


def linear_bf16(x, weight, bias, relu=False, x_is_relu_output=False):
    if getattr(weight, "_pv_grad_mode", None) == "ksharded" and getattr(weight, "_pv_kshard", None) is not None:
        return LinearBF16KSharded.apply(x, weight, bias, relu, x_is_relu_output)
    return LinearBF16.apply(x, weight, bias, relu, x_is_relu_output)


F32_WGRAD_ON_F16X2 = True       # tools/ab_step-style switch: False = always the f32 matrix-core weight gradient


def _triple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (int(v),) * 3


def _wgrad_on_f16x2(x, dy, y, weight, stride) -> bool:
    """3x3x3, stride 1, 32 output channels, <= 32 input channels, dy already gated (y is None), a batch worth the six
    launches, and the split kernel's alignment (voxels per sample % 4 == 0 on both tensors)."""
    if not F32_WGRAD_ON_F16X2 or exact_f32() or y is not None or tuple(weight.shape[2:]) != (3, 3, 3) or _triple(stride) != (1, 1, 1):
        return False
    if weight.shape[0] != 32 or weight.shape[1] > 32:
        return False
    vx, vy = x.shape[2] * x.shape[3] * x.shape[4], dy.shape[2] * dy.shape[3] * dy.shape[4]
    # the split kernels' own limits (pv_conv3d_bwd_weight_bf16: batch on grid.z, 1 GiB of packed voxels per sample): beyond them
    # the general f32 kernel runs, as before the split existed
    if x.shape[0] > 65535 or vx * 64 > (1 << 30):
        return False
    return vx % 4 == 0 and vy % 4 == 0 and x.shape[0] * vy >= (1 << 16)


F32_CONV_ON_F16X2 = True        # 32 -> 32 channel 3x3x3 layers of the f32 model: forward / dgrad as three half-float launches (False: f32 matrix instruction)


def conv_f16x2_takes(batch, c_in, t, h, w, weight, stride, padding, x_requires_grad=True) -> bool:
    """Does the half-float form of the f32 Conv3d (Conv3dF32OnF16x2) take a layer of this geometry?  3x3x3, stride 1, 32 output
    channels, <= 32 input channels, a size worth three launches + a sum pass, inside the input-stationary kernel's limits
    (hip_ops.conv3d_f16x2_covers) for the forward AND the data gradient (pad = 2 - pad)."""
    if not F32_CONV_ON_F16X2 or exact_f32() or weight.dim() != 5 or tuple(weight.shape[2:]) != (3, 3, 3) or _triple(stride) != (1, 1, 1):
        return False
    p = _triple(padding)
    if weight.shape[0] != 32 or weight.shape[1] > 32 or any(q < 0 or q > 2 for q in p) or c_in != weight.shape[1]:
        return False
    if weight.shape[1] < 32 and x_requires_grad:      # (the first layer: its operand image is padded to 32 channels, its dx is not formed)
        return False
    to, ho, wo = t + 2 * p[0] - 2, h + 2 * p[1] - 2, w + 2 * p[2] - 2
    if min(to, ho, wo) <= 0 or (t * h * w) % 4 or (to * ho * wo) % 4 or batch * to * ho * wo < (1 << 16):
        return False
    return (K.conv3d_f16x2_covers(batch, 32, 32, t, h, w, p) and
            K.conv3d_f16x2_covers(batch, 32, 32, to, ho, wo, tuple(2 - q for q in p)))


def is_operand_images(x) -> bool:
    """A chained f16x2 conv's output: the two half-float operand images [2, B, T, H, W, 32] with their scale state attached."""
    return x.dtype == torch.float16 and x.dim() == 6 and x.shape[0] == 2 and x.shape[5] == 32 and hasattr(x, "_pv_state")


def _conv_on_f16x2(x, weight, stride, padding) -> bool:
    if is_operand_images(x):
        return conv_f16x2_takes(x.shape[1], 32, x.shape[2], x.shape[3], x.shape[4], weight, stride, padding)
    if x.dim() != 5 or x.dtype != torch.float32:
        return False
    return conv_f16x2_takes(x.shape[0], x.shape[1], x.shape[2], x.shape[3], x.shape[4], weight, stride, padding, x.requires_grad)


def _register_gated(t: torch.Tensor, registry: dict, value) -> None:
    """Something a gradient tensor's producer leaves for its consumer -- the largest magnitude (_GATED_MAX) or the finished
    operand images (_GATED_PLANES) of the two-term split -- keyed by storage, valid inside this backward pass (see
    ReluGateF32.backward)."""
    task = torch._C._current_graph_task_id()
    if _GATED_MAX_TASK[0] != task:
        _GATED_MAX.clear()
        _GATED_PLANES.clear()
        _GATED_MAX_TASK[0] = task
    # the entry HOLDS the tensor: while it is registered its storage cannot be handed to another gradient of the same backward
    # pass by the caching allocator, so a later tensor can never find a stale entry at "its" address; an entry
    # nobody consumes lives until the next backward pass starts (one gradient tensor kept alive, at most)
    registry[(t.data_ptr(), t.numel())] = (value, t)


def _gated_planes_of(t: torch.Tensor):
    if _GATED_MAX_TASK[0] != torch._C._current_graph_task_id():
        return None
    hit = _GATED_PLANES.pop((t.data_ptr(), t.numel()), None)
    return hit[0] if hit is not None else None


class Conv3dF32OnF16x2(torch.autograd.Function):
    """nn.Conv3d(c_in <= 32, 32, 3) + optional ReLU in float32 (reference: models/conv3d/model.py:80-90,113-120) with every
    product on the half-float matrix cores at f32 accuracy (csrc/conv3d_f16x2.hip): x and w split in two half-float terms, three
    launches per pass, one ordered-sum pass that also writes its output's OWN split.
    x: a float32 [B,C,T,H,W] tensor (split here) or the operand images of a chained producer (is_operand_images).
    chain_out: the consumer is another layer of this kind -- the output IS the pair of operand images (f16 [2,B,To,Ho,Wo,32],
    scale state attached by the wrapper), no float32 tensor is written; its gradient arrives in the same form (exactly one
    consumer).  The operand images of x serve the weight gradient, those of dy both gradients.
    x_is_relu_output / dy_pregated as in Conv3dGeneralF32."""

    @staticmethod
    def forward(ctx, x, weight, bias, padding, relu, x_is_relu_output, dy_pregated, chain_out):
        p = _triple(padding)
        ctx.x_is_images = is_operand_images(x)
        if ctx.x_is_images:
            xh, xl, xs = x[0], x[1], x._pv_state
        else:
            x = x.contiguous()
            planes = getattr(x, "_pv_planes", None)      # (xh, xl, state, x's version then) left by the sum pass that produced x
            if planes is None or planes[3] != x._version or planes[0].shape[:4] != (x.shape[0],) + tuple(x.shape[2:]):
                planes = K.pack_split2_ncdhw_f32_to_ndhwc_f16(x, cpad32=True)      # (none, or x was changed in place since)
            xh, xl, xs = planes[:3]
        wp, ws = split2_conv_weight(weight)
        y, ys, yp = K.conv3d_f32_on_f16x2(xh, xl, xs, wp[0], wp[1], ws, 32, 32, p, bias=bias.contiguous() if bias is not None else None,
                                          relu=relu, want_f32=not chain_out)
        ctx.save_for_backward(y if (relu and not dy_pregated) else None, xh, xl, xs, wp, ws)
        ctx.padding, ctx.has_bias, ctx.c_in, ctx.gate_dx, ctx.chain_out = p, bias is not None, weight.shape[1], x_is_relu_output, chain_out
        if chain_out:
            _FWD_PLANES[0] = (yp.data_ptr(), ys)
            return yp
        _FWD_PLANES[0] = (y.data_ptr(), (yp[0], yp[1], ys))
        return y

    @staticmethod
    def backward(ctx, dy):
        y, xh, xl, xs, wp, ws = ctx.saved_tensors
        dy = dy.contiguous()
        planes = None
        if ctx.chain_out:
            planes = _gated_planes_of(dy)
            if planes is None:
                raise RuntimeError("Conv3dF32OnF16x2: the gradient of a chained output (operand images) must come from exactly one "
                                   "Conv3dF32OnF16x2 consumer's backward; build the layer with chain_out=False for any other use")
        elif y is not None:
            dy, dmax = K.relu_gate_f32(dy, y, want_max=True)
        else:
            planes = _gated_planes_of(dy)
            dmax = None if planes is not None else _gated_max_of(dy)
        if planes is None:
            planes = K.pack_split2_ncdhw_f32_to_ndhwc_f16(dy, maxabs_state=dmax)
        dh, dl, ds = planes
        dx = None
        if ctx.needs_input_grad[0]:
            # (the gate: x > 0 read off x's h image -- rne_f16(x s) > 0 exactly when x > 2^-25 / s, i.e. for every x a ReLU produced
            # above 2^-39 of the tensor's largest value)
            dx, dxs, dxp = K.conv3d_f32_on_f16x2(dh, dl, ds, wp[2], wp[3], ws, 32, 32, tuple(2 - q for q in ctx.padding),
                                                 gate_h=xh if ctx.gate_dx else None, data_gradient=True, want_f32=not ctx.x_is_images)
            if ctx.x_is_images:
                dx = dxp
            _register_gated(dx, _GATED_PLANES, (dxp[0], dxp[1], dxs))
        dw, db = K.conv3d_bwd_weight_f32_from_split2(xh, xl, xs, dh, dl, ds, 32, ctx.padding)
        if ctx.c_in < 32:
            dw = dw[:, :ctx.c_in].contiguous()      # (the padded channels' gradient is zero)
        return dx, dw, (db if ctx.has_bias else None), None, None, None, None, None


_FWD_PLANES = [None]      # (data_ptr of the newest f16x2 forward's output, its operand images + state): attached to the tensor by the wrapper
_GATED_PLANES = {}


class Conv3dGeneralF32(torch.autograd.Function):
    """nn.Conv3d with kernel extents 1..3, any stride / padding (+ optional fused ReLU) on the f32 kernels (weight gradient of
    32-channel 3x3x3 layers: f32-accurate split products, see exact_f32()):
    the layers of the optical-flow notebook model (13_…ipynb:969-985) and Conv3dMaxPool's conv.
    In a conv+ReLU chain the ReLU gating of the activation gradient is moved into the PRODUCING dgrad kernel:
    x_is_relu_output -> this layer's dx leaves already zeroed where x <= 0; dy_pregated -> the incoming dy was gated
    that way by the next layer, so dgrad / wgrad read it without touching y again."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, relu, x_is_relu_output, dy_pregated):
        x = x.contiguous()
        y = K.conv3d_general_fwd_f32(x, weight.contiguous(), bias.contiguous() if bias is not None else None, stride,
                                     padding, relu)
        ctx.save_for_backward(x, weight, y if (relu and not dy_pregated) else None)
        ctx.stride, ctx.padding, ctx.has_bias, ctx.x_is_relu_output = stride, padding, bias is not None, x_is_relu_output
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = K.conv3d_general_bwd_data_f32(dy, y, weight.contiguous(), tuple(x.shape), ctx.stride, ctx.padding,
                                               x_mask=x if ctx.x_is_relu_output else None)
        if _wgrad_on_f16x2(x, dy, y, weight, ctx.stride):
            # the PV-yield model's 3x3x3 layers: three launches of the 16-bit weight-gradient kernel on two-term half-float
            # operands instead of one at the f32 matrix rate, f32-accurate (hip_ops.conv3d_bwd_weight_f32_on_f16x2)
            dw, db = K.conv3d_bwd_weight_f32_on_f16x2(x, dy, _triple(ctx.padding), dy_maxabs_state=_gated_max_of(dy))
            return dx, dw, (db if ctx.has_bias else None), None, None, None, None, None
        dw, db = K.conv3d_general_bwd_weight_f32(x, dy, y, tuple(weight.shape), ctx.stride, ctx.padding,
                                                 need_bias=ctx.has_bias)
        return dx, dw, db, None, None, None, None, None


_GATED_MAX = {}
_GATED_MAX_TASK = [-1]


def _gated_max_of(t: torch.Tensor):
    if _GATED_MAX_TASK[0] != torch._C._current_graph_task_id():
        return None
    hit = _GATED_MAX.pop((t.data_ptr(), t.numel()), None)   # consumed once
    return hit[0] if hit is not None else None


class ReluGateF32(torch.autograd.Function):
    """Identity on a ReLU output y; its backward applies that ReLU's derivative (dy where y > 0) in one streaming pass, so
    the layer that produced y can be told `dy_pregated` (its matrix-core dgrad / wgrad then read dy as it is)."""

    @staticmethod
    def forward(ctx, y):
        ctx.save_for_backward(y)
        return y.view_as(y)

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        g, state = K.relu_gate_f32(dy, y, want_max=True)
        # the gated gradient's largest magnitude, for the two-term split of the producing layer's weight gradient (keyed by the
        # tensor's storage like _PREGATED_DX, valid inside this backward pass)
        _register_gated(g, _GATED_MAX, state)
        return g


def relu_gate_f32(y):
    return ReluGateF32.apply(y)


def conv3d_general_f32(x, weight, bias, stride=1, padding=0, relu=False, x_is_relu_output=False, dy_pregated=False, chain_out=False):
    """chain_out: the ONE consumer of the result is another conv3d_general_f32 that the half-float form takes (conv_f16x2_takes):
    the result may then be the pair of operand images (is_operand_images) instead of a float32 tensor."""
    if x.is_cuda and _conv_on_f16x2(x, weight, stride, padding):
        chain = bool(chain_out) and (dy_pregated or not relu)
        y = Conv3dF32OnF16x2.apply(x, weight, bias, padding, relu, x_is_relu_output, dy_pregated, chain)
        # the sum pass's split of y travels with the tensor object: the next layer reads it instead of splitting y again
        tag, _FWD_PLANES[0] = _FWD_PLANES[0], None
        if tag is not None and tag[0] == y.data_ptr():
            if chain:
                y._pv_state = tag[1]
            else:
                y._pv_planes = tag[1] + (y._version,)
        return y
    if is_operand_images(x):
        raise RuntimeError("conv3d_general_f32: operand images (a chained half-float conv's output) reached a layer the half-float "
                           "form does not take; the producer must be called with chain_out=False")
    return Conv3dGeneralF32.apply(x, weight, bias, stride, padding, relu, x_is_relu_output, dy_pregated)


class MaxPool3dF32(torch.autograd.Function):
    """nn.MaxPool3d (perceiver_conv3d_nwp_sat.py:53-57): argmax saved as int32, backward is a deterministic gather."""

    @staticmethod
    def forward(ctx, x, kernel, stride, padding):
        y, idx, geom = K.maxpool3d_fwd_f32(x.contiguous(), kernel, stride, padding)
        ctx.save_for_backward(idx)
        ctx.geom = geom
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return K.maxpool3d_bwd_f32(dy.contiguous(), idx, ctx.geom), None, None, None


def maxpool3d_f32(x, kernel=3, stride=None, padding=0):
    return MaxPool3dF32.apply(x, kernel, stride, padding)


class MSELossF32(torch.autograd.Function):
    """F.mse_loss(y_hat, y) (13_…ipynb:1008): scalar mean, gradient 2 (y_hat - y) / n produced in the same pass."""

    @staticmethod
    def forward(ctx, y_hat, y):
        out, grad = K.mse_loss(y_hat.contiguous(), y.contiguous(), need_grad=True)
        ctx.save_for_backward(grad)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


def mse_loss(y_hat, y):
    return MSELossF32.apply(y_hat, y)


class EmbeddingF32(torch.autograd.Function):
    """nn.Embedding lookup (model_sat_nwp.py:251-260) as a gather kernel; backward = deterministic per-row sums."""

    @staticmethod
    def forward(ctx, table, ids):
        ids = ids.to(device=table.device, dtype=torch.int64).contiguous()
        ctx.save_for_backward(ids)
        ctx.n_rows = table.shape[0]
        return K.embedding_fwd(table.contiguous(), ids)

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        return K.embedding_bwd(dout.contiguous(), ids, ctx.n_rows), None


def embedding(table, ids):
    return EmbeddingF32.apply(table, ids)


# ---------------------------------------------------------------------------------------------
# loss
# ---------------------------------------------------------------------------------------------
_UNIT_GRADIENTS = {}


def unit_gradient(loss: torch.Tensor) -> torch.Tensor:
    """A cached 0-dim tensor of one on the loss's device: `loss.backward(unit_gradient(loss))` is what `loss.backward()`
    does, minus the fill kernel autograd launches for the root gradient every step -- and ForecastLosses.backward recognises
    THIS object and hands out its gradient without the multiply by one (two ~5 us launches of the 31-launch train step)."""
    key = (loss.device, loss.dtype)
    t = _UNIT_GRADIENTS.get(key)
    if t is None:
        t = torch.ones((), dtype=loss.dtype, device=loss.device)
        _UNIT_GRADIENTS[key] = t
    return t


class ForecastLosses(torch.autograd.Function):
    """Returns the four scalars (mse, nmae, mse_exp, mae_exp) as separate 0-dim tensors (views of the kernel's f32[4]
    output); only nmae carries a gradient, as in the reference where the returned loss is nmae (base_model.py:146).
    Handing them out separately keeps torch's select-backward (a zero fill and a copy, ~5 us each) out of the step."""

    @staticmethod
    def forward(ctx, y_hat, y):
        y_hat = y_hat.contiguous()
        out4, grad = K.forecast_losses(y_hat, y, need_grad=True)
        ctx.save_for_backward(grad)
        ctx.set_materialize_grads(False)  # no zero-filled gradients for the three metric outputs
        mse, nmae, mse_exp, mae_exp = out4.unbind(0)
        ctx.mark_non_differentiable(mse, mse_exp, mae_exp)
        return mse, nmae, mse_exp, mae_exp

    @staticmethod
    def backward(ctx, g_mse, g_nmae, g_mse_exp, g_mae_exp):
        (grad,) = ctx.saved_tensors
        if g_nmae is None:
            return None, None
        if any(g_nmae is u for u in _UNIT_GRADIENTS.values()):
            return grad, None          # the root gradient is the cached constant one: nothing to scale
        return grad * g_nmae, None


def forecast_losses(y_hat, y):
    return ForecastLosses.apply(y_hat, y)


def forecast_losses_with_horizons(y_hat, y):
    """Validation / test form (no gradient): ((mse, nmae, mse_exp, mae_exp), per-step mse [n], per-step mae [n]) from
    ONE launch of pv_forecast_losses_f32 (base_model.py:98-103,121-141)."""
    out4, _, horizons = K.forecast_losses(y_hat.detach().contiguous(), y, need_grad=False, per_horizon=True)
    return tuple(out4.unbind(0)), horizons[0], horizons[1]
