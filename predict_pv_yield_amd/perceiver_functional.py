"""torch.autograd bindings of the Perceiver-path kernels (include/pv_yield_hip.h: pv_gemm_f32, pv_layernorm_*,
pv_softmax_*, pv_geglu_*, pv_mean_axis1_*).  Shapes, reshapes, residual adds and concatenations stay torch views /
data movement; every contraction and every row-wise arithmetic op runs in the HIP kernels.  No CPU path."""
import torch

from . import hip_ops as K


# ---- gradients of weights that several layers share ---------------------------------------------------------------------
# perceiver_pytorch's weight_tie_layers=True (predict_pv_yield/models/perceiver/perceiver.py:70-80) applies the same
# nn.Linear / nn.LayerNorm up to 19 times per forward; autograd then sums the 19 gradient contributions with one add
# kernel each.  Here the first contribution that arrives in a backward pass is handed to autograd as usual and KEPT; every
# later one of the same pass is added into that very tensor by the kernel that produces it (same order of additions as
# autograd's: arrival order) and autograd is told "no gradient" -- it still holds the first tensor, which now carries the sum.
ACCUMULATE_TIED_GRADS = True
_TIED = {}
_TIED_TASK = [-1]


_GRAPH_GEN = [0]      # number of backward passes seen so far: a use count noted before the latest one is stale


def _note_use(param) -> None:
    """Forward side, called by the wrappers below BEFORE Function.apply (inside Function.forward the grad mode is always
    off and ctx.needs_input_grad ignores it): counts how often `param` is applied in the graph being built.  A parameter
    applied ONCE per forward (exp-003's untied layers, every Linear of the heads) is never registered below: holding a
    second reference to its gradient would keep AccumulateGrad from taking the tensor over (one extra copy kernel per
    parameter and step) and keep a second copy of every gradient alive.  A forward under torch.no_grad() (validation, the
    sanity check) builds no graph and is not counted; counts left over by a graph that was never backpropagated are dropped
    at the first use after the next backward pass (they carry the generation they were noted in)."""
    if param is None or not param.requires_grad or not torch.is_grad_enabled():
        return
    if getattr(param, "_pv_uses_gen", -1) != _GRAPH_GEN[0]:
        param._pv_uses, param._pv_uses_gen = 0, _GRAPH_GEN[0]
    param._pv_uses += 1


def _tied_slot(param: torch.Tensor):
    """-> (key, buffer | None): the tensor this backward pass already handed out for `param`, if any.  key is None when
    nothing has to be kept (the parameter was applied once)."""
    task = torch._C._current_graph_task_id()
    if not ACCUMULATE_TIED_GRADS or task < 0:
        return None, None
    if _TIED_TASK[0] != task:               # entries live for one backward pass (as _SHARED_ACT below)
        _TIED.clear()
        _TIED_TASK[0] = task
        _GRAPH_GEN[0] += 1
    uses = getattr(param, "_pv_uses", 0)
    if uses > 0:
        param._pv_uses = uses - 1           # this call consumes one of the forward's applications
    key = (param.data_ptr(), param.numel())
    hit = _TIED.get(key)
    if hit is not None:
        if uses <= 1:                       # the last contribution: nothing more will be added to the kept tensor
            _TIED.pop(key, None)
        return key, hit
    return (key, None) if uses > 1 else (None, None)


def _tied_keep(key, grad) -> None:
    if key is not None:
        _TIED[key] = grad


# the same for an ACTIVATION that several layers consume (the projected context of weight-tied cross-attention layers,
# tagged `_pv_shared` by its producer): entries live for one backward pass only -- they are dropped as soon as another pass
# is seen, so no gradient tensor outlives its step here
_SHARED_ACT = {}
_SHARED_ACT_TASK = [-1]


def mark_shared(t: torch.Tensor) -> torch.Tensor:
    t._pv_shared = True
    return t


def _shared_activation_slot(t: torch.Tensor):
    task = torch._C._current_graph_task_id()
    if not ACCUMULATE_TIED_GRADS or task < 0 or not getattr(t, "_pv_shared", False):
        return None, None
    if _SHARED_ACT_TASK[0] != task:
        _SHARED_ACT.clear()
        _SHARED_ACT_TASK[0] = task
    key = (t.data_ptr(), t.numel())
    return key, _SHARED_ACT.get(key)


# Operand precision of every linear() issued while the flag is set: "bf16" = both operands rounded once to bf16, one
# matrix-core product, f32 accumulation -- what torch.autocast makes of nn.Linear under the reference's Lightning precision=16
# (experiments/003_...py:40,288-294); "f32" = the f32-accurate three-term products.  A layer's backward products follow the
# precision its forward ran in.
_LINEAR_BF16 = [False]


class linear_operands:
    """with linear_operands("bf16"): ... -- the Perceiver core wraps its forward in it (operand_dtype)."""

    def __init__(self, dtype: str):
        if dtype not in ("f32", "bf16"):
            raise ValueError("linear_operands: 'f32' or 'bf16'")
        self.bf16 = dtype == "bf16"

    def __enter__(self):
        self.prev = _LINEAR_BF16[0]
        _LINEAR_BF16[0] = self.bf16

    def __exit__(self, *exc):
        _LINEAR_BF16[0] = self.prev
        return False


class LinearRowsF32(torch.autograd.Function):
    """y[..., out] = x[..., in] @ W[out, in]^T (+ bias) (+ residual): nn.Linear over an arbitrary number of rows
    (rows >> in, out); `residual` (shaped like y) rides in the GEMM epilogue and receives dy unchanged."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual=None):
        x2 = x.contiguous().view(-1, x.shape[-1])
        res2 = residual.contiguous().view(-1, weight.shape[0]) if residual is not None else None
        ctx.bf16 = _LINEAR_BF16[0]
        y = K.gemm(x2, weight.t(), bias=bias, residual=res2, bf16_operands=ctx.bf16)
        ctx.save_for_backward(x2, weight, bias)
        ctx.has_bias, ctx.x_shape = bias is not None, x.shape
        return y.view(x.shape[:-1] + (weight.shape[0],))

    @staticmethod
    def backward(ctx, dy):
        x2, weight, bias = ctx.saved_tensors
        dy = dy.contiguous()
        dy2 = dy.view(-1, weight.shape[0])
        bf = getattr(ctx, "bf16", False)
        dx = K.gemm(dy2, weight, bf16_operands=bf).view(ctx.x_shape) if ctx.needs_input_grad[0] else None
        key, acc = _tied_slot(weight)
        if acc is not None:
            K.gemm_splitk(dy2.t(), x2, accumulate_into=acc, bf16_operands=bf)   # [out, rows] @ [rows, in], += into the kept tensor
            dw = None
        else:
            dw = K.gemm_splitk(dy2.t(), x2, bf16_operands=bf)
            _tied_keep(key, dw)
        db = None
        if ctx.has_bias:
            key, acc = _tied_slot(bias)
            if acc is not None:
                K.colsum(dy2, accumulate_into=acc)
            else:
                db = K.colsum(dy2)
                _tied_keep(key, db)
        return dx, dw, db, (dy if len(ctx.needs_input_grad) > 3 and ctx.needs_input_grad[3] else None)


def linear(x, weight, bias=None, residual=None):
    _note_use(weight)
    _note_use(bias)
    return LinearRowsF32.apply(x, weight, bias, residual)


class LinearRowsKV16(torch.autograd.Function):
    """The key / value projection of a cross-attention in bf16-operand mode: the product is written as bf16 (what the attention
    kernels would round K and V to anyway), half the bytes of both the HBM-bound projection store and the HBM-bound attention
    reads.  Autograd sees an f32 tensor of the projection's shape -- a zero-stride placeholder that owns no memory and holds
    NaN -- because the gradient that comes back for it IS f32 (dK / dV of the attention backward); the values travel beside it
    as `._pv_bf16` (set by linear_kv16) and only AttentionCoreF32's fused bf16 kernels take them."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = x.contiguous().view(-1, x.shape[-1])
        ctx.bf16 = _LINEAR_BF16[0]
        kv16 = K.gemm_rows_bf16out(x2, weight.t(), bias=bias, bf16_operands=ctx.bf16).view(x.shape[:-1] + (weight.shape[0],))
        ctx.save_for_backward(x2, weight, bias)
        ctx.has_bias, ctx.x_shape = bias is not None, x.shape
        ctx.mark_non_differentiable(kv16)
        ctx.set_materialize_grads(False)      # (else every backward fills a zero bf16 "gradient" of kv16's size: 85 us)
        placeholder = torch.full((), float("nan"), dtype=torch.float32, device=x.device).expand(kv16.shape)
        return placeholder, kv16

    @staticmethod
    def backward(ctx, dy, _):
        return LinearRowsF32.backward(ctx, dy)[:3]


def linear_kv16(x, weight, bias=None):
    """linear() whose values are stored as bf16 (see LinearRowsKV16); falls back to linear() where the kernel's shape rule
    (a tall input with <= 64 features) does not hold."""
    x2 = x.reshape(-1, x.shape[-1]) if x.is_contiguous() else None
    if x2 is None or not K.gemm_rows_bf16out_supported(x2, weight.t()) or weight.shape[0] % 8:
        return linear(x, weight, bias)
    _note_use(weight)
    _note_use(bias)
    y, kv16 = LinearRowsKV16.apply(x, weight, bias)
    y._pv_bf16 = kv16
    return y


class CrossAttentionKV16(torch.autograd.Function):
    """A whole cross-attention against a context that ONE layer consumes, in bf16-operand mode: to_kv (bias-free) and the
    attention as one node, so that K / V AND their gradient can stay bf16 between the kernels -- the projection writes bf16, the
    attention reads it and writes dK / dV as bf16, and to_kv's two backward products (which round that operand to bf16 anyway:
    identical bits) read 2 instead of 4 bytes per element.  (A context shared by weight-tied layers keeps the separate nodes:
    its gradient is a sum over the layers, collected in f32 -- linear_kv16 + attention_core.)"""

    @staticmethod
    def forward(ctx, q, context, w_kv, heads, scale):
        q = q.contiguous()
        c2 = context.contiguous().view(-1, context.shape[-1])
        kv16 = K.gemm_rows_bf16out(c2, w_kv.t(), bf16_operands=True).view(context.shape[:-1] + (w_kv.shape[0],))
        out, lse = K.attention_fwd(q, kv16, heads, scale, bf16_operands=True)
        ctx.save_for_backward(q, c2, w_kv, kv16, out, lse)
        ctx.heads, ctx.scale, ctx.c_shape = heads, scale, context.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        q, c2, w_kv, kv16, out, lse = ctx.saved_tensors
        dq, dkv16 = K.attention_bwd(q, kv16, out, dout.contiguous(), lse, ctx.heads, ctx.scale, bf16_operands=True, dkv_bf16=True)
        g2 = dkv16.view(-1, w_kv.shape[0])
        dc = K.gemm(g2, w_kv, bf16_operands=True).view(ctx.c_shape) if ctx.needs_input_grad[1] else None
        key, acc = _tied_slot(w_kv)
        if acc is not None:
            K.gemm_splitk(g2.t(), c2, accumulate_into=acc, bf16_operands=True)
            dw = None
        else:
            dw = K.gemm_splitk(g2.t(), c2, bf16_operands=True)
            _tied_keep(key, dw)
        return dq, dc, dw, None, None


class CrossAttentionNormKV16(torch.autograd.Function):
    """CrossAttentionKV16 with the context's LayerNorm inside the node, for a context that needs no gradient itself (the
    images): norm_context + to_kv + attention.  The backward then needs from d(K | V) only dW_kv and the LayerNorm's
    (d weight, d bias), and those two sums are formed straight from the bf16 gradient rows
    (K.layernorm_bwd_params_from_proj: the product dKV W per 32-row block on the matrix cores, folded into the sums in the
    accumulators) -- the [rows, d] gradient of the normalised context is neither written by a GEMM nor read back by a LayerNorm
    backward."""

    @staticmethod
    def forward(ctx, q, data, ln_w, ln_b, eps, w_kv, heads, scale, pos=None):
        # pos (optional): position features [P, d2] that follow the channels of every pixel (data [..., P, d1]): the context rows
        # are read from the two tensors by the one-pass kernels, never concatenated
        q = q.contiguous()
        x = data.contiguous()
        ctx.one_pass = ONE_PASS_CONTEXT_BACKWARD and w_kv.shape[0] == 128
        if pos is not None and not (ctx.one_pass and ONE_PASS_CONTEXT_FORWARD):
            raise RuntimeError("cross_attention_norm_kv16: separate position features need the one-pass context kernels")
        if ctx.one_pass and ONE_PASS_CONTEXT_FORWARD and K.context_fwd_supported(x, w_kv, pos):
            kv16, mean, rstd = K.context_fwd(x, ln_w, ln_b, w_kv, eps, x2=pos)      # the normalised context is never written
            c2 = None
        else:
            y, mean, rstd = K.layernorm_fwd(x, ln_w, ln_b, eps)
            c2 = y.view(-1, y.shape[-1])
            kv16 = K.gemm_rows_bf16out(c2, w_kv.t(), bf16_operands=True).view(x.shape[:-1] + (w_kv.shape[0],))
        out, lse = K.attention_fwd(q, kv16, heads, scale, bf16_operands=True)
        # (the one-pass backward re-forms the normalised context from x, mean, rstd: it is not kept)
        ctx.save_for_backward(q, x, ln_w, ln_b, mean, rstd, None if ctx.one_pass else c2, w_kv, kv16, out, lse, pos)
        ctx.heads, ctx.scale = heads, scale
        return out

    @staticmethod
    def backward(ctx, dout):
        q, x, ln_w, ln_b, mean, rstd, c2, w_kv, kv16, out, lse, pos = ctx.saved_tensors
        dq, dkv16 = K.attention_bwd(q, kv16, out, dout.contiguous(), lse, ctx.heads, ctx.scale, bf16_operands=True, dkv_bf16=True)
        g2 = dkv16.view(-1, w_kv.shape[0])
        key, acc = _tied_slot(w_kv)
        nkey, nacc = _tied_slot(ln_w)
        if ctx.one_pass:
            dw, dlw, dlb = K.context_bwd(g2, w_kv, x, mean, rstd, ln_w, ln_b, accumulate_kv_into=acc, accumulate_ln_into=nacc, x2=pos)
            if acc is None:
                _tied_keep(key, dw)
            else:
                dw = None
            if nacc is None:
                _tied_keep(nkey, (dlw, dlb))
            else:
                dlw = dlb = None
            return dq, None, dlw, dlb, None, dw, None, None, None
        if acc is not None:
            K.gemm_splitk(g2.t(), c2, accumulate_into=acc, bf16_operands=True)
            dw = None
        else:
            dw = K.gemm_splitk(g2.t(), c2, bf16_operands=True)
            _tied_keep(key, dw)
        if nacc is not None:
            K.layernorm_bwd_params_from_proj(g2, w_kv, x, mean, rstd, accumulate_into=nacc)
            dlw = dlb = None
        else:
            dlw, dlb = K.layernorm_bwd_params_from_proj(g2, w_kv, x, mean, rstd)
            _tied_keep(nkey, (dlw, dlb))
        return dq, None, dlw, dlb, None, dw, None, None, None


ONE_PASS_CONTEXT_BACKWARD = True      # False: weight-gradient GEMM + the LayerNorm-parameter kernel (two passes over dK | dV)
ONE_PASS_CONTEXT_FORWARD = True       # False: LayerNorm kernel + projection GEMM (the normalised context through memory)


def cross_attention_norm_kv16_supported(q, data, ln_w, w_kv, heads, pos=None) -> bool:
    """The fused node's shape rules, plus: the un-normalised context takes no gradient and the LayerNorm is narrow enough for
    the accumulator-side sums (d <= 64).  pos: position features kept apart from the channels (see the node)."""
    d = data.shape[-1] + (pos.shape[-1] if pos is not None else 0)
    if data.requires_grad or not data.is_contiguous() or data.dtype != torch.float32 or d > 64:
        return False
    if w_kv.shape[0] not in (64, 128) or not w_kv.is_contiguous() or ln_w.shape[0] != d or w_kv.shape[1] != d:
        return False
    if not (q.shape[-1] // heads == 64 and q.shape[1] <= 128 and w_kv.shape[0] == 2 * q.shape[-1]):
        return False
    if pos is not None:
        return (ONE_PASS_CONTEXT_BACKWARD and ONE_PASS_CONTEXT_FORWARD and not pos.requires_grad
                and K.context_fwd_supported(data, w_kv, pos) and data.numel() // data.shape[-1] < 2 ** 31)
    return K.gemm_rows_bf16out_supported(data.view(-1, d), w_kv.t())


def cross_attention_norm_kv16(q, data, ln_w, ln_b, eps, w_kv, heads, scale, pos=None):
    _note_use(w_kv)
    _note_use(ln_w)
    return CrossAttentionNormKV16.apply(q, data, ln_w, ln_b, eps, w_kv, heads, scale, pos)


def cross_attention_kv16_supported(q, context, w_kv, heads) -> bool:
    c2 = context.reshape(-1, context.shape[-1]) if context.is_contiguous() else None
    return (c2 is not None and K.gemm_rows_bf16out_supported(c2, w_kv.t()) and w_kv.shape[0] == 2 * q.shape[-1]
            and q.shape[-1] // heads == 64 and q.shape[1] <= 128 and w_kv.shape[0] % 8 == 0)


def cross_attention_kv16(q, context, w_kv, heads, scale):
    _note_use(w_kv)
    return CrossAttentionKV16.apply(q, context, w_kv, heads, scale)


class MatmulF32(torch.autograd.Function):
    """C = A @ B for strided batched views; dA = dC @ B^T, dB = A^T @ dC (same kernel, transposed views)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return K.gemm(a, b)

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        dc = dc.contiguous()
        da = K.gemm(dc, b.transpose(-1, -2)) if ctx.needs_input_grad[0] else None
        db = K.gemm(a.transpose(-1, -2), dc) if ctx.needs_input_grad[1] else None
        return da, db


def matmul(a, b):
    return MatmulF32.apply(a, b)


def _head_view(t: torch.Tensor, h: int, lo: int, hi: int) -> torch.Tensor:
    """Columns [lo, hi) of t [b, n, c] as a strided per-head view [b, h, n, (hi-lo)/h]: no copy, the GEMM reads / writes
    through the strides."""
    b, n, _ = t.shape
    return t[..., lo:hi].unflatten(-1, (h, (hi - lo) // h)).permute(0, 2, 1, 3)


class AttentionCoreF32(torch.autograd.Function):
    """softmax(scale * q k^T) v of Attention.forward for q [b, i, h*d] and the fused key/value projection kv [b, j, 2*h*d]
    (k = first half, v = second half of the last dimension, heads interleaved as '(h d)').  The four GEMMs of the forward
    and the five of the backward read q / k / v / dout per head through strided views and WRITE out, dq and the two halves
    of dkv in place through strided views as well, so no chunk / permute / slice-gradient copies or zero fills exist;
    the scores tensor is softmax'ed in place and kept for the backward."""

    @staticmethod
    def forward(ctx, q, kv, heads, scale, bf16_operands=False):
        kv16 = getattr(kv, "_pv_bf16", None)      # the projection stored its values as bf16 (linear_kv16): kv is a placeholder
        if kv16 is not None and not (bf16_operands and q.shape[-1] // heads == 64 and q.shape[1] <= 128):
            raise NotImplementedError("a bf16-stored kv is for the fused bf16-operand kernels (head_dim 64, <= 128 queries)")
        q = q.contiguous()
        if kv16 is not None:
            if getattr(kv, "_pv_shared", False):
                kv16._pv_shared = True      # (the gradient of a context shared by weight-tied layers collects in one tensor)
            kv = kv16
        else:
            kv = kv.contiguous()
        inner = q.shape[-1]
        ctx.heads, ctx.scale = heads, scale
        ctx.fused = inner // heads == 64 and q.shape[1] <= 128
        ctx.bf16 = bool(bf16_operands)
        if ctx.bf16 and not ctx.fused:
            raise NotImplementedError("bf16 attention operands are built for the fused kernels (head_dim 64, <= 128 queries)")
        if ctx.fused:
            # online-softmax kernels: the [queries x keys] scores / probabilities never reach memory; the backward
            # recomputes them tile by tile from the saved log-sum-exp
            out, lse = K.attention_fwd(q, kv, heads, scale, bf16_operands=ctx.bf16)
            ctx.save_for_backward(q, kv, out, lse)
            return out
        qh, kh, vh = _head_view(q, heads, 0, inner), _head_view(kv, heads, 0, inner), _head_view(kv, heads, inner, 2 * inner)
        p = K.softmax_fwd_(K.gemm(qh, kh.transpose(-1, -2)), scale)                 # [b, h, i, j]
        out = torch.empty_like(q)
        K.gemm(p, vh, out=_head_view(out, heads, 0, inner))
        ctx.save_for_backward(q, kv, p)
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.fused:
            q, kv, out, lse = ctx.saved_tensors
            # the projected context of weight-tied layers is ONE tensor used by every layer: its gradient collects in the
            # first arrival's tensor (see _tied_slot), added by the backward kernel's own dK / dV store
            key, acc = _shared_activation_slot(kv) if ctx.bf16 else (None, None)
            if acc is not None:
                dq, _ = K.attention_bwd(q, kv, out, dout.contiguous(), lse, ctx.heads, ctx.scale, bf16_operands=True,
                                        accumulate_dkv_into=acc)
                return dq, None, None, None, None
            dq, dkv = K.attention_bwd(q, kv, out, dout.contiguous(), lse, ctx.heads, ctx.scale, bf16_operands=ctx.bf16)
            if key is not None:
                _SHARED_ACT[key] = dkv
            return dq, dkv, None, None, None
        q, kv, p = ctx.saved_tensors
        h, inner = ctx.heads, q.shape[-1]
        dout = dout.contiguous()
        qh, kh, vh = _head_view(q, h, 0, inner), _head_view(kv, h, 0, inner), _head_view(kv, h, inner, 2 * inner)
        doh = _head_view(dout, h, 0, inner)
        dkv = torch.empty_like(kv)
        K.gemm(p.transpose(-1, -2), doh, out=_head_view(dkv, h, inner, 2 * inner))          # dv = p^T dout
        ds = K.softmax_bwd_(p, K.gemm(doh, vh.transpose(-1, -2)), ctx.scale)                # dp -> ds, in place
        dq = torch.empty_like(q)
        K.gemm(ds, kh, out=_head_view(dq, h, 0, inner))                                     # dq = ds k
        K.gemm(ds.transpose(-1, -2), qh, out=_head_view(dkv, h, 0, inner))                  # dk = ds^T q
        return dq, dkv, None, None, None


def attention_core(q, kv, heads, scale, bf16_operands=False):
    return AttentionCoreF32.apply(q, kv, heads, scale, bf16_operands)


class SoftmaxScaledF32(torch.autograd.Function):
    """softmax(scale * x, dim=-1), computed IN PLACE on x (the scores are a fresh GEMM output nobody else reads);
    the backward overwrites the incoming gradient the same way."""

    @staticmethod
    def forward(ctx, x, scale):
        p = K.softmax_fwd_(x, scale)
        ctx.mark_dirty(x)
        ctx.save_for_backward(p)
        ctx.scale = scale
        return p

    @staticmethod
    def backward(ctx, dp):
        (p,) = ctx.saved_tensors
        dp = dp.contiguous()
        if dp.data_ptr() == p.data_ptr():
            dp = dp.clone()
        return K.softmax_bwd_(p, dp, ctx.scale), None


def softmax_scaled_(x, scale):
    return SoftmaxScaledF32.apply(x, scale)


class LayerNormF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x = x.contiguous()
        y, mean, rstd = K.layernorm_fwd(x, weight, bias, eps)
        ctx.save_for_backward(x, weight, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, mean, rstd = ctx.saved_tensors
        key, acc = _tied_slot(weight)      # (dw | db) live in one [2 d] tensor, keyed by the weight
        if acc is not None:
            dx, _, _ = K.layernorm_bwd(x, weight, dy.contiguous(), mean, rstd, need_dx=ctx.needs_input_grad[0],
                                       accumulate_into=acc)
            return dx, None, None, None
        dx, dw, db = K.layernorm_bwd(x, weight, dy.contiguous(), mean, rstd, need_dx=ctx.needs_input_grad[0])
        _tied_keep(key, (dw, db))
        return dx, dw, db, None


def layer_norm(x, weight, bias, eps=1e-5):
    _note_use(weight)
    return LayerNormF32.apply(x, weight, bias, eps)


class LayerNormForkF32(torch.autograd.Function):
    """(norm(x), x) of a PreNorm residual block `fn(norm(x)) + x`: the second output is x itself, routed through this node so
    that BOTH gradients of x -- through the normalised branch and past it -- arrive here and leave as one tensor: the
    LayerNorm backward kernel adds the by-pass gradient while it stores dx (autograd otherwise adds the two with one
    elementwise launch per block and step)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x = x.contiguous()
        y, mean, rstd = K.layernorm_fwd(x, weight, bias, eps)
        ctx.save_for_backward(x, weight, mean, rstd)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dpass):
        x, weight, mean, rstd = ctx.saved_tensors
        if dy is None:                       # only the by-pass was used
            return dpass, None, None, None
        add = dpass.contiguous() if dpass is not None and ctx.needs_input_grad[0] else None
        key, acc = _tied_slot(weight)
        if acc is not None:
            dx, _, _ = K.layernorm_bwd(x, weight, dy.contiguous(), mean, rstd, need_dx=ctx.needs_input_grad[0],
                                       accumulate_into=acc, dx_add=add)
            return dx, None, None, None
        dx, dw, db = K.layernorm_bwd(x, weight, dy.contiguous(), mean, rstd, need_dx=ctx.needs_input_grad[0], dx_add=add)
        _tied_keep(key, (dw, db))
        return dx, dw, db, None


def layer_norm_fork(x, weight, bias, eps=1e-5):
    """-> (layer_norm(x), x): see LayerNormForkF32."""
    _note_use(weight)
    return LayerNormForkF32.apply(x, weight, bias, eps)


class GEGLUF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return K.geglu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return K.geglu_bwd(x, dy.contiguous())


def geglu(x):
    return GEGLUF32.apply(x)


class MeanAxis1F32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.n = x.shape[1]
        return K.mean_axis1_fwd(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return K.mean_axis1_bwd(dy.contiguous(), ctx.n)


def mean_axis1(x):
    return MeanAxis1F32.apply(x)


class GRUSequenceF32(torch.autograd.Function):
    """Sequential part of one nn.GRU layer: (gi [B,T,3H], h0 [B,H] | None, w_hh, b_hh) -> (out [B,T,H], h_T [B,H])."""

    @staticmethod
    def forward(ctx, gi, h0, w_hh, b_hh):
        gi = gi.contiguous()
        h0c = h0.contiguous() if h0 is not None else None
        out, saved = K.gru_seq_fwd(gi, h0c, w_hh.contiguous(), b_hh.contiguous())
        ctx.save_for_backward(h0c, out, saved, w_hh)
        ctx.has_h0 = h0 is not None
        return out, out[:, -1].clone()

    @staticmethod
    def backward(ctx, dout, dh_last):
        h0, out, saved, w_hh = ctx.saved_tensors
        dgi, dh0, dw, db = K.gru_seq_bwd(dout.contiguous() if dout is not None else None,
                                         dh_last.contiguous() if dh_last is not None else None, h0, out, saved,
                                         w_hh.contiguous(), need_dh0=ctx.has_h0 and ctx.needs_input_grad[1])
        return dgi, dh0, dw, db


def gru(x, params, h0=None):
    """nn.GRU(batch_first=True) forward: x [B,T,in]; params = the module's nn.GRU (weight_ih_l{k}, weight_hh_l{k}, bias_*);
    h0 [layers,B,H] or None.  Returns (output of the last layer [B,T,H], h_n [layers,B,H])."""
    h_n = []
    inp = x
    for layer in range(params.num_layers):
        w_ih, w_hh = getattr(params, f"weight_ih_l{layer}"), getattr(params, f"weight_hh_l{layer}")
        b_ih, b_hh = getattr(params, f"bias_ih_l{layer}"), getattr(params, f"bias_hh_l{layer}")
        if inp.shape[-1] >= 2048:
            # very wide input (the flattened NWP image rides along): few rows x huge K -> the split-K linear kernels
            from .functional import linear_f32
            gi = linear_f32(inp.reshape(-1, inp.shape[-1]), w_ih, b_ih, relu=False).view(inp.shape[0], inp.shape[1], -1)
        else:
            gi = linear(inp, w_ih, b_ih)                   # every time step's input projection in one GEMM
        inp, h_last = GRUSequenceF32.apply(gi, h0[layer] if h0 is not None else None, w_hh, b_hh)
        h_n.append(h_last)
    return inp, torch.stack(h_n)
