"""torch.autograd bindings of the Perceiver-path kernels (include/pv_yield_hip.h: pv_gemm_f32, pv_layernorm_*,
pv_softmax_*, pv_geglu_*, pv_mean_axis1_*).  Shapes, reshapes, residual adds and concatenations stay torch views /
data movement; every contraction and every row-wise arithmetic op runs in the HIP kernels.  No CPU path."""
import torch

from . import hip_ops as K


class LinearRowsF32(torch.autograd.Function):
    """y[..., out] = x[..., in] @ W[out, in]^T (+ bias): nn.Linear over an arbitrary number of rows (rows >> in, out)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = x.contiguous().view(-1, x.shape[-1])
        y = K.gemm(x2, weight.t(), bias=bias)
        ctx.save_for_backward(x2, weight)
        ctx.has_bias, ctx.x_shape = bias is not None, x.shape
        return y.view(x.shape[:-1] + (weight.shape[0],))

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        dy2 = dy.contiguous().view(-1, weight.shape[0])
        dx = K.gemm(dy2, weight).view(ctx.x_shape) if ctx.needs_input_grad[0] else None
        dw = K.gemm_splitk(dy2.t(), x2)                                   # [out, rows] @ [rows, in]
        db = None
        if ctx.has_bias:
            ones = torch.ones((1, dy2.shape[0]), dtype=torch.float32, device=dy2.device)
            db = K.gemm_splitk(ones, dy2).view(-1)
        return dx, dw, db


def linear(x, weight, bias=None):
    return LinearRowsF32.apply(x, weight, bias)


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
        dx, dw, db = K.layernorm_bwd(x, weight, dy.contiguous(), mean, rstd, need_dx=ctx.needs_input_grad[0])
        return dx, dw, db, None


def layer_norm(x, weight, bias, eps=1e-5):
    return LayerNormF32.apply(x, weight, bias, eps)


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
