"""Perceiver — MI355X implementation of the `perceiver_pytorch.Perceiver` module the reference instantiates
(predict_pv_yield/models/perceiver/perceiver.py:70-80, perceiver_nwp_sat.py:69-79, perceiver_conv3d_nwp_sat.py:97-107,
experiments/003_...py:105-114; third-party, unpinned: requirements.txt:12).

Same constructor keywords, same module tree and state_dict keys as the package's 0.7 series (layers.{i}.{0,1,2}.fn/.norm/
.norm_context, to_logits.{1,2}; with weight_tie_layers layer 0 owns its blocks and layers >= 1 share one set).  The
modules are parameter holders: every contraction (projections, q.k^T, softmax.v, feed-forward) runs on the f32 matrix
cores through pv_gemm_f32, LayerNorm / softmax / GEGLU / mean in their own HIP kernels (perceiver_functional.py).

Two things the package recomputes in every layer are computed once here because they are identical by construction:
the Fourier position features (constant per image size) and — under weight tying — the LayerNorm'd context and its
key/value projection, which only depend on the shared parameters: one kv tensor for layer 0 and one for all tied layers.
"""
from math import pi

import torch
from torch import nn

from ... import perceiver_functional as PF


def fourier_position_features(axis, max_freq: float, num_bands: int) -> torch.Tensor:
    """[*axis, len(axis) * (2 * num_bands + 1)]: for every axis coordinate p in linspace(-1, 1): sin(p s pi), cos(p s pi) for
    s in linspace(1, max_freq / 2, num_bands), then p itself.  Input independent: built once per image size."""
    axis_pos = [torch.linspace(-1.0, 1.0, steps=size) for size in axis]
    pos = torch.stack(torch.meshgrid(*axis_pos, indexing="ij"), dim=-1).unsqueeze(-1)
    scales = torch.linspace(1.0, max_freq / 2, num_bands)
    x = pos * scales * pi
    enc = torch.cat([x.sin(), x.cos(), pos], dim=-1)
    return enc.reshape(*axis, -1)


class PreNorm(nn.Module):
    def __init__(self, dim, fn, context_dim=None):
        super().__init__()
        self.fn = fn
        self.norm = nn.LayerNorm(dim)
        self.norm_context = nn.LayerNorm(context_dim) if context_dim is not None else None


class GEGLU(nn.Module):
    pass


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, dim * mult * 2), GEGLU(), nn.Linear(dim * mult, dim))


class Attention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64):
        super().__init__()
        inner_dim = dim_head * heads
        context_dim = query_dim if context_dim is None else context_dim
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(context_dim, inner_dim * 2, bias=False)
        self.to_out = nn.Linear(inner_dim, query_dim)


FUSE_RESIDUALS = True   # `fn(norm(x)) + x`: by-pass added in fn's last GEMM epilogue / the LayerNorm backward store (False: torch adds)


KV_STORED_AS_BF16 = True      # tools/ab-style switch: False = f32 K / V in memory, rounded inside the attention kernels
LINEARS_FOLLOW_OPERAND_DTYPE = True      # False: the Linears keep f32-accurate products in "bf16" mode (rounds 2-3)
NORM_CONTEXT_IN_THE_ATTENTION_NODE = True      # False: norm_context as its own node (d(normalised context) through memory)
SPLIT_CONTEXT = True      # False: always build the [b, positions, channels + fourier] tensor


def _query_probe(x: torch.Tensor, block) -> torch.Tensor:
    """The shape the block's queries will have (a meta tensor: the fused node's shape rules look at nothing else of q)."""
    return torch.empty((x.shape[0], x.shape[1], block.fn.to_q.weight.shape[0]), device="meta")


def _attend(attn: Attention, xn: torch.Tensor, kv: torch.Tensor, bf16_operands: bool = False, residual=None) -> torch.Tensor:
    """Attention.forward given the normalised query input and the (already projected) keys/values [b, j, 2*inner];
    `residual` (the block's `+ x`) is added in the epilogue of to_out."""
    q = PF.linear(xn, attn.to_q.weight)                                  # [b, i, inner]
    out = PF.attention_core(q, kv, attn.heads, attn.scale, bf16_operands=bf16_operands)   # softmax(scale q k^T) v, per head
    return PF.linear(out, attn.to_out.weight, attn.to_out.bias, residual=residual)


def _attend_raw_context(block: "PreNorm", xn: torch.Tensor, data, residual=None) -> torch.Tensor:
    """_attend_context given the context BEFORE its LayerNorm (a context that takes no gradient: the images): norm_context,
    to_kv and the attention as one node, whose backward forms the LayerNorm's parameter gradients straight from d(K | V)."""
    attn = block.fn
    q = PF.linear(xn, attn.to_q.weight)
    data, pos = data if isinstance(data, tuple) else (data, None)      # (channels, position features) kept apart, or one tensor
    out = PF.cross_attention_norm_kv16(q, data, block.norm_context.weight, block.norm_context.bias, block.norm_context.eps,
                                       attn.to_kv.weight, attn.heads, attn.scale, pos)
    return PF.linear(out, attn.to_out.weight, attn.to_out.bias, residual=residual)


def _attend_context(attn: Attention, xn: torch.Tensor, context: torch.Tensor, residual=None) -> torch.Tensor:
    """Attention.forward in bf16-operand mode given the normalised query input and the NORMALISED, not yet projected context of
    a cross-attention that is this context's only consumer: to_kv + attention as one node (falls back to the separate nodes
    where the kernels' shape rules do not hold)."""
    q = PF.linear(xn, attn.to_q.weight)
    if PF.cross_attention_kv16_supported(q, context, attn.to_kv.weight, attn.heads):
        out = PF.cross_attention_kv16(q, context, attn.to_kv.weight, attn.heads, attn.scale)
    else:
        out = PF.attention_core(q, PF.linear_kv16(context, attn.to_kv.weight), attn.heads, attn.scale, bf16_operands=True)
    return PF.linear(out, attn.to_out.weight, attn.to_out.bias, residual=residual)


def _feed_forward(block: PreNorm, x: torch.Tensor) -> torch.Tensor:
    """`block(x) + x` of a PreNorm(FeedForward): the by-pass rides in the last Linear's epilogue forward and in the LayerNorm
    kernel backward."""
    ff = block.fn
    if not FUSE_RESIDUALS:
        y = PF.layer_norm(x, block.norm.weight, block.norm.bias, block.norm.eps)
        y = PF.geglu(PF.linear(y, ff.net[0].weight, ff.net[0].bias))
        return PF.linear(y, ff.net[2].weight, ff.net[2].bias) + x
    y, x_pass = PF.layer_norm_fork(x, block.norm.weight, block.norm.bias, block.norm.eps)
    y = PF.linear(y, ff.net[0].weight, ff.net[0].bias)
    y = PF.geglu(y)
    return PF.linear(y, ff.net[2].weight, ff.net[2].bias, residual=x_pass)


class Perceiver(nn.Module):
    def __init__(self, *, num_freq_bands, depth, max_freq, freq_base=2, input_channels=3, input_axis=2, num_latents=512,
                 latent_dim=512, cross_heads=1, latent_heads=8, cross_dim_head=64, latent_dim_head=64, num_classes=1000,
                 attn_dropout=0.0, ff_dropout=0.0, weight_tie_layers=False, fourier_encode_data=True, self_per_cross_attn=1,
                 operand_dtype: str = "f32"):
        super().__init__()
        # operand_dtype (new, optional): "f32" = exact-f32 matrix-core products everywhere; "bf16" = the attention products
        # (q k^T, p v and their backward) AND the Linear layers take bf16 operands with f32 accumulation, softmax / LayerNorm
        # stay f32 -- what Lightning's
        # precision=16 does to them in experiments/003 (:40, :290)
        if operand_dtype not in ("f32", "bf16"):
            raise ValueError("operand_dtype must be 'f32' or 'bf16'")
        self.operand_dtype = operand_dtype
        if attn_dropout or ff_dropout:
            raise NotImplementedError("dropout is not built (the reference never sets it)")
        if not fourier_encode_data:
            raise NotImplementedError("fourier_encode_data=False is not built (the reference never sets it)")
        self.input_axis, self.max_freq, self.num_freq_bands = input_axis, max_freq, num_freq_bands
        input_dim = input_axis * ((num_freq_bands * 2) + 1) + input_channels
        self.latents = nn.Parameter(torch.randn(num_latents, latent_dim))
        make = dict(
            cross_attn=lambda: PreNorm(latent_dim, Attention(latent_dim, input_dim, heads=cross_heads, dim_head=cross_dim_head),
                                       context_dim=input_dim),
            cross_ff=lambda: PreNorm(latent_dim, FeedForward(latent_dim)),
            latent_attn=lambda: PreNorm(latent_dim, Attention(latent_dim, heads=latent_heads, dim_head=latent_dim_head)),
            latent_ff=lambda: PreNorm(latent_dim, FeedForward(latent_dim)))
        cache = {}

        def get(name, cached):
            if not cached:
                return make[name]()
            if name not in cache:
                cache[name] = make[name]()
            return cache[name]

        self.layers = nn.ModuleList([])
        for i in range(depth):
            c = i > 0 and weight_tie_layers
            self_attns = nn.ModuleList([nn.ModuleList([get("latent_attn", c), get("latent_ff", c)])
                                        for _ in range(self_per_cross_attn)])
            self.layers.append(nn.ModuleList([get("cross_attn", c), get("cross_ff", c), self_attns]))
        # Reduce('b n d -> b d', 'mean') has no parameters; Identity keeps the package's indices (to_logits.1 / .2)
        self.to_logits = nn.Sequential(nn.Identity(), nn.LayerNorm(latent_dim), nn.Linear(latent_dim, num_classes))
        self._pos_cache = {}

    def _position_features(self, axis, device):
        key = (tuple(axis), str(device))
        if key not in self._pos_cache:
            self._pos_cache[key] = fourier_position_features(axis, self.max_freq, self.num_freq_bands).to(device)
        return self._pos_cache[key]

    def forward(self, data, mask=None):
        if mask is not None:
            raise NotImplementedError("attention masks are not built (the reference never passes one)")
        if not data.is_cuda:
            raise RuntimeError("predict_pv_yield_amd Perceiver runs on the MI355X only: move the module and the batch to "
                               "cuda (there is no CPU fallback)")
        # operand_dtype="bf16": every Linear of the Perceiver takes bf16 operands too (one product, f32 accumulation), as
        # torch.autocast does under the reference's precision=16; LayerNorm, softmax, GEGLU and the residual stream stay f32
        with PF.linear_operands(self.operand_dtype if LINEARS_FOLLOW_OPERAND_DTYPE else "f32"):
            return self._forward(data)

    def _forward(self, data):
        b, *axis, _ = data.shape
        assert len(axis) == self.input_axis, "input data must have the right number of axis"
        enc = self._position_features(axis, data.device)
        x = self.latents.unsqueeze(0).expand(b, -1, -1)
        bf = self.operand_dtype == "bf16"
        # When EVERY cross-attention block takes its context through the one-pass kernels (one consumer each, bf16-operand mode, no
        # gradient into the images), those read a pixel's channels and its position features from where they are: the
        # [b, positions, channels + fourier] tensor of Perceiver.forward is not built (b x positions x (c + f) floats written
        # once and read by every block, forward and backward)
        split = None
        if SPLIT_CONTEXT and bf and KV_STORED_AS_BF16 and LINEARS_FOLLOW_OPERAND_DTYPE and NORM_CONTEXT_IN_THE_ATTENTION_NODE:
            chans = data.float().reshape(b, -1, data.shape[-1]).contiguous()
            pos = enc.reshape(-1, enc.shape[-1]).contiguous()
            blocks = {id(l[0]): l[0] for l in self.layers}
            counts = {}
            for l in self.layers:
                counts[id(l[0])] = counts.get(id(l[0]), 0) + 1
            if all(counts[k] == 1 and PF.cross_attention_norm_kv16_supported(_query_probe(x, blk), chans, blk.norm_context.weight,
                                                                              blk.fn.to_kv.weight, blk.fn.heads, pos)
                   for k, blk in blocks.items()):
                split = (chans, pos)
        if split is None:
            data = torch.cat((data.float(), enc.unsqueeze(0).expand(b, *enc.shape)), dim=-1)
            data = data.reshape(b, -1, data.shape[-1])                     # [b, positions, channels + fourier]
        kv_of = {}                                                         # cross-attention block -> projected context
        ctx_of = {}                                                        # ... -> normalised context (one-consumer blocks)
        raw_of = {}                                                        # ... -> the context before its LayerNorm (one consumer, no gradient)
        uses = {}
        for cross_attn, _, _ in self.layers:
            uses[id(cross_attn)] = uses.get(id(cross_attn), 0) + 1
        for cross_attn, cross_ff, self_attns in self.layers:
            key = id(cross_attn)
            one_use = bf and KV_STORED_AS_BF16 and LINEARS_FOLLOW_OPERAND_DTYPE and uses[key] == 1
            if split is not None:
                raw_of[key] = split
            elif (one_use and NORM_CONTEXT_IN_THE_ATTENTION_NODE and key not in raw_of and
                    PF.cross_attention_norm_kv16_supported(_query_probe(x, cross_attn), data, cross_attn.norm_context.weight,
                                                           cross_attn.fn.to_kv.weight, cross_attn.fn.heads)):
                raw_of[key] = data
            if key not in kv_of and key not in ctx_of and key not in raw_of:
                ctx = PF.layer_norm(data, cross_attn.norm_context.weight, cross_attn.norm_context.bias,
                                    cross_attn.norm_context.eps)
                if one_use:
                    # a context ONE layer consumes: to_kv and the attention run as one node (PF.cross_attention_kv16), K / V
                    # and their gradient bf16 between the kernels
                    ctx_of[key] = ctx
                else:
                    # consumed by every tied layer.  bf16-operand mode: K / V are STORED as bf16 (the values the attention
                    # kernels round to anyway): the projection's store and the attention's reads are both bound by these bytes
                    project = PF.linear_kv16 if (bf and KV_STORED_AS_BF16) else PF.linear
                    kv_of[key] = PF.mark_shared(project(ctx, cross_attn.fn.to_kv.weight))
            # every `fn(norm(x)) + x`: the by-pass is added in the epilogue of fn's last Linear (forward) and in the LayerNorm
            # backward kernel's store (backward) -- no elementwise launches
            if FUSE_RESIDUALS:
                xn, x_pass = PF.layer_norm_fork(x, cross_attn.norm.weight, cross_attn.norm.bias, cross_attn.norm.eps)
                x = (_attend_raw_context(cross_attn, xn, raw_of[key], residual=x_pass) if key in raw_of else
                     _attend_context(cross_attn.fn, xn, ctx_of[key], residual=x_pass) if key in ctx_of else
                     _attend(cross_attn.fn, xn, kv_of[key], bf, residual=x_pass))
            else:
                xn = PF.layer_norm(x, cross_attn.norm.weight, cross_attn.norm.bias, cross_attn.norm.eps)
                x = (_attend_raw_context(cross_attn, xn, raw_of[key]) if key in raw_of else
                     _attend_context(cross_attn.fn, xn, ctx_of[key]) if key in ctx_of else
                     _attend(cross_attn.fn, xn, kv_of[key], bf)) + x
            x = _feed_forward(cross_ff, x)
            for self_attn, self_ff in self_attns:
                if FUSE_RESIDUALS:
                    xn, x_pass = PF.layer_norm_fork(x, self_attn.norm.weight, self_attn.norm.bias, self_attn.norm.eps)
                    kv = PF.linear(xn, self_attn.fn.to_kv.weight)
                    x = _attend(self_attn.fn, xn, kv, bf, residual=x_pass)
                else:
                    xn = PF.layer_norm(x, self_attn.norm.weight, self_attn.norm.bias, self_attn.norm.eps)
                    kv = PF.linear(xn, self_attn.fn.to_kv.weight)
                    x = _attend(self_attn.fn, xn, kv, bf) + x
                x = _feed_forward(self_ff, x)
        x = PF.mean_axis1(x)
        x = PF.layer_norm(x, self.to_logits[1].weight, self.to_logits[1].bias, self.to_logits[1].eps)
        return PF.linear(x, self.to_logits[2].weight, self.to_logits[2].bias)
