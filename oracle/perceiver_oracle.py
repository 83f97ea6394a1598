"""CPU restatement of perceiver_pytorch.Perceiver (lucidrains) as the reference instantiates it
(predict_pv_yield/models/perceiver/perceiver.py:70-80, perceiver_nwp_sat.py:69-79, perceiver_conv3d_nwp_sat.py:97-107,
experiments/003_...py:105-114).

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).  Never imported by the product package.

PARITY UNPINNED: the package is a third-party dependency that is neither installed here nor pinned by the reference
(requirements.txt:12 `perceiver_pytorch`, no version) and the reference's tests pin output shapes only
(tests/models/perceiver/test_perceiver.py:34-37).  This file restates the published architecture of the 0.7-series
(Aug-Nov 2021, the period of the reference's experiments) in plain torch:

  fourier_encode(x, max_freq, num_bands): scales = linspace(1, max_freq / 2, num_bands); x*scales*pi -> cat(sin, cos, x)
  Perceiver.forward: pos = meshgrid(linspace(-1, 1, size) per axis) -> fourier features, concatenated to the data;
      x = latents; per layer: x += cross_attn(x, context); x += cross_ff(x); x += self_attn(x); x += self_ff(x)
      to_logits = mean over latents -> LayerNorm -> Linear
  weight_tie_layers: layer 0 owns its blocks, layers >= 1 share one set (cache_fn with _cache = i > 0)
  PreNorm(LayerNorm on x, LayerNorm on the context for cross-attention), Attention(to_q, to_kv without bias, to_out
  with bias, scale = dim_head**-0.5, softmax over the context), FeedForward(Linear(d, 8d) -> GEGLU -> Linear(4d, d)).
The module tree (and therefore the state_dict keys) follows the package: layers.{i}.{0:cross_attn,1:cross_ff,2:self_attns}
.fn / .norm / .norm_context, to_logits.{1,2}.
"""
from math import pi

import torch
import torch.nn.functional as F
from torch import nn


def fourier_encode(x: torch.Tensor, max_freq: float, num_bands: int = 4) -> torch.Tensor:
    x = x.unsqueeze(-1)
    orig_x = x
    scales = torch.linspace(1.0, max_freq / 2, num_bands, device=x.device, dtype=x.dtype)
    scales = scales[(*((None,) * (len(x.shape) - 1)), Ellipsis)]
    x = x * scales * pi
    x = torch.cat([x.sin(), x.cos()], dim=-1)
    return torch.cat((x, orig_x), dim=-1)


def position_encoding(axis, max_freq: float, num_bands: int) -> torch.Tensor:
    """[*axis, len(axis) * (2 * num_bands + 1)] Fourier features of the pixel grid."""
    axis_pos = [torch.linspace(-1.0, 1.0, steps=size) for size in axis]
    pos = torch.stack(torch.meshgrid(*axis_pos, indexing="ij"), dim=-1)
    enc = fourier_encode(pos, max_freq, num_bands)
    return enc.reshape(*axis, -1)


class PreNorm(nn.Module):
    def __init__(self, dim, fn, context_dim=None):
        super().__init__()
        self.fn = fn
        self.norm = nn.LayerNorm(dim)
        self.norm_context = nn.LayerNorm(context_dim) if context_dim is not None else None

    def forward(self, x, **kwargs):
        x = self.norm(x)
        if self.norm_context is not None:
            kwargs.update(context=self.norm_context(kwargs["context"]))
        return self.fn(x, **kwargs)


class GEGLU(nn.Module):
    def forward(self, x):
        x, gates = x.chunk(2, dim=-1)
        return x * F.gelu(gates)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, dim * mult * 2), GEGLU(), nn.Linear(dim * mult, dim))

    def forward(self, x):
        return self.net(x)


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

    def forward(self, x, context=None):
        h = self.heads
        q = self.to_q(x)
        context = x if context is None else context
        k, v = self.to_kv(context).chunk(2, dim=-1)
        b, n, _ = q.shape
        split = lambda t: t.reshape(b, t.shape[1], h, -1).permute(0, 2, 1, 3).reshape(b * h, t.shape[1], -1)
        q, k, v = map(split, (q, k, v))
        sim = torch.einsum("b i d, b j d -> b i j", q, k) * self.scale
        attn = sim.softmax(dim=-1)
        out = torch.einsum("b i j, b j d -> b i d", attn, v)
        out = out.reshape(b, h, n, -1).permute(0, 2, 1, 3).reshape(b, n, -1)
        return self.to_out(out)


class OraclePerceiver(nn.Module):
    def __init__(self, *, num_freq_bands, depth, max_freq, input_channels=3, input_axis=2, num_latents=512, latent_dim=512,
                 cross_heads=1, latent_heads=8, cross_dim_head=64, latent_dim_head=64, num_classes=1000,
                 weight_tie_layers=False, self_per_cross_attn=1):
        super().__init__()
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
        self.to_logits = nn.Sequential(nn.Identity(), nn.LayerNorm(latent_dim), nn.Linear(latent_dim, num_classes))

    def forward(self, data):
        b, *axis, _ = data.shape
        assert len(axis) == self.input_axis
        enc = position_encoding(axis, self.max_freq, self.num_freq_bands).to(data)
        data = torch.cat((data, enc.unsqueeze(0).expand(b, *enc.shape)), dim=-1)
        data = data.reshape(b, -1, data.shape[-1])
        x = self.latents.unsqueeze(0).expand(b, -1, -1)
        for cross_attn, cross_ff, self_attns in self.layers:
            x = cross_attn(x, context=data) + x
            x = cross_ff(x) + x
            for self_attn, self_ff in self_attns:
                x = self_attn(x) + x
                x = self_ff(x) + x
        x = x.mean(dim=1)
        return self.to_logits(x)
