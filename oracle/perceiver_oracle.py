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
        self.net = nn.Sequential(_Linear(dim, dim * mult * 2), GEGLU(), _Linear(dim * mult, dim))

    def forward(self, x):
        return self.net(x)


class _RoundBF16(torch.autograd.Function):
    """bf16 round-trip in forward and backward: the operand rounding of a 16-bit matrix product (precision=16)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


EMULATE_BF16_ATTENTION = False   # tests flip this to compare the bf16-operand HIP kernels at tighter tolerance
EMULATE_BF16_LINEAR = False      # the same for the Perceiver's Linear layers (operand_dtype="bf16" rounds their operands too)


class _LinearBF16Ops(torch.autograd.Function):
    """nn.Linear whose three matrix products take bf16-rounded operands (f32 arithmetic on the rounded values = f32
    accumulation): y = R(x) R(w)^T + b, dx = R(dy) R(w), dw = R(dy)^T R(x), db = sum dy."""

    @staticmethod
    def forward(ctx, x, w, b):
        r = lambda t: t.to(torch.bfloat16).to(torch.float32)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return F.linear(r(x), r(w), b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        r = lambda t: t.to(torch.bfloat16).to(torch.float32)
        g = r(dy)
        dx = g @ r(w)
        dw = g.reshape(-1, g.shape[-1]).t() @ r(x).reshape(-1, x.shape[-1])
        db = dy.reshape(-1, dy.shape[-1]).sum(0) if ctx.has_bias else None
        return dx, dw, db


class _Linear(nn.Linear):
    """nn.Linear (same parameters, same state-dict keys) that can emulate 16-bit operands."""

    def forward(self, x):
        if EMULATE_BF16_LINEAR:
            return _LinearBF16Ops.apply(x, self.weight, self.bias)
        return super().forward(x)


class Attention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64):
        super().__init__()
        inner_dim = dim_head * heads
        context_dim = query_dim if context_dim is None else context_dim
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.to_q = _Linear(query_dim, inner_dim, bias=False)
        self.to_kv = _Linear(context_dim, inner_dim * 2, bias=False)
        self.to_out = _Linear(inner_dim, query_dim)

    def forward(self, x, context=None):
        h = self.heads
        q = self.to_q(x)
        context = x if context is None else context
        k, v = self.to_kv(context).chunk(2, dim=-1)
        b, n, _ = q.shape
        split = lambda t: t.reshape(b, t.shape[1], h, -1).permute(0, 2, 1, 3).reshape(b * h, t.shape[1], -1)
        q, k, v = map(split, (q, k, v))
        if EMULATE_BF16_ATTENTION:
            q, k, v = _RoundBF16.apply(q), _RoundBF16.apply(k), _RoundBF16.apply(v)
        sim = torch.einsum("b i d, b j d -> b i j", q, k) * self.scale
        attn = sim.softmax(dim=-1)
        if EMULATE_BF16_ATTENTION:
            attn = _RoundBF16.apply(attn)
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
        self.to_logits = nn.Sequential(nn.Identity(), nn.LayerNorm(latent_dim), _Linear(latent_dim, num_classes))

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


# ---- PerceiverModel: predict_pv_yield/models/perceiver/perceiver.py:42-200 ---------------------------------------------
N_SAT_CHANNELS = 11            # len(SAT_VARIABLE_NAMES[1:]), perceiver.py:24
PERCEIVER_OUTPUT_SIZE = 512
FC_OUTPUT_SIZE = 8
RNN_HIDDEN_SIZE = 16


class OraclePerceiverModel(nn.Module):
    """Same layer graph and attribute (state_dict) names as the reference PerceiverModel; forward takes plain tensors.
    nwp image size is the 64 x 64 the reference hard-codes in NWP_SIZE (perceiver.py:36)."""

    def __init__(self, history_minutes=30, forecast_minutes=120, n_nwp_channels=10, batch_size=32, num_latents=128,
                 latent_dim=64, embedding_dem=16, output_variable="pv_yield", nwp_size=None):
        super().__init__()
        from .conv3d_oracle import timestep_arithmetic
        for k, v in timestep_arithmetic(history_minutes, forecast_minutes, output_variable).items():
            setattr(self, k, v)
        self.batch_size, self.embedding_dem, self.output_variable = batch_size, embedding_dem, output_variable
        self.total_seq_length = history_minutes // 5 + forecast_minutes // 5 + 1
        nwp_size = n_nwp_channels * 64 * 64 if nwp_size is None else nwp_size
        self.perceiver = OraclePerceiver(input_channels=N_SAT_CHANNELS, input_axis=2, num_freq_bands=6, max_freq=10,
                                         depth=self.total_seq_length, num_latents=num_latents, latent_dim=latent_dim,
                                         num_classes=PERCEIVER_OUTPUT_SIZE, weight_tie_layers=True)
        self.fc1 = nn.Linear(PERCEIVER_OUTPUT_SIZE, 256)
        self.fc2 = nn.Linear(256 + embedding_dem, 128)
        self.fc3 = nn.Linear(128, 64)
        self.fc4 = nn.Linear(64, 32)
        self.fc5 = nn.Linear(32, FC_OUTPUT_SIZE)
        if embedding_dem:
            self.pv_system_id_embedding = nn.Embedding(2048, embedding_dem)
        self.encoder_rnn = nn.GRU(FC_OUTPUT_SIZE + 1 + nwp_size, RNN_HIDDEN_SIZE, num_layers=2, batch_first=True)
        self.decoder_rnn = nn.GRU(FC_OUTPUT_SIZE + nwp_size, RNN_HIDDEN_SIZE, num_layers=2, batch_first=True)
        self.decoder_fc1 = nn.Linear(RNN_HIDDEN_SIZE, 8)
        self.decoder_fc2 = nn.Linear(8, 1)

    def forward(self, sat, nwp, yield_history, ids):
        """sat [B,C,T,H,W]; nwp [B,C,T_nwp,h,w]; yield_history = pv.pv_yield or gsp.gsp_yield [B,T,n]; ids [B,n]."""
        sat = sat[: self.batch_size].float()
        batch_size, n_chans, seq_len, width, height = sat.shape
        x = sat.permute(0, 2, 3, 4, 1).reshape(batch_size * seq_len, width, height, n_chans)
        out = self.perceiver(x).reshape(batch_size * seq_len, PERCEIVER_OUTPUT_SIZE)
        out = F.relu(self.fc1(out))
        if self.embedding_dem:
            i = ids[: self.batch_size, 0].long().repeat_interleave(self.total_seq_length)
            out = torch.cat((out, self.pv_system_id_embedding(i)), dim=1)
        out = F.relu(self.fc2(out))
        out = F.relu(self.fc3(out))
        out = F.relu(self.fc4(out))
        out = F.relu(self.fc5(out))
        out = out.reshape(batch_size, self.total_seq_length, FC_OUTPUT_SIZE)
        nwp = nwp[: self.batch_size].float().permute(0, 2, 1, 3, 4)
        b, nwp_seq_len, c, w, h = nwp.shape
        nwp = torch.cat([nwp, torch.zeros(b, seq_len - nwp_seq_len, c, w, h)], dim=1).reshape(b, seq_len, c * w * h)
        rnn_input = torch.cat((out, nwp), dim=2)
        n_hist = (self.history_len_5 if self.output_variable == "pv_yield" else self.history_len_30) + 1
        hist = yield_history[: self.batch_size][:, :n_hist, 0].unsqueeze(-1).float()
        encoder_input = torch.cat((rnn_input[:, :n_hist], hist), dim=2)
        _, encoder_hidden = self.encoder_rnn(encoder_input)
        decoder_output, _ = self.decoder_rnn(rnn_input[:, -self.forecast_len:], encoder_hidden)
        decoder_output = F.relu(self.decoder_fc1(decoder_output))
        return self.decoder_fc2(decoder_output).squeeze(dim=-1)


class _OracleHeadMixin:
    def _make_head(self, embedding_dem, rnn_extra):
        self.fc1 = nn.Linear(PERCEIVER_OUTPUT_SIZE, 256)
        self.fc2 = nn.Linear(256 + embedding_dem, 128)
        self.fc3 = nn.Linear(128, 64)
        self.fc4 = nn.Linear(64, 32)
        self.fc5 = nn.Linear(32, FC_OUTPUT_SIZE)
        if embedding_dem:
            self.pv_system_id_embedding = nn.Embedding(2048, embedding_dem)
        self.encoder_rnn = nn.GRU(FC_OUTPUT_SIZE + 1 + rnn_extra, RNN_HIDDEN_SIZE, num_layers=2, batch_first=True)
        self.decoder_rnn = nn.GRU(FC_OUTPUT_SIZE + rnn_extra, RNN_HIDDEN_SIZE, num_layers=2, batch_first=True)
        self.decoder_fc1 = nn.Linear(RNN_HIDDEN_SIZE, 8)
        self.decoder_fc2 = nn.Linear(8, 1)

    def _head(self, out, batch_size, yield_history, row_ids):
        out = F.relu(self.fc1(out.reshape(-1, PERCEIVER_OUTPUT_SIZE)))
        if self.embedding_dem:
            i = row_ids[: self.batch_size, 0].long().repeat_interleave(self.total_seq_length)
            out = torch.cat((out, self.pv_system_id_embedding(i)), dim=1)
        out = F.relu(self.fc5(F.relu(self.fc4(F.relu(self.fc3(F.relu(self.fc2(out))))))))
        rnn_input = out.reshape(batch_size, self.total_seq_length, FC_OUTPUT_SIZE)
        n_hist = (self.history_len_5 if self.output_variable == "pv_yield" else self.history_len_30) + 1
        hist = yield_history[: self.batch_size][:, :n_hist, 0].unsqueeze(-1).float()
        _, hidden = self.encoder_rnn(torch.cat((rnn_input[:, :n_hist], hist), dim=2))
        dec, _ = self.decoder_rnn(rnn_input[:, -self.forecast_len:], hidden)
        return self.decoder_fc2(F.relu(self.decoder_fc1(dec))).squeeze(dim=-1)


def _stack(sat, nwp):
    b, c, t, w, h = sat.shape
    sat = sat.permute(0, 2, 3, 4, 1).reshape(b * t, w, h, c)
    nwp = nwp.permute(0, 2, 3, 4, 1)
    _, tn, wn, hn, cn = nwp.shape
    nwp = torch.cat([nwp, torch.zeros(b, t - tn, wn, hn, cn)], dim=1).reshape(b * t, wn, hn, cn)
    return torch.cat((sat, nwp), dim=-1), b


class OraclePerceiverNwpSatModel(nn.Module, _OracleHeadMixin):
    """predict_pv_yield/models/perceiver/perceiver_nwp_sat.py:41-204."""

    def __init__(self, history_minutes, forecast_minutes, n_nwp_channels=10, batch_size=32, num_latents=128, latent_dim=64,
                 embedding_dem=16, output_variable="pv_yield"):
        super().__init__()
        from .conv3d_oracle import timestep_arithmetic
        for k, v in timestep_arithmetic(history_minutes, forecast_minutes, output_variable).items():
            setattr(self, k, v)
        self.batch_size, self.embedding_dem, self.output_variable = batch_size, embedding_dem, output_variable
        self.total_seq_length = history_minutes // 5 + forecast_minutes // 5 + 1
        self.perceiver = OraclePerceiver(input_channels=N_SAT_CHANNELS + n_nwp_channels, input_axis=2, num_freq_bands=6,
                                         max_freq=10, depth=self.total_seq_length, num_latents=num_latents,
                                         latent_dim=latent_dim, num_classes=PERCEIVER_OUTPUT_SIZE, weight_tie_layers=True)
        self._make_head(embedding_dem, 0)

    def forward(self, sat, nwp, yield_history, row_ids):
        data, b = _stack(sat[: self.batch_size].float(), nwp[: self.batch_size].float())
        return self._head(self.perceiver(data), b, yield_history, row_ids)


class OraclePerceiverConv3dNwpSatModel(nn.Module, _OracleHeadMixin):
    """predict_pv_yield/models/perceiver/perceiver_conv3d_nwp_sat.py:60-235."""

    def __init__(self, history_minutes, forecast_minutes, n_nwp_channels=10, batch_size=32, num_latents=128, latent_dim=64,
                 embedding_dem=16, output_variable="pv_yield", conv3d_channels=16, use_future_satellite_images=True):
        super().__init__()
        from .conv3d_oracle import OracleConv3dMaxPool, timestep_arithmetic
        for k, v in timestep_arithmetic(history_minutes, forecast_minutes, output_variable).items():
            setattr(self, k, v)
        self.batch_size, self.embedding_dem, self.output_variable = batch_size, embedding_dem, output_variable
        self.use_future_satellite_images = use_future_satellite_images
        self.total_seq_length = history_minutes // 5 + forecast_minutes // 5 + 1
        self.sat_conv3d_maxpool = OracleConv3dMaxPool(out_channels=conv3d_channels, in_channels=N_SAT_CHANNELS)
        self.nwp_conv3d_maxpool = OracleConv3dMaxPool(out_channels=conv3d_channels, in_channels=n_nwp_channels)
        self.perceiver = OraclePerceiver(input_channels=2 * conv3d_channels, input_axis=2, num_freq_bands=6, max_freq=10,
                                         depth=self.total_seq_length, num_latents=num_latents, latent_dim=latent_dim,
                                         num_classes=PERCEIVER_OUTPUT_SIZE, weight_tie_layers=True)
        self._make_head(embedding_dem, 0)

    def forward(self, sat, nwp, yield_history, row_ids):
        sat = sat[: self.batch_size].float()
        if not self.use_future_satellite_images:
            sat = sat.clone()
            sat[:, -self.forecast_len_5:] = 0          # dim 1 (channels), as written in the reference
        data, b = _stack(self.sat_conv3d_maxpool(sat), self.nwp_conv3d_maxpool(nwp[: self.batch_size].float()))
        return self._head(self.perceiver(data), b, yield_history, row_ids)


# ---- experiments/003_perceiver_processes_single_sat_image_then_rnn.py:95-253 (BASELINE configs[4]) -------------------------
class OracleExp003LitModel(nn.Module):
    """Same layer graph and state_dict names as the experiment's LitModel: Perceiver(depth 2, untied, 12 channels, 128 x 64
    latents, 512 classes) -> fc1 (+ Embedding(940, 16)) .. fc5 -> 2-layer GRU encoder / decoder over [8 features | NWP 40 |
    4 datetime features (| PV history)] -> decoder_fc1/2.  forward takes the experiment's dict batch."""

    def __init__(self, history_len=6, forecast_len=12):
        super().__init__()
        self.history_len, self.forecast_len = history_len, forecast_len
        self.total_seq_len = history_len + forecast_len + 1
        self.perceiver = OraclePerceiver(input_channels=12, input_axis=2, num_freq_bands=6, max_freq=10, depth=2,
                                         num_latents=128, latent_dim=64, num_classes=PERCEIVER_OUTPUT_SIZE)
        self.fc1 = nn.Linear(PERCEIVER_OUTPUT_SIZE, 256)
        self.fc2 = nn.Linear(256 + 16, 128)
        self.fc3 = nn.Linear(128, 64)
        self.fc4 = nn.Linear(64, 32)
        self.fc5 = nn.Linear(32, FC_OUTPUT_SIZE)
        self.pv_system_id_embedding = nn.Embedding(940, 16)
        nwp_size, n_datetime = 10 * 2 * 2, 4
        self.encoder_rnn = nn.GRU(FC_OUTPUT_SIZE + n_datetime + 1 + nwp_size, RNN_HIDDEN_SIZE, num_layers=2, batch_first=True)
        self.decoder_rnn = nn.GRU(FC_OUTPUT_SIZE + n_datetime + nwp_size, RNN_HIDDEN_SIZE, num_layers=2, batch_first=True)
        self.decoder_fc1 = nn.Linear(RNN_HIDDEN_SIZE, 8)
        self.decoder_fc2 = nn.Linear(8, 1)

    def forward(self, x):
        sat = x["sat_data"]
        b, t, w, h, c = sat.shape
        out = self.perceiver(sat.reshape(b * t, w, h, c)).reshape(b * t, PERCEIVER_OUTPUT_SIZE)
        out = F.relu(self.fc1(out))
        row = x["pv_system_row_number"].long().repeat_interleave(self.total_seq_len)
        out = torch.cat((out, self.pv_system_id_embedding(row)), dim=1)
        out = F.relu(self.fc5(F.relu(self.fc4(F.relu(self.fc3(F.relu(self.fc2(out))))))))
        out = out.reshape(b, self.total_seq_len, FC_OUTPUT_SIZE)
        nwp = x["nwp"].float().permute(0, 2, 1, 3, 4)
        nwp = nwp.reshape(b, nwp.shape[1], -1)
        rnn_input = torch.cat((out, nwp, x["hour_of_day_sin"].unsqueeze(-1), x["hour_of_day_cos"].unsqueeze(-1),
                               x["day_of_year_sin"].unsqueeze(-1), x["day_of_year_cos"].unsqueeze(-1)), dim=2)
        hist = x["pv_yield"][:, : self.history_len + 1].unsqueeze(-1)
        _, hidden = self.encoder_rnn(torch.cat((rnn_input[:, : self.history_len + 1], hist), dim=2))
        dec, _ = self.decoder_rnn(rnn_input[:, -self.forecast_len:], hidden)
        return self.decoder_fc2(F.relu(self.decoder_fc1(dec))).squeeze(dim=-1)

    def losses(self, batch):
        y_hat = self(batch)
        y = batch["pv_yield"][:, -self.forecast_len:]
        return F.mse_loss(y_hat, y), (y_hat - y).abs().mean()
