"""PerceiverModel — host-side mirror of predict_pv_yield/models/perceiver/perceiver.py:42-200.

Same constructor kwargs and defaults, same attribute / state_dict names (`perceiver.*`, `fc1..fc5`,
`pv_system_id_embedding`, `encoder_rnn`, `decoder_rnn`, `decoder_fc1/2`), same `forward(x: dict | BatchML) ->
[B, forecast_len]`.  Every time step of the satellite stack is one Perceiver "example" (B*T images of H*W positions):
cross-attention from 128 latents into the image, weight-tied over depth = T layers; the per-step features feed a GRU
encoder over the history (+ yield history) and a GRU decoder over the forecast steps.  All arithmetic runs in the
gfx950 kernels behind include/pv_yield_hip.h: perceiver_core.Perceiver (pv_gemm_f32 on the f32 matrix cores,
LayerNorm / softmax / GEGLU kernels), pv_linear_*_f32 for the head, pv_embedding_*, pv_gru_seq_* for the RNNs.
"""
from typing import Iterable

import torch
from torch import nn

from ...data.batch import BatchML
from ..base_model import BaseModel
from .perceiver_core import Perceiver

# nowcasting_dataset.consts (absent): the first ten NWP variables and the satellite channels (HRV first)
NWP_VARIABLE_NAMES = ("t", "dswrf", "prate", "r", "sde", "si10", "vis", "lcc", "mcc", "hcc")
SAT_VARIABLE_NAMES = ("HRV", "IR_016", "IR_039", "IR_087", "IR_097", "IR_108", "IR_120", "IR_134", "VIS006", "VIS008",
                      "WV_062", "WV_073")

params = dict(
    batch_size=32,
    history_minutes=30,  #: Number of timesteps of history, not including t0.
    forecast_minutes=120,  #: Number of timesteps of forecast.
    image_size_pixels=64,
    nwp_channels=NWP_VARIABLE_NAMES[0:10],
    sat_channels=SAT_VARIABLE_NAMES[1:],
)

TOTAL_SEQ_LEN = params["history_minutes"] // 5 + params["forecast_minutes"] // 5 + 1
NWP_SIZE = len(params["nwp_channels"]) * 64 * 64  # channels x width x height
N_DATETIME_FEATURES = 4
PERCEIVER_OUTPUT_SIZE = 512
FC_OUTPUT_SIZE = 8
RNN_HIDDEN_SIZE = 16


def perceiver_head(model, out, x, batch_size, id_from_output_variable: bool, rnn_extra=None):
    """Everything after the Perceiver, shared by the three Perceiver models (perceiver.py:131-200,
    perceiver_nwp_sat.py:140-204, perceiver_conv3d_nwp_sat.py:172-235): fc1 -> [system-id embedding] -> fc2..fc5 -> per-step
    features [B, T, 8] (+ rnn_extra [B, T, E]) -> GRU encoder over the history (+ yield history) -> GRU decoder over the
    forecast steps -> decoder_fc1/2 -> [B, forecast_len]."""
    from ... import functional as Fn
    from ... import perceiver_functional as PF
    out = out.reshape(-1, PERCEIVER_OUTPUT_SIZE)
    out = Fn.linear_f32(out, model.fc1.weight, model.fc1.bias, relu=True)

    # ********************** Embedding of PV system ID ********************
    if model.embedding_dem:
        if id_from_output_variable and model.output_variable != "pv_yield":
            id = x.gsp.gsp_id[0 : model.batch_size, 0]
        else:
            id = x.pv.pv_system_row_number[0 : model.batch_size, 0]
        id = id.to(device=out.device, dtype=torch.int64).repeat_interleave(model.total_seq_length)
        out = torch.cat((out, Fn.embedding(model.pv_system_id_embedding.weight, id)), dim=1)

    # Fully connected layers.
    out = Fn.linear_f32(out, model.fc2.weight, model.fc2.bias, relu=True)
    out = Fn.linear_f32(out, model.fc3.weight, model.fc3.bias, relu=True)
    out = Fn.linear_f32(out, model.fc4.weight, model.fc4.bias, relu=True)
    out = Fn.linear_f32(out, model.fc5.weight, model.fc5.bias, relu=True)

    # ******************* PREP DATA FOR RNN *******************************
    out = out.reshape(batch_size, model.total_seq_length, FC_OUTPUT_SIZE)
    rnn_input = out if rnn_extra is None else torch.cat((out, rnn_extra), dim=2)

    if model.output_variable == "pv_yield":
        # the history of the pv yield of this system
        pv_yield_history = x.pv.pv_yield[0 : model.batch_size][:, : model.history_len_5 + 1, 0].unsqueeze(-1).float()
        encoder_input = torch.cat((rnn_input[:, : model.history_len_5 + 1], pv_yield_history), dim=2)
    elif model.output_variable == "gsp_yield":
        gsp_history = x.gsp.gsp_yield[0 : model.batch_size][:, : model.history_len_30 + 1, 0].unsqueeze(-1).float()
        encoder_input = torch.cat((rnn_input[:, : model.history_len_30 + 1], gsp_history), dim=2)

    _, encoder_hidden = PF.gru(encoder_input, model.encoder_rnn)
    decoder_output, _ = PF.gru(rnn_input[:, -model.forecast_len :], model.decoder_rnn, encoder_hidden)
    # decoder_output is shape batch_size, seq_len, rnn_hidden_size

    b, t, h = decoder_output.shape
    decoder_output = Fn.linear_f32(decoder_output.reshape(b * t, h), model.decoder_fc1.weight, model.decoder_fc1.bias, relu=True)
    decoder_output = Fn.linear_f32(decoder_output, model.decoder_fc2.weight, model.decoder_fc2.bias, relu=False)
    return decoder_output.reshape(b, t)


def make_perceiver_head(model, rnn_extra_size: int):
    """The layers perceiver_head() uses, with the reference's attribute names (state_dict keys)."""
    model.fc1 = nn.Linear(in_features=PERCEIVER_OUTPUT_SIZE, out_features=256)
    model.fc2 = nn.Linear(in_features=256 + model.embedding_dem, out_features=128)
    model.fc3 = nn.Linear(in_features=128, out_features=64)
    model.fc4 = nn.Linear(in_features=64, out_features=32)
    model.fc5 = nn.Linear(in_features=32, out_features=FC_OUTPUT_SIZE)
    if model.embedding_dem:
        model.pv_system_id_embedding = nn.Embedding(num_embeddings=2048, embedding_dim=model.embedding_dem)
    model.encoder_rnn = nn.GRU(input_size=FC_OUTPUT_SIZE + 1 + rnn_extra_size, hidden_size=RNN_HIDDEN_SIZE, num_layers=2,
                               batch_first=True)   # plus 1 for history
    model.decoder_rnn = nn.GRU(input_size=FC_OUTPUT_SIZE + rnn_extra_size, hidden_size=RNN_HIDDEN_SIZE, num_layers=2,
                               batch_first=True)
    model.decoder_fc1 = nn.Linear(in_features=RNN_HIDDEN_SIZE, out_features=8)
    model.decoder_fc2 = nn.Linear(in_features=8, out_features=1)


def require_cuda_input(t, who):
    if not t.is_cuda:
        raise RuntimeError(f"predict_pv_yield_amd {who} runs on the MI355X only: move the module and the batch to cuda "
                           "(there is no CPU fallback)")


class PerceiverModel(BaseModel):

    name = "perceiver"

    def __init__(
        self,
        history_minutes: int = params["history_minutes"],
        forecast_minutes: int = params["forecast_minutes"],
        nwp_channels: Iterable[str] = params["nwp_channels"],
        batch_size: int = 32,
        num_latents: int = 128,
        latent_dim: int = 64,
        embedding_dem: int = 16,
        output_variable: str = "pv_yield",
        operand_dtype: str = "f32",
    ):
        # operand_dtype (new, optional): "bf16" runs the attention products on the bf16 matrix cores (Lightning precision=16)
        self.history_minutes = history_minutes
        self.forecast_minutes = forecast_minutes
        self.nwp_channels = nwp_channels
        self.batch_size = batch_size
        self.num_latents = num_latents
        self.latent_dim = latent_dim
        self.embedding_dem = embedding_dem
        self.output_variable = output_variable

        self.total_seq_length = self.history_minutes // 5 + self.forecast_minutes // 5 + 1

        super().__init__()

        self.perceiver = Perceiver(
            input_channels=len(params["sat_channels"]),
            input_axis=2,
            num_freq_bands=6,
            max_freq=10,
            depth=self.total_seq_length,
            num_latents=self.num_latents,
            latent_dim=self.latent_dim,
            num_classes=PERCEIVER_OUTPUT_SIZE,
            weight_tie_layers=True,
            operand_dtype=operand_dtype,
        )

        make_perceiver_head(self, rnn_extra_size=NWP_SIZE)

    def forward(self, x):
        if type(x) == dict:
            x = BatchML(**x)
        # ******************* Satellite imagery *************************
        # Shape: batch_size, channel, seq_length, height, width
        sat_data = x.satellite.data[0 : self.batch_size].float()
        require_cuda_input(sat_data, "PerceiverModel")
        batch_size, n_chans, seq_len, width, height = sat_data.shape

        # Stack timesteps as examples (to make a large batch), channels last
        sat_data = sat_data.permute(0, 2, 3, 4, 1)
        new_batch_size = batch_size * seq_len
        sat_data = sat_data.reshape(new_batch_size, width, height, n_chans)

        out = self.perceiver(sat_data)

        # *********************** NWP Data (rides along into the RNNs, flattened) ************************************
        # Shape: batch_size, channel, seq_length, width, height -> seq_len to dim 1
        nwp_data = x.nwp.data[0 : self.batch_size].float().permute(0, 2, 1, 3, 4)
        batch_size, nwp_seq_len, n_nwp_chans, nwp_width, nwp_height = nwp_data.shape
        # nwp gets the same seq_len as sat (zero padded)
        nwp_data_zeros = torch.zeros(size=(batch_size, seq_len - nwp_seq_len, n_nwp_chans, nwp_width, nwp_height),
                                     device=nwp_data.device)
        nwp_data = torch.cat([nwp_data, nwp_data_zeros], dim=1)
        nwp_data = nwp_data.reshape(batch_size, seq_len, n_nwp_chans * nwp_width * nwp_height)
        return perceiver_head(self, out, x, batch_size, id_from_output_variable=True, rnn_extra=nwp_data)
