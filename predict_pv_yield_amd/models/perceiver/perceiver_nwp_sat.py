"""Perceiver over satellite + NWP channels — host-side mirror of predict_pv_yield/models/perceiver/perceiver_nwp_sat.py:41-204.

The NWP fields (same pixel size as the satellite crop) are concatenated to the satellite channels of every time step
before the Perceiver; the GRU encoder / decoder see the per-step features only.  Same constructor kwargs, attribute /
state_dict names and forward contract as the reference; all arithmetic on the HIP kernels (perceiver.py: perceiver_head).
"""
from typing import Iterable

import torch

from ...data.batch import BatchML
from ..base_model import BaseModel
from .perceiver import (PERCEIVER_OUTPUT_SIZE, make_perceiver_head, params, perceiver_head, require_cuda_input)
from .perceiver_core import Perceiver


def stack_sat_and_nwp(sat_data: torch.Tensor, nwp_data: torch.Tensor):
    """sat [B,C,T,H,W], nwp [B,Cn,Tn,H,W] -> [B*T, H, W, C+Cn] (time steps as examples, channels last; NWP zero-padded to T
    steps) — perceiver_nwp_sat.py:112-136."""
    batch_size, n_chans, seq_len, width, height = sat_data.shape
    sat_data = sat_data.permute(0, 2, 3, 4, 1).reshape(batch_size * seq_len, width, height, n_chans)
    nwp_data = nwp_data.permute(0, 2, 3, 4, 1)
    _, nwp_seq_len, nwp_width, nwp_height, n_nwp_chans = nwp_data.shape
    zeros = torch.zeros(size=(batch_size, seq_len - nwp_seq_len, nwp_width, nwp_height, n_nwp_chans), device=nwp_data.device)
    nwp_data = torch.cat([nwp_data, zeros], dim=1).reshape(batch_size * seq_len, nwp_width, nwp_height, n_nwp_chans)
    assert nwp_width == width, f"data {nwp_width} should be the model {width}"
    assert nwp_height == height
    return torch.cat((sat_data, nwp_data), dim=-1), batch_size


class Model(BaseModel):

    name = "perceiver_nwp_sat"

    def __init__(
        self,
        history_minutes: int,
        forecast_minutes: int,
        nwp_channels: Iterable[str] = params["nwp_channels"],
        batch_size: int = 32,
        num_latents: int = 128,
        latent_dim: int = 64,
        embedding_dem: int = 16,
        output_variable: str = "pv_yield",
    ):
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
            input_channels=len(params["sat_channels"]) + len(nwp_channels),
            input_axis=2,
            num_freq_bands=6,
            max_freq=10,
            depth=self.total_seq_length,
            num_latents=self.num_latents,
            latent_dim=self.latent_dim,
            num_classes=PERCEIVER_OUTPUT_SIZE,
            weight_tie_layers=True,
        )
        make_perceiver_head(self, rnn_extra_size=0)

    def forward(self, x):
        if type(x) == dict:
            x = BatchML(**x)
        sat_data = x.satellite.data[0 : self.batch_size].float()
        require_cuda_input(sat_data, "perceiver_nwp_sat.Model")
        data, batch_size = stack_sat_and_nwp(sat_data, x.nwp.data[0 : self.batch_size].float())
        out = self.perceiver(data)
        # (the reference takes the embedding id from pv_system_row_number whatever the output variable)
        return perceiver_head(self, out, x, batch_size, id_from_output_variable=False)
