"""Conv3dMaxPool — host-side mirror of predict_pv_yield/models/perceiver/perceiver_conv3d_nwp_sat.py:42-57.

`Model` (perceiver_conv3d_nwp_sat.py:60-235) puts one Conv3dMaxPool in front of the satellite stack and one in front of
the NWP stack (halving H and W), concatenates the two feature maps per time step and feeds them to the Perceiver; the
head is the shared perceiver_head() of perceiver.py.

  sat_conv3d  nn.Conv3d(in_channels, out_channels, kernel_size=(3,3,3), padding=(1,1,1))   (no ReLU)
  sat_maxpool nn.MaxPool3d(3, stride=(1,2,2), padding=(1,1,1))                             (time length kept)

Both run on the gfx950 kernels behind include/pv_yield_hip.h (pv_conv3d_general_*_f32, pv_maxpool3d_*_f32); the
modules are parameter / hyper-parameter holders with the reference's attribute (state_dict) names.
"""
from typing import Iterable

import torch
from torch import nn

from ... import functional as Fn
from ...data.batch import BatchML
from ..base_model import BaseModel


class Conv3dMaxPool(nn.Module):

    def __init__(self, out_channels: int, in_channels: int):
        super().__init__()
        # convolution layer, padded so the output is the same size
        self.sat_conv3d = nn.Conv3d(in_channels=in_channels, out_channels=out_channels, kernel_size=(3, 3, 3),
                                    padding=(1, 1, 1))
        # max pool, keeps the time sequence the same length
        self.sat_maxpool = nn.MaxPool3d(3, stride=(1, 2, 2), padding=(1, 1, 1))

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("predict_pv_yield_amd Conv3dMaxPool runs on the MI355X only (there is no CPU fallback)")
        x = Fn.conv3d_general_f32(x.float(), self.sat_conv3d.weight, self.sat_conv3d.bias, stride=(1, 1, 1),
                                  padding=(1, 1, 1), relu=False)
        return Fn.maxpool3d_f32(x, kernel=(3, 3, 3), stride=(1, 2, 2), padding=(1, 1, 1))


class Model(BaseModel):

    name = "perceiver_conv3d_nwp_sat"

    def __init__(
        self,
        history_minutes: int,
        forecast_minutes: int,
        nwp_channels: Iterable[str] = None,
        batch_size: int = 32,
        num_latents: int = 128,
        latent_dim: int = 64,
        embedding_dem: int = 16,
        output_variable: str = "pv_yield",
        conv3d_channels: int = 16,
        use_future_satellite_images: bool = True,  # option not to use future sat images
    ):
        from .perceiver import PERCEIVER_OUTPUT_SIZE, make_perceiver_head, params
        from .perceiver_core import Perceiver
        nwp_channels = params["nwp_channels"] if nwp_channels is None else nwp_channels
        self.history_minutes = history_minutes
        self.forecast_minutes = forecast_minutes
        self.nwp_channels = nwp_channels
        self.batch_size = batch_size
        self.num_latents = num_latents
        self.latent_dim = latent_dim
        self.embedding_dem = embedding_dem
        self.output_variable = output_variable
        self.use_future_satellite_images = use_future_satellite_images
        self.total_seq_length = self.history_minutes // 5 + self.forecast_minutes // 5 + 1

        super().__init__()

        self.sat_conv3d_maxpool = Conv3dMaxPool(out_channels=conv3d_channels, in_channels=len(params["sat_channels"]))
        self.nwp_conv3d_maxpool = Conv3dMaxPool(out_channels=conv3d_channels, in_channels=len(nwp_channels))
        self.perceiver = Perceiver(
            input_channels=2 * conv3d_channels,
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
        from .perceiver import perceiver_head, require_cuda_input
        from .perceiver_nwp_sat import stack_sat_and_nwp
        if type(x) == dict:
            x = BatchML(**x)
        sat_data = x.satellite.data[0 : self.batch_size].float()
        require_cuda_input(sat_data, "perceiver_conv3d_nwp_sat.Model")
        if not self.use_future_satellite_images:
            # as written in the reference (perceiver_conv3d_nwp_sat.py:151-152) the slice runs over dim 1, the CHANNEL
            # axis of [B, C, T, H, W]; kept as is, on a copy so that the caller's batch is not modified
            sat_data = sat_data.clone()
            sat_data[:, -self.forecast_len_5 :] = 0
        sat_data = self.sat_conv3d_maxpool(sat_data)
        nwp_data = self.nwp_conv3d_maxpool(x.nwp.data[0 : self.batch_size].float())
        data, batch_size = stack_sat_and_nwp(sat_data, nwp_data)
        out = self.perceiver(data)
        return perceiver_head(self, out, x, batch_size, id_from_output_variable=False)
