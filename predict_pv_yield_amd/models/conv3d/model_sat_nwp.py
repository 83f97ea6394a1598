"""Conv3D satellite + NWP model — host-side mirror of predict_pv_yield/models/conv3d/model_sat_nwp.py:14-270.

Same constructor kwargs and defaults (model_sat_nwp.py:18-37), same attribute / state_dict names
(`sat_conv{i}`, `nwp_conv{i}`, `fc1`, `fc2`, `nwp_fc1`, `nwp_fc2`, `pv_system_id_embedding`, `pv_fc1`, `fc3`,
`fc4`), same `forward(x: dict | BatchML) -> [B, forecast_len]`.  Two conv towers with padding (1,0,0) (time
length preserved, 2 px lost per layer in H and W), each followed by two fully-connected layers; the joins
(yield history, 5-minute PV history through `pv_fc1`, the NWP tower, the system-id embedding) are concatenated
in the reference's order before fc3/fc4.  All arithmetic runs in the gfx950 kernels behind
include/pv_yield_hip.h (conv towers + fc1/nwp_fc1 on the bf16 MFMA path by default, the small layers and the
embedding gather in f32); the modules are parameter holders only.
"""
import logging

import torch
from torch import nn

from ...data.batch import BatchML
from ..base_model import BaseModel

logging.basicConfig()
_LOG = logging.getLogger("predict_pv_yield_amd")


class Model(BaseModel):

    name = "conv3d_sat_nwp"

    def __init__(
        self,
        include_pv_or_gsp_yield_history: bool = True,
        include_nwp: bool = True,
        forecast_minutes: int = 30,
        history_minutes: int = 60,
        number_of_conv3d_layers: int = 4,
        conv3d_channels: int = 32,
        image_size_pixels: int = 64,
        nwp_image_size_pixels: int = 64,
        number_sat_channels: int = 12,
        number_nwp_channels: int = 10,
        fc1_output_features: int = 128,
        fc2_output_features: int = 128,
        fc3_output_features: int = 64,
        output_variable: str = "pv_yield",
        embedding_dem: int = 16,
        include_pv_yield_history: int = True,
        include_future_satellite: int = True,
        precision: str = "bf16",
    ):
        self.include_pv_or_gsp_yield_history = include_pv_or_gsp_yield_history
        self.include_nwp = include_nwp
        self.number_of_conv3d_layers = number_of_conv3d_layers
        self.number_of_nwp_features = 128
        self.fc1_output_features = fc1_output_features
        self.fc2_output_features = fc2_output_features
        self.fc3_output_features = fc3_output_features
        self.forecast_minutes = forecast_minutes
        self.history_minutes = history_minutes
        self.output_variable = output_variable
        self.number_nwp_channels = number_nwp_channels
        self.number_sat_channels = number_sat_channels
        self.conv3d_channels = conv3d_channels
        self.embedding_dem = embedding_dem
        self.include_pv_yield_history = include_pv_yield_history
        self.include_future_satellite = include_future_satellite
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        self.precision = precision

        super().__init__()

        # model_sat_nwp.py:84-98: padding (1,0,0) keeps every time step
        if include_future_satellite:
            cnn_output_size_time = self.forecast_len_5 + self.history_len_5 + 1
        else:
            cnn_output_size_time = self.history_len_5 + 1
        self.cnn_output_size = (
            conv3d_channels * ((image_size_pixels - 2 * self.number_of_conv3d_layers) ** 2) * cnn_output_size_time
        )
        self.nwp_cnn_output_size = (
            conv3d_channels
            * ((nwp_image_size_pixels - 2 * self.number_of_conv3d_layers) ** 2)
            * (self.forecast_len_60 + self.history_len_60 + 1)
        )
        if image_size_pixels <= 2 * number_of_conv3d_layers or (
                include_nwp and nwp_image_size_pixels <= 2 * number_of_conv3d_layers):
            raise ValueError("image too small for the number of 3x3x3 convolutions")

        def tower(prefix, c_in):
            setattr(self, f"{prefix}_conv0", nn.Conv3d(c_in, conv3d_channels, kernel_size=(3, 3, 3), padding=(1, 0, 0)))
            for i in range(0, self.number_of_conv3d_layers - 1):
                setattr(self, f"{prefix}_conv{i + 1}",
                        nn.Conv3d(conv3d_channels, conv3d_channels, kernel_size=(3, 3, 3), padding=(1, 0, 0)))

        tower("sat", number_sat_channels)
        self.fc1 = nn.Linear(in_features=self.cnn_output_size, out_features=self.fc1_output_features)
        self.fc2 = nn.Linear(in_features=self.fc1_output_features, out_features=self.fc2_output_features)

        if include_nwp:
            tower("nwp", number_nwp_channels)
            self.nwp_fc1 = nn.Linear(in_features=self.nwp_cnn_output_size, out_features=self.fc1_output_features)
            self.nwp_fc2 = nn.Linear(in_features=self.fc1_output_features, out_features=self.number_of_nwp_features)

        if self.embedding_dem:
            self.pv_system_id_embedding = nn.Embedding(num_embeddings=940, embedding_dim=self.embedding_dem)

        if self.include_pv_yield_history:
            self.pv_fc1 = nn.Linear(in_features=self.number_of_pv_samples_per_batch * (self.history_len_5 + 1),
                                    out_features=128)

        fc3_in_features = self.fc2_output_features
        if include_pv_or_gsp_yield_history:
            fc3_in_features += self.number_of_samples_per_batch * (self.history_len_30 + 1)
        if include_nwp:
            fc3_in_features += 128
        if self.embedding_dem:
            fc3_in_features += self.embedding_dem
        if self.include_pv_yield_history:
            fc3_in_features += 128

        self.fc3 = nn.Linear(in_features=fc3_in_features, out_features=self.fc3_output_features)
        self.fc4 = nn.Linear(in_features=self.fc3_output_features, out_features=self.forecast_len)

    # ------------------------------------------------------------------------------------------
    def _tower_layers(self, prefix):
        return [getattr(self, f"{prefix}_conv{i}") for i in range(self.number_of_conv3d_layers)]

    def forward(self, x):
        if type(x) == dict:
            x = BatchML(**x)
        from ... import functional as Fn
        from ._tower import bf16_tower_supported, conv_tower_fc1

        # ******************* satellite tower (model_sat_nwp.py:180-200) *************************
        sat_data = x.satellite.data.float()  # [B, C, T, H, W]
        batch_size = sat_data.shape[0]
        if not self.include_future_satellite:
            sat_data = sat_data[:, :, : self.history_len_5 + 1].contiguous()
        bf16 = self.precision == "bf16"
        out = conv_tower_fc1(sat_data, self._tower_layers("sat"), self.fc1, self.number_sat_channels,
                             self.conv3d_channels, (1, 0, 0), self.cnn_output_size,
                             bf16 and bf16_tower_supported(self.number_sat_channels, self.conv3d_channels,
                                                           self.cnn_output_size))
        out = Fn.linear_f32(out, self.fc2.weight, self.fc2.bias, relu=True)

        # yield history at the 30-minute length (model_sat_nwp.py:203-219)
        if self.include_pv_or_gsp_yield_history:
            h = x[self.output_variable][:, : self.history_len_30 + 1].nan_to_num(nan=0.0).float()
            out = torch.cat((out, h.reshape(h.shape[0], h.shape[1] * h.shape[2])), dim=1)

        # 5-minute PV history of the first 128 systems through pv_fc1 (model_sat_nwp.py:222-233)
        if self.include_pv_yield_history:
            h = x.pv.pv_yield[:, : self.history_len_5 + 1, :128].nan_to_num(nan=0.0).float()
            h = h.reshape(h.shape[0], h.shape[1] * h.shape[2])
            out = torch.cat((out, Fn.linear_f32(h, self.pv_fc1.weight, self.pv_fc1.bias, relu=True)), dim=1)

        # ******************* NWP tower (model_sat_nwp.py:236-250) *************************
        if self.include_nwp:
            nwp_data = x.nwp.data.float()
            out_nwp = conv_tower_fc1(nwp_data, self._tower_layers("nwp"), self.nwp_fc1, self.number_nwp_channels,
                                     self.conv3d_channels, (1, 0, 0), self.nwp_cnn_output_size,
                                     bf16 and bf16_tower_supported(self.number_nwp_channels, self.conv3d_channels,
                                                                   self.nwp_cnn_output_size))
            out_nwp = Fn.linear_f32(out_nwp, self.nwp_fc2.weight, self.nwp_fc2.bias, relu=True)
            out = torch.cat((out, out_nwp), dim=1)

        # ******************* system-id embedding (model_sat_nwp.py:253-263) *************************
        if self.embedding_dem:
            if self.output_variable == "pv_yield":
                id = x.pv.pv_system_row_number[0 : self.batch_size, 0]
            else:
                id = x.gsp.gsp_id[0 : self.batch_size, 0]
            out = torch.cat((out, Fn.embedding(self.pv_system_id_embedding.weight, id)), dim=1)

        out = Fn.linear_f32(out, self.fc3.weight, self.fc3.bias, relu=True)
        out = Fn.linear_f32(out, self.fc4.weight, self.fc4.bias, relu=False)
        return out.reshape(batch_size, self.forecast_len)
