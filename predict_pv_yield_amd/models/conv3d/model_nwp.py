"""NWP-only Conv3D model — host-side mirror of predict_pv_yield/models/conv3d/model_nwp.py:14-153.

Same constructor kwargs and defaults (model_nwp.py:18-35), same attribute / state_dict names and registration order
(`nwp_conv{i}`, `nwp_fc1`, `nwp_fc2`, `pv_system_id_embedding`, `pv_fc1`, `fc3`, `fc4`), same
`forward(x: dict | BatchML) -> [B, forecast_len]`.  As in the reference the class keeps the name "conv3d_sat_nwp"
(model_nwp.py:16), builds the id embedding and `pv_fc1` when their switches are on but never reads them in
`forward` (model_nwp.py:127-153: only the NWP tower -> nwp_fc1 -> nwp_fc2 -> fc3 -> fc4 carries data), so those two
parameter groups receive no gradient and the optimiser leaves them untouched.

The tower is the one of model_sat_nwp.py — 3x3x3 convolutions with padding (1,0,0) over the hourly NWP frames — and
runs in the same gfx950 kernels (include/pv_yield_hip.h: bf16 MFMA conv + nwp_fc1 by default, f32 small layers).
"""
import logging

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
        nwp_image_size_pixels: int = 64,
        number_nwp_channels: int = 10,
        fc1_output_features: int = 128,
        fc2_output_features: int = 128,
        fc3_output_features: int = 64,
        output_variable: str = "gsp_yield",
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
        self.conv3d_channels = conv3d_channels
        self.embedding_dem = embedding_dem
        self.include_pv_yield_history = include_pv_yield_history
        self.include_future_satellite = include_future_satellite
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        self.precision = precision

        super().__init__()

        if nwp_image_size_pixels <= 2 * number_of_conv3d_layers:
            raise ValueError("image too small for the number of 3x3x3 convolutions")
        # model_nwp.py:82-86: padding (1,0,0) keeps every hourly step, each layer trims 2 px
        self.nwp_cnn_output_size = (
            conv3d_channels
            * ((nwp_image_size_pixels - 2 * self.number_of_conv3d_layers) ** 2)
            * (self.forecast_len_60 + self.history_len_60 + 1)
        )

        for i in range(self.number_of_conv3d_layers):
            setattr(self, f"nwp_conv{i}", nn.Conv3d(number_nwp_channels if i == 0 else conv3d_channels, conv3d_channels,
                                                    kernel_size=(3, 3, 3), padding=(1, 0, 0)))
        self.nwp_fc1 = nn.Linear(in_features=self.nwp_cnn_output_size, out_features=self.fc1_output_features)
        self.nwp_fc2 = nn.Linear(in_features=self.fc1_output_features, out_features=self.number_of_nwp_features)

        # built but unused by forward, exactly as model_nwp.py:112-121
        if self.embedding_dem:
            self.pv_system_id_embedding = nn.Embedding(num_embeddings=940, embedding_dim=self.embedding_dem)
        if self.include_pv_yield_history:
            self.pv_fc1 = nn.Linear(in_features=self.number_of_pv_samples_per_batch * (self.history_len_5 + 1),
                                    out_features=128)

        self.fc3 = nn.Linear(in_features=self.number_of_nwp_features, out_features=self.fc3_output_features)
        self.fc4 = nn.Linear(in_features=self.fc3_output_features, out_features=self.forecast_len)

    def forward(self, x):
        if type(x) == dict:
            x = BatchML(**x)
        from ... import functional as Fn
        from ._tower import bf16_tower_supported, conv_tower_fc1

        nwp_data = x.nwp.data.float()  # [B, C, T60, H, W]
        batch_size = nwp_data.shape[0]
        convs = [getattr(self, f"nwp_conv{i}") for i in range(self.number_of_conv3d_layers)]
        use_bf16 = self.precision == "bf16" and bf16_tower_supported(self.number_nwp_channels, self.conv3d_channels,
                                                                     self.nwp_cnn_output_size)
        out = conv_tower_fc1(nwp_data, convs, self.nwp_fc1, self.number_nwp_channels, self.conv3d_channels, (1, 0, 0),
                             self.nwp_cnn_output_size, use_bf16)
        out = Fn.linear_f32(out, self.nwp_fc2.weight, self.nwp_fc2.bias, relu=True)
        out = Fn.linear_f32(out, self.fc3.weight, self.fc3.bias, relu=True)
        out = Fn.linear_f32(out, self.fc4.weight, self.fc4.bias, relu=False)
        return out.reshape(batch_size, self.forecast_len)
