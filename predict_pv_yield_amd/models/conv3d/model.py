"""Conv3D PV/GSP-yield model — host-side mirror of predict_pv_yield/models/conv3d/model.py:14-156.

Same constructor kwargs and defaults (model.py:18-32), same attribute / state_dict names
(`sat_conv0`, `conv3d_{i}`, `fc1..fc4`, `fc_nwp`), same `forward(x: dict | BatchML) -> [B, forecast_len]`.
The layers are plain parameter holders (nn.Conv3d / nn.Linear are used for their parameters and default
initialisation only); every FLOP of forward and backward runs in the hand-written gfx950 kernels behind
include/pv_yield_hip.h:

  precision="bf16" (default)  NDHWC bf16 activations, MFMA implicit-GEMM conv (fwd, dgrad, wgrad), bf16
                              streaming fc1, f32 accumulation everywhere, f32 master weights + Adam
  precision="fp32"            reference layout, exact f32 FMA kernels (tight parity with torch CPU)

New optional knobs: `precision`, `future_frames` ("true" = use the batch's future
satellite frames like the reference does; "optical_flow" = replace the forecast_len_5 future frames by
HIP Farnebäck-advected ones, i.e. the reference's `# TODO: Use optical flow` made real).
"""
import logging

import torch
from torch import nn

from ...data.batch import BatchML
from ..base_model import BaseModel

logging.basicConfig()
_LOG = logging.getLogger("predict_pv_yield_amd")


# fc1.weight's columns in the tower's own (t, h, w, c) order for precision="bf16" (no NCDHW epilogue in the last layer, no
# repack of fc1's input gradient: -42 us of the 1.54 ms step); False restores the reference's column order in memory
FC1_CHANNELS_LAST = True


class Model(BaseModel):

    name = "conv3d"

    def __init__(
        self,
        include_pv_yield: bool = True,
        include_nwp: bool = True,
        forecast_minutes: int = 30,
        history_minutes: int = 60,
        number_of_conv3d_layers: int = 4,
        conv3d_channels: int = 32,
        image_size_pixels: int = 64,
        number_sat_channels: int = 12,
        fc1_output_features: int = 128,
        fc2_output_features: int = 128,
        fc3_output_features: int = 64,
        output_variable: str = "pv_yield",
        precision: str = "bf16",
        future_frames: str = "true",
    ):
        self.include_pv_yield = include_pv_yield
        self.include_nwp = include_nwp
        self.number_of_conv3d_layers = number_of_conv3d_layers
        self.number_of_nwp_features = 10 * 19 * 2 * 2
        self.fc1_output_features = fc1_output_features
        self.fc2_output_features = fc2_output_features
        self.fc3_output_features = fc3_output_features
        self.forecast_minutes = forecast_minutes
        self.history_minutes = history_minutes
        self.output_variable = output_variable
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        if future_frames not in ("true", "optical_flow"):
            raise ValueError("future_frames must be 'true' or 'optical_flow'")
        self.__dict__["_precision"] = precision      # read-only from here on (property below): fc1's stored layout follows it
        self.future_frames = future_frames
        self.number_sat_channels = number_sat_channels
        self.conv3d_channels = conv3d_channels

        super().__init__()

        # model.py:74-78
        self.cnn_output_size = (
            conv3d_channels
            * ((image_size_pixels - 2 * self.number_of_conv3d_layers) ** 2)
            * (self.forecast_len_5 + self.history_len_5 + 1 - 2 * self.number_of_conv3d_layers)
        )
        if self.cnn_output_size <= 0:
            raise ValueError("image / sequence too small for the number of valid 3x3x3 convolutions")

        self.sat_conv0 = nn.Conv3d(number_sat_channels, conv3d_channels, kernel_size=(3, 3, 3), padding=0)
        for i in range(0, self.number_of_conv3d_layers - 1):
            setattr(self, f"conv3d_{i + 1}", nn.Conv3d(conv3d_channels, conv3d_channels, kernel_size=(3, 3, 3), padding=0))

        self.fc1 = nn.Linear(in_features=self.cnn_output_size, out_features=self.fc1_output_features)
        self.fc2 = nn.Linear(in_features=self.fc1_output_features, out_features=self.fc2_output_features)

        fc3_in_features = self.fc2_output_features
        if include_pv_yield:
            fc3_in_features += self.number_of_samples_per_batch * (self.history_len_30 + 1)
        if include_nwp:
            self.fc_nwp = nn.Linear(in_features=self.number_of_nwp_features, out_features=128)
            fc3_in_features += 128
        self.fc3 = nn.Linear(in_features=fc3_in_features, out_features=self.fc3_output_features)
        self.fc4 = nn.Linear(in_features=self.fc3_output_features, out_features=self.forecast_len)

        # bf16 tower: fc1 consumes the last conv activation as the tower keeps it ([B, T, H, W, C]) -- the columns of
        # fc1.weight are stored in that order; state_dict() / load_state_dict() speak the reference's (_fc1_layout.py)
        self._fc1_k_channels = 0
        if FC1_CHANNELS_LAST and precision == "bf16" and conv3d_channels == 32 and self._bf16_supported():
            from . import _fc1_layout
            _fc1_layout.install(self, conv3d_channels)

    # ------------------------------------------------------------------------------------------
    def _conv_layers(self):
        return [self.sat_conv0] + [getattr(self, f"conv3d_{i + 1}") for i in range(self.number_of_conv3d_layers - 1)]

    @property
    def precision(self) -> str:
        """ "bf16" | "fp32", fixed at construction: the bf16 tower stores fc1.weight's columns channels-last
        (_fc1_layout.py); flipping the arithmetic afterwards would multiply an NCDHW flatten by a column-permuted weight."""
        return self.__dict__["_precision"]

    @precision.setter
    def precision(self, value):
        raise AttributeError("Model.precision is fixed at construction (fc1.weight's stored column order depends on it): "
                             "build Model(precision=...) and load_state_dict() instead")

    def _mark_fc1_layout(self) -> None:
        from . import _fc1_layout
        _fc1_layout.mark(self)

    # The layout mark of fc1.weight is a Python attribute of the Parameter object; whatever creates fresh Parameter objects
    # (copy.deepcopy / pickle: __setstate__; .to() / .half() / .cuda() of a module whose parameters get replaced: _apply)
    # loses it, while the MODULE attribute `_fc1_k_channels` -- the truth -- survives.  Re-applied after each of them and at the
    # top of forward(), so HipAdam(model.parameters()) sees the mark however the module came to be.
    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._mark_fc1_layout()
        return out

    def __setstate__(self, state):
        super().__setstate__(state)
        self._mark_fc1_layout()

    def _bf16_supported(self) -> bool:
        from ._tower import bf16_tower_supported
        return bf16_tower_supported(self.number_sat_channels, self.conv3d_channels, self.cnn_output_size)

    def _satellite_input(self, x: BatchML) -> torch.Tensor:
        sat = x.satellite.data
        if self.future_frames == "optical_flow":
            if getattr(sat, "_pv_advected", False):      # optical_flow.AdvectingLoader already ran the pipeline
                return sat
            if sat.dtype == torch.int16:
                # config 3 proper: the batch carries the OBSERVED frames only, as raw 10-bit counts in
                # the zarr's time-major order [B, T_obs, C, H, W] (13_...ipynb:415-441).  Flow is computed on the
                # counts, frames are normalised, and the forecast_len_5 future slices are advected from the last
                # observed frame -- one device-resident pipeline straight into the model input [B, C, T, H, W].
                from ...optical_flow import advect_future_frames
                if sat.dim() != 5 or sat.shape[1] != self.history_len_5 + 1 or sat.shape[2] != self.number_sat_channels:
                    raise ValueError(f"raw satellite counts must be [B, {self.history_len_5 + 1}, "
                                     f"{self.number_sat_channels}, H, W] int16, got {tuple(sat.shape)}")
                return advect_future_frames(sat, n_future=self.forecast_len_5)
            from ...optical_flow import replace_future_frames_with_flow
            return replace_future_frames_with_flow(sat.float(), n_future=self.forecast_len_5)
        return sat.float()  # [B, C, T, H, W]  (model.py:112-114)

    def forward(self, x):
        if type(x) == dict:
            x = BatchML(**x)
        sat_data = self._satellite_input(x)
        from ... import functional as Fn
        from ._tower import conv_tower_fc1
        self._mark_fc1_layout()

        batch_size = sat_data.shape[0]
        out = conv_tower_fc1(sat_data, self._conv_layers(), self.fc1, self.number_sat_channels, self.conv3d_channels,
                             (0, 0, 0), self.cnn_output_size, self.precision == "bf16" and self._bf16_supported(),
                             fc1_channels_last=bool(self._fc1_k_channels))
        out = Fn.linear_f32(out, self.fc2.weight, self.fc2.bias, relu=True)

        if self.include_pv_yield:
            # model.py:130-136 (note the 30-minute history length, also for 5-minute PV data)
            pv_yield_history = x[self.output_variable][:, : self.history_len_30 + 1].nan_to_num(nan=0.0).float()
            pv_yield_history = pv_yield_history.reshape(pv_yield_history.shape[0],
                                                        pv_yield_history.shape[1] * pv_yield_history.shape[2])
            out = torch.cat((out, pv_yield_history), dim=1)
        if self.include_nwp:
            nwp_data = x["nwp"].float().flatten(start_dim=1)
            out_nwp = Fn.linear_f32(nwp_data, self.fc_nwp.weight, self.fc_nwp.bias, relu=True)
            out = torch.cat((out, out_nwp), dim=1)

        out = Fn.linear_f32(out, self.fc3.weight, self.fc3.bias, relu=True)
        out = Fn.linear_f32(out, self.fc4.weight, self.fc4.bias, relu=False)
        return out.reshape(batch_size, self.forecast_len)
