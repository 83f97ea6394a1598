"""LitAutoEncoder — host-side mirror of the model in notebooks/13_3d_conv_with_optical_flow_predictions.ipynb:962-1027
(SURVEY.md §8a row a-16): the 3-D CNN that refines an optical-flow prediction of the next satellite image.

  input  x[HISTORICAL_SAT_IMAGES]     [B, 4, 128, 128]   normalised history (every third 5-minute frame, ending at t0)
         x[OPTICAL_FLOW_PREDICTIONS]  [B, 128, 128]      t0 image advected to the target time (optical_flow.py)
         x[FORECAST_HORIZON]          [B]                normalise_forecast_horizon(seconds)
         x[TARGET_SAT_IMAGE]          [B, 64, 64]        centre crop of the true image at the target time
  graph  cat(history, flow prediction) -> [B, 1, 5, 128, 128]; the horizon is broadcast as a second channel;
         Conv3d(2->16->32->32, kernel (2,3,3), padding (0,1,1)) + ReLU, then Conv3d(32->1, stride (1,2,2)) -> [B,1,1,64,64]
  loss   F.mse_loss(y_hat.squeeze(), target); Adam(lr=1e-4)

Same attribute / state_dict names as the notebook (`conv.0`, `conv.2`, `conv.4`, `conv.6`).  nn.Sequential is a
parameter holder; every convolution, the loss and the optimiser run on the gfx950 kernels behind
include/pv_yield_hip.h (pv_conv3d_general_*_f32, pv_mse_loss_f32, pv_adam_step_f32).
"""
import numpy as np
import torch
from torch import nn

from ... import lightning as pl

FORECAST_HORIZON = "FORECAST_HORIZON"
HISTORICAL_SAT_IMAGES = "HISTORICAL_SAT_IMAGES"
OPTICAL_FLOW_PREDICTIONS = "OPTICAL_FLOW_PREDICTIONS"
TARGET_SAT_IMAGE = "TARGET_SAT_IMAGE"

CHANNELS = 32
KERNEL = (2, 3, 3)  # depth (timestep), height, width
PADDING = (0, 1, 1)

# 13_…ipynb:655-668
MINUTES_PER_TIMESTEP = 5
SECONDS_PER_TIMESTEP = MINUTES_PER_TIMESTEP * 60
FCST_HORIZON_SEQ = np.arange(1, 24, dtype=np.float32) * SECONDS_PER_TIMESTEP
FCST_HORIZON_MEAN = FCST_HORIZON_SEQ.mean()
FCST_HORIZON_STD = FCST_HORIZON_SEQ.std()


def normalise_forecast_horizon(forecast_horizon):
    """forecast_horizon: number of seconds -> standardised float32."""
    forecast_horizon = np.float32(forecast_horizon)
    forecast_horizon -= FCST_HORIZON_MEAN
    forecast_horizon /= FCST_HORIZON_STD
    return forecast_horizon


class LitAutoEncoder(pl.LightningModule):
    def __init__(self):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv3d(in_channels=2, out_channels=CHANNELS // 2, kernel_size=KERNEL, padding=PADDING),
            nn.ReLU(),
            nn.Conv3d(in_channels=CHANNELS // 2, out_channels=CHANNELS, kernel_size=KERNEL, padding=PADDING),
            nn.ReLU(),
            nn.Conv3d(in_channels=CHANNELS, out_channels=CHANNELS, kernel_size=KERNEL, padding=PADDING),
            nn.ReLU(),
            nn.Conv3d(in_channels=CHANNELS, out_channels=1, kernel_size=KERNEL, padding=PADDING, stride=(1, 2, 2)),
        )

    def forward(self, x):
        from ... import functional as Fn
        images = torch.cat((x[HISTORICAL_SAT_IMAGES], x[OPTICAL_FLOW_PREDICTIONS].unsqueeze(1)), dim=1)
        if not images.is_cuda:
            raise RuntimeError("predict_pv_yield_amd LitAutoEncoder runs on the MI355X only: move the module and the "
                               "batch to cuda (there is no CPU fallback)")
        batch_size, n_channels, height, width = images.shape
        # forecast horizon as an extra channel on every timestep
        forecast_horizon = x[FORECAST_HORIZON].to(images.dtype).view(-1, 1, 1, 1, 1).expand(
            batch_size, 1, n_channels, height, width)
        out = torch.cat((images.unsqueeze(1), forecast_horizon), dim=1).float()  # [B, 2, depth, H, W]
        layers = list(self.conv)
        prev_relu = False
        for i, layer in enumerate(layers):
            if not isinstance(layer, nn.Conv3d):
                continue
            relu = i + 1 < len(layers) and isinstance(layers[i + 1], nn.ReLU)
            # a ReLU output feeds a conv: that conv's dgrad applies the ReLU gate, so this layer sees a pre-gated dy
            feeds_conv = relu and any(isinstance(m, nn.Conv3d) for m in layers[i + 2:i + 3])
            out = Fn.conv3d_general_f32(out, layer.weight, layer.bias, stride=tuple(layer.stride),
                                        padding=tuple(layer.padding), relu=relu, x_is_relu_output=prev_relu,
                                        dy_pregated=feeds_conv)
            prev_relu = relu
        return out

    def _training_or_validation_step(self, batch, is_train_step):
        from ... import functional as Fn
        y_hat = self(batch).squeeze()
        y = batch[TARGET_SAT_IMAGE]
        loss = Fn.mse_loss(y_hat, y.float())
        tag = "Loss/Train" if is_train_step else "Loss/Validation"
        self.log_dict({tag: loss}, on_step=is_train_step, on_epoch=True)
        return loss

    def training_step(self, batch, batch_idx):
        return self._training_or_validation_step(batch, is_train_step=True)

    def validation_step(self, batch, batch_idx):
        return self._training_or_validation_step(batch, is_train_step=False)

    def configure_optimizers(self):
        from ...optim import HipAdam
        return HipAdam(self.parameters(), lr=0.0001)
