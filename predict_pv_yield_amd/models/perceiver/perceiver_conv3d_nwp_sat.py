"""Conv3dMaxPool — host-side mirror of predict_pv_yield/models/perceiver/perceiver_conv3d_nwp_sat.py:42-57.

Only the convolution + max-pool block of that file is built; the Perceiver `Model` that
consumes it depends on the third-party perceiver_pytorch package and is a later row.

  sat_conv3d  nn.Conv3d(in_channels, out_channels, kernel_size=(3,3,3), padding=(1,1,1))   (no ReLU)
  sat_maxpool nn.MaxPool3d(3, stride=(1,2,2), padding=(1,1,1))                             (time length kept)

Both run on the gfx950 kernels behind include/pv_yield_hip.h (pv_conv3d_general_*_f32, pv_maxpool3d_*_f32); the
modules are parameter / hyper-parameter holders with the reference's attribute (state_dict) names.
"""
from torch import nn

from ... import functional as Fn


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
