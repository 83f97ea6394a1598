"""CPU restatement of the Conv3D PV-yield model, its loss and its optimiser.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).  Never imported by
the product package.

The reference's arithmetic for this half of the path lives in third-party torch (unpinned:
requirements.txt:10); the same torch build runs on CPU here, so this restatement calls the SAME
operators the reference calls -- F.conv3d, F.relu, F.linear, torch.cat, torch.optim.Adam -- in the
order of:
  Model.__init__ / cnn_output_size   predict_pv_yield/models/conv3d/model.py:18-105 (:74-78)
  Model.forward                      predict_pv_yield/models/conv3d/model.py:107-156
  sat+nwp Model                      predict_pv_yield/models/conv3d/model_sat_nwp.py:14-270
  timestep arithmetic                predict_pv_yield/models/base_model.py:38-76
  loss / metrics                     predict_pv_yield/models/base_model.py:91-103
  optimiser                          predict_pv_yield/models/base_model.py:255-257
Pinned against the reference's own module source executed under import stubs
(tests/golden/make_conv3d_golden.py -> tests/golden/conv3d_small.npz, bit-exact on CPU).
WeightedLosses (nowcasting_utils, absent) is restated from its documented behaviour: parity unpinned
for mse_exp / mae_exp, which are logging-only.
"""
import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


def timestep_arithmetic(history_minutes: int, forecast_minutes: int, output_variable: str = "pv_yield") -> Dict[str, int]:
    """base_model.py:41-73."""
    d = dict(
        history_len_5=history_minutes // 5, forecast_len_5=forecast_minutes // 5,
        history_len_30=history_minutes // 30, forecast_len_30=forecast_minutes // 30,
        history_len_60=int(np.ceil(history_minutes / 60)), forecast_len_60=forecast_minutes // 60,
    )
    if output_variable == "pv_yield":
        d.update(forecast_len=d["forecast_len_5"], history_len=d["history_len_5"], number_of_samples_per_batch=128)
    else:
        d.update(forecast_len=d["forecast_len_30"], history_len=d["history_len_30"], number_of_samples_per_batch=32)
    d["number_of_pv_samples_per_batch"] = 128
    return d


class _RoundBF16(torch.autograd.Function):
    """Round to bf16 (nearest even) in forward AND in backward: models a bf16 tensor hand-off between two
    kernels of the MFMA path (activations forward, activation gradients backward)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


class _RoundWeightBF16(torch.autograd.Function):
    """bf16 operand copy of an f32 master weight; the gradient goes to the master unchanged."""

    @staticmethod
    def forward(ctx, w):
        return w.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


class OracleConv3dModel(nn.Module):
    """Same layer graph and attribute names (state_dict keys) as the reference Model.

    emulate_bf16=True restates, on the CPU and in f32 arithmetic, exactly WHICH values the bf16 MFMA path rounds
    (conv/fc1 operands, activations between conv layers, activation gradients between dgrad kernels) so the
    HIP path can be compared at accumulation-order tolerance instead of bf16-noise tolerance."""

    def __init__(self, include_pv_yield=True, include_nwp=True, forecast_minutes=30, history_minutes=60,
                 number_of_conv3d_layers=4, conv3d_channels=32, image_size_pixels=64, number_sat_channels=12,
                 fc1_output_features=128, fc2_output_features=128, fc3_output_features=64,
                 output_variable="pv_yield", emulate_bf16=False):
        super().__init__()
        self.emulate_bf16 = emulate_bf16
        self.include_pv_yield, self.include_nwp = include_pv_yield, include_nwp
        self.number_of_conv3d_layers = number_of_conv3d_layers
        self.number_of_nwp_features = 10 * 19 * 2 * 2
        self.output_variable = output_variable
        for k, v in timestep_arithmetic(history_minutes, forecast_minutes, output_variable).items():
            setattr(self, k, v)
        self.cnn_output_size = (conv3d_channels * ((image_size_pixels - 2 * number_of_conv3d_layers) ** 2)
                                * (self.forecast_len_5 + self.history_len_5 + 1 - 2 * number_of_conv3d_layers))
        self.sat_conv0 = nn.Conv3d(number_sat_channels, conv3d_channels, (3, 3, 3), padding=0)
        for i in range(number_of_conv3d_layers - 1):
            setattr(self, f"conv3d_{i + 1}", nn.Conv3d(conv3d_channels, conv3d_channels, (3, 3, 3), padding=0))
        self.fc1 = nn.Linear(self.cnn_output_size, fc1_output_features)
        self.fc2 = nn.Linear(fc1_output_features, fc2_output_features)
        fc3_in = fc2_output_features
        if include_pv_yield:
            fc3_in += self.number_of_samples_per_batch * (self.history_len_30 + 1)
        if include_nwp:
            self.fc_nwp = nn.Linear(self.number_of_nwp_features, 128)
            fc3_in += 128
        self.fc3 = nn.Linear(fc3_in, fc3_output_features)
        self.fc4 = nn.Linear(fc3_output_features, self.forecast_len)

    def forward(self, sat_data: torch.Tensor, yield_history: Optional[torch.Tensor] = None,
                nwp: Optional[torch.Tensor] = None, return_activations: bool = False):
        acts = []
        sat_data = sat_data.float()
        batch_size = sat_data.shape[0]
        if self.emulate_bf16:
            out = bf16_round(sat_data)
            layers = [self.sat_conv0] + [getattr(self, f"conv3d_{i + 1}") for i in range(self.number_of_conv3d_layers - 1)]
            for layer in layers:
                out = F.relu(F.conv3d(out, _RoundWeightBF16.apply(layer.weight), layer.bias))
                out = _RoundBF16.apply(out)
                acts.append(out)
            out = out.reshape(batch_size, self.cnn_output_size)
            out = F.relu(F.linear(out, _RoundWeightBF16.apply(self.fc1.weight), self.fc1.bias))
        else:
            out = F.relu(self.sat_conv0(sat_data))
            acts.append(out)
            for i in range(self.number_of_conv3d_layers - 1):
                out = F.relu(getattr(self, f"conv3d_{i + 1}")(out))
                acts.append(out)
            out = out.reshape(batch_size, self.cnn_output_size)
            out = F.relu(self.fc1(out))
        acts.append(out)
        out = F.relu(self.fc2(out))
        acts.append(out)
        if self.include_pv_yield:
            h = yield_history[:, : self.history_len_30 + 1].nan_to_num(nan=0.0).float()
            out = torch.cat((out, h.reshape(h.shape[0], h.shape[1] * h.shape[2])), dim=1)
        if self.include_nwp:
            out = torch.cat((out, F.relu(self.fc_nwp(nwp.float().flatten(start_dim=1)))), dim=1)
        out = F.relu(self.fc3(out))
        acts.append(out)
        out = self.fc4(out).reshape(batch_size, self.forecast_len)
        return (out, acts) if return_activations else out


class OracleConv3dSatNwpModel(nn.Module):
    """Restatement of predict_pv_yield/models/conv3d/model_sat_nwp.py:14-270 (same attribute / state_dict names,
    same join order: fc2 | yield history | pv_fc1 | nwp tower | id embedding).  Pinned against the reference
    module's own source by tests/golden/make_conv3d_golden.py (case "sat_nwp*")."""

    def __init__(self, include_pv_or_gsp_yield_history=True, include_nwp=True, forecast_minutes=30, history_minutes=60,
                 number_of_conv3d_layers=4, conv3d_channels=32, image_size_pixels=64, nwp_image_size_pixels=64,
                 number_sat_channels=12, number_nwp_channels=10, fc1_output_features=128, fc2_output_features=128,
                 fc3_output_features=64, output_variable="pv_yield", embedding_dem=16, include_pv_yield_history=True,
                 include_future_satellite=True, emulate_bf16=False, batch_size=32):
        super().__init__()
        self.emulate_bf16 = emulate_bf16
        self.batch_size = batch_size
        self.include_pv_or_gsp_yield_history, self.include_nwp = include_pv_or_gsp_yield_history, include_nwp
        self.include_pv_yield_history, self.include_future_satellite = include_pv_yield_history, include_future_satellite
        self.embedding_dem = embedding_dem
        self.number_of_conv3d_layers = number_of_conv3d_layers
        self.output_variable = output_variable
        for k, v in timestep_arithmetic(history_minutes, forecast_minutes, output_variable).items():
            setattr(self, k, v)
        t_sat = self.forecast_len_5 + self.history_len_5 + 1 if include_future_satellite else self.history_len_5 + 1
        self.cnn_output_size = conv3d_channels * ((image_size_pixels - 2 * number_of_conv3d_layers) ** 2) * t_sat
        self.nwp_cnn_output_size = (conv3d_channels * ((nwp_image_size_pixels - 2 * number_of_conv3d_layers) ** 2)
                                    * (self.forecast_len_60 + self.history_len_60 + 1))

        def tower(prefix, c_in):
            for i in range(number_of_conv3d_layers):
                setattr(self, f"{prefix}_conv{i}",
                        nn.Conv3d(c_in if i == 0 else conv3d_channels, conv3d_channels, (3, 3, 3), padding=(1, 0, 0)))

        tower("sat", number_sat_channels)
        self.fc1 = nn.Linear(self.cnn_output_size, fc1_output_features)
        self.fc2 = nn.Linear(fc1_output_features, fc2_output_features)
        if include_nwp:
            tower("nwp", number_nwp_channels)
            self.nwp_fc1 = nn.Linear(self.nwp_cnn_output_size, fc1_output_features)
            self.nwp_fc2 = nn.Linear(fc1_output_features, 128)
        if embedding_dem:
            self.pv_system_id_embedding = nn.Embedding(940, embedding_dem)
        if include_pv_yield_history:
            self.pv_fc1 = nn.Linear(self.number_of_pv_samples_per_batch * (self.history_len_5 + 1), 128)
        fc3_in = fc2_output_features
        if include_pv_or_gsp_yield_history:
            fc3_in += self.number_of_samples_per_batch * (self.history_len_30 + 1)
        if include_nwp:
            fc3_in += 128
        if embedding_dem:
            fc3_in += embedding_dem
        if include_pv_yield_history:
            fc3_in += 128
        self.fc3 = nn.Linear(fc3_in, fc3_output_features)
        self.fc4 = nn.Linear(fc3_output_features, self.forecast_len)

    def _tower(self, prefix, data, fc1, flat):
        out = bf16_round(data) if self.emulate_bf16 else data
        for i in range(self.number_of_conv3d_layers):
            layer = getattr(self, f"{prefix}_conv{i}")
            if self.emulate_bf16:
                out = _RoundBF16.apply(F.relu(F.conv3d(out, _RoundWeightBF16.apply(layer.weight), layer.bias,
                                                        padding=(1, 0, 0))))
            else:
                out = F.relu(layer(out))
        out = out.reshape(data.shape[0], flat)
        w = _RoundWeightBF16.apply(fc1.weight) if self.emulate_bf16 else fc1.weight
        return F.relu(F.linear(out, w, fc1.bias))

    def forward(self, sat_data, pv_yield=None, gsp_yield=None, nwp=None, ids=None):
        """ids: pv.pv_system_row_number (output_variable pv_yield) or gsp.gsp_id (gsp_yield), [B, n]."""
        sat_data = sat_data.float()
        batch_size = sat_data.shape[0]
        if not self.include_future_satellite:
            sat_data = sat_data[:, :, : self.history_len_5 + 1]
        out = F.relu(self.fc2(self._tower("sat", sat_data, self.fc1, self.cnn_output_size)))
        if self.include_pv_or_gsp_yield_history:
            src = gsp_yield if self.output_variable == "gsp_yield" else pv_yield
            h = src[:, : self.history_len_30 + 1].nan_to_num(nan=0.0).float()
            out = torch.cat((out, h.reshape(h.shape[0], h.shape[1] * h.shape[2])), dim=1)
        if self.include_pv_yield_history:
            h = pv_yield[:, : self.history_len_5 + 1, :128].nan_to_num(nan=0.0).float()
            out = torch.cat((out, F.relu(self.pv_fc1(h.reshape(h.shape[0], h.shape[1] * h.shape[2])))), dim=1)
        if self.include_nwp:
            o = F.relu(self.nwp_fc2(self._tower("nwp", nwp.float(), self.nwp_fc1, self.nwp_cnn_output_size)))
            out = torch.cat((out, o), dim=1)
        if self.embedding_dem:
            out = torch.cat((out, self.pv_system_id_embedding(ids[0 : self.batch_size, 0].long())), dim=1)
        out = F.relu(self.fc3(out))
        return self.fc4(out).reshape(batch_size, self.forecast_len)


class OracleConv3dNwpModel(nn.Module):
    """Restatement of predict_pv_yield/models/conv3d/model_nwp.py:14-153: the NWP tower (3x3x3 convolutions, padding
    (1,0,0)) -> nwp_fc1 -> nwp_fc2 -> fc3 -> fc4.  The id embedding and pv_fc1 are registered (model_nwp.py:112-121)
    but carry no data in forward (model_nwp.py:127-153).  Pinned against the reference module's own source by
    tests/golden/make_conv3d_nwp_golden.py."""

    def __init__(self, include_pv_or_gsp_yield_history=True, include_nwp=True, forecast_minutes=30, history_minutes=60,
                 number_of_conv3d_layers=4, conv3d_channels=32, nwp_image_size_pixels=64, number_nwp_channels=10,
                 fc1_output_features=128, fc2_output_features=128, fc3_output_features=64, output_variable="gsp_yield",
                 embedding_dem=16, include_pv_yield_history=True, include_future_satellite=True, emulate_bf16=False):
        super().__init__()
        self.emulate_bf16 = emulate_bf16
        self.number_of_conv3d_layers = number_of_conv3d_layers
        self.output_variable = output_variable
        for k, v in timestep_arithmetic(history_minutes, forecast_minutes, output_variable).items():
            setattr(self, k, v)
        self.nwp_cnn_output_size = (conv3d_channels * ((nwp_image_size_pixels - 2 * number_of_conv3d_layers) ** 2)
                                    * (self.forecast_len_60 + self.history_len_60 + 1))
        for i in range(number_of_conv3d_layers):
            setattr(self, f"nwp_conv{i}",
                    nn.Conv3d(number_nwp_channels if i == 0 else conv3d_channels, conv3d_channels, (3, 3, 3), padding=(1, 0, 0)))
        self.nwp_fc1 = nn.Linear(self.nwp_cnn_output_size, fc1_output_features)
        self.nwp_fc2 = nn.Linear(fc1_output_features, 128)
        if embedding_dem:
            self.pv_system_id_embedding = nn.Embedding(940, embedding_dem)
        if include_pv_yield_history:
            self.pv_fc1 = nn.Linear(self.number_of_pv_samples_per_batch * (self.history_len_5 + 1), 128)
        self.fc3 = nn.Linear(128, fc3_output_features)
        self.fc4 = nn.Linear(fc3_output_features, self.forecast_len)

    _tower = OracleConv3dSatNwpModel._tower

    def forward(self, nwp):
        out = F.relu(self.nwp_fc2(self._tower("nwp", nwp.float(), self.nwp_fc1, self.nwp_cnn_output_size)))
        out = F.relu(self.fc3(out))
        return self.fc4(out).reshape(nwp.shape[0], self.forecast_len)


class OracleConv3dMaxPool(nn.Module):
    """predict_pv_yield/models/perceiver/perceiver_conv3d_nwp_sat.py:42-57: Conv3d(k 3, pad 1) -> MaxPool3d(3, stride
    (1,2,2), pad 1).  Pinned by tests/golden/make_flow_model_golden.py (the reference class itself, under stubs)."""

    def __init__(self, out_channels: int, in_channels: int):
        super().__init__()
        self.sat_conv3d = nn.Conv3d(in_channels, out_channels, kernel_size=(3, 3, 3), padding=(1, 1, 1))
        self.sat_maxpool = nn.MaxPool3d(3, stride=(1, 2, 2), padding=(1, 1, 1))

    def forward(self, x):
        return self.sat_maxpool(self.sat_conv3d(x))


class OracleLitAutoEncoder(nn.Module):
    """Restatement of LitAutoEncoder, notebooks/13_3d_conv_with_optical_flow_predictions.ipynb:962-1027 (forward,
    mse loss, Adam lr 1e-4).  Pinned by tests/golden/make_flow_model_golden.py, which executes the notebook cell."""

    def __init__(self):
        super().__init__()
        k, p = (2, 3, 3), (0, 1, 1)
        self.conv = nn.Sequential(
            nn.Conv3d(2, 16, k, padding=p), nn.ReLU(), nn.Conv3d(16, 32, k, padding=p), nn.ReLU(),
            nn.Conv3d(32, 32, k, padding=p), nn.ReLU(), nn.Conv3d(32, 1, k, padding=p, stride=(1, 2, 2)))

    def forward(self, history, flow_prediction, horizon):
        images = torch.cat((history, flow_prediction.unsqueeze(1)), dim=1)
        b, n, h, w = images.shape
        hz = horizon.view(-1, 1, 1, 1, 1).expand(b, 1, n, h, w)
        return self.conv(torch.cat((images.unsqueeze(1), hz), dim=1))

    def loss(self, history, flow_prediction, horizon, target):
        return F.mse_loss(self(history, flow_prediction, horizon).squeeze(), target)

    def train_steps(self, history, flow_prediction, horizon, target, n_steps=1):
        opt = torch.optim.Adam(self.parameters(), lr=0.0001)
        losses = []
        for _ in range(n_steps):
            opt.zero_grad()
            loss = self.loss(history, flow_prediction, horizon, target)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        return losses


def weighted_losses_weights(forecast_length: int) -> torch.Tensor:
    """nowcasting_utils WeightedLosses: w_i = exp(-ln2 * i), normalised to mean 1 (f32)."""
    w = torch.FloatTensor([math.exp(-math.log(2) * i) for i in range(forecast_length)])
    return w / w.sum() * len(w)


def forecast_losses(y_hat: torch.Tensor, y: torch.Tensor):
    """(mse, nmae, mse_exp, mae_exp) as in base_model.py:98-103; nmae is the training loss (:146)."""
    mse = F.mse_loss(y_hat, y)
    nmae = (y_hat - y).abs().mean()
    w = weighted_losses_weights(y_hat.shape[1]).to(y_hat.device)
    mse_exp = torch.mean(w * (y_hat - y) ** 2)
    mae_exp = torch.mean(w * torch.abs(y_hat - y))
    return mse, nmae, mse_exp, mae_exp


def horizon_metrics(y_hat: torch.Tensor, y: torch.Tensor):
    """(mse per forecast step, mae per forecast step): nowcasting_utils' mse_each_forecast_horizon /
    mae_each_forecast_horizon as called at base_model.py:123-124 = means over the batch axis."""
    return torch.mean((y_hat - y) ** 2, dim=0), torch.mean(torch.abs(y_hat - y), dim=0)


def logged_horizon_metrics(y_hat: torch.Tensor, y: torch.Tensor, forecast_len_30: int, tag: str) -> Dict[str, float]:
    """The dictionary base_model.py:126-135 logs: metrics_mse and metrics_mae are BOTH keyed
    `MSE_forecast_horizon_{i}/{tag}` for i in range(forecast_len_30) and merged as {**mse, **mae}, so the value that
    survives under the MSE name is the MAE."""
    mse_h, mae_h = horizon_metrics(y_hat, y)
    metrics_mse = {f"MSE_forecast_horizon_{i}/{tag}": float(mse_h[i]) for i in range(forecast_len_30)}
    metrics_mae = {f"MSE_forecast_horizon_{i}/{tag}": float(mae_h[i]) for i in range(forecast_len_30)}
    return {**metrics_mse, **metrics_mae}


def validation_results_rows(model_output: np.ndarray, gsp_yield: np.ndarray, gsp_capacity: np.ndarray, gsp_id: np.ndarray,
                            t0_datetime_utc_ns: np.ndarray, forecast_len_30: int):
    """Rows of the validation results table, base_model.py:223-239 -> nowcasting_utils.make_validation_results:
    (t0, target time = t0 + 30 min * (i + 1), gsp_id, actual MW, forecast MW) per example and 30-minute step, MW =
    normalised value * capacity.  The reference pins the row count and these five columns
    (tests/models/baseline/test_baseline_model_gsp.py:104-111)."""
    capacity = gsp_capacity[:, -forecast_len_30:, 0]
    predictions = model_output[:, -forecast_len_30:] * capacity
    truths = gsp_yield[:, -forecast_len_30:, 0] * capacity
    rows = []
    for b in range(predictions.shape[0]):
        t0 = np.datetime64(int(t0_datetime_utc_ns[b]), "ns")
        for i in range(forecast_len_30):
            rows.append((t0, t0 + np.timedelta64(30 * (i + 1), "m"), int(gsp_id[b, 0]), float(truths[b, i]),
                         float(predictions[b, i])))
    return rows


def select_target(yield_tensor: torch.Tensor, forecast_len: int, batch_size: int = 32) -> torch.Tensor:
    """y = yield[0:batch_size, -forecast_len:, 0]  (base_model.py:91-95)."""
    return yield_tensor[0:batch_size, -forecast_len:, 0]


def make_optimizer(model: nn.Module) -> torch.optim.Optimizer:
    """base_model.py:255-257."""
    return torch.optim.Adam(model.parameters(), lr=0.0005)


def train_steps(model: OracleConv3dModel, sat: torch.Tensor, yield_tensor: torch.Tensor, n_steps: int = 1,
                optimizer: Optional[torch.optim.Optimizer] = None):
    """n optimiser steps on one fixed batch; returns the list of nmae losses (before each step)."""
    opt = optimizer or make_optimizer(model)
    losses = []
    for _ in range(n_steps):
        opt.zero_grad()
        y_hat = model(sat)
        _, nmae, _, _ = forecast_losses(y_hat, select_target(yield_tensor, model.forecast_len))
        nmae.backward()
        opt.step()
        losses.append(float(nmae.detach()))
    return losses


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    """Round-to-nearest-even to bf16 and back: the operand rounding of the MFMA path."""
    return t.to(torch.bfloat16).to(torch.float32)
