"""Generates tests/golden/conv3d_small.npz by EXECUTING THE REFERENCE'S OWN MODULE SOURCE
(/root/reference/predict_pv_yield/models/conv3d/model.py, model_sat_nwp.py + base_model.py) on the CPU.

Run here (the reference tree does not travel to the GPU box):   python tests/golden/make_conv3d_golden.py

The reference modules import packages that are not installed in this image (pytorch_lightning,
nowcasting_utils, nowcasting_dataset, nowcasting_dataloader).  None of them contributes arithmetic to
Model.forward / configure_optimizers; they are replaced by import stubs *in this generator only*:
  pytorch_lightning.LightningModule -> torch.nn.Module (+ a no-op log_dict)
  nowcasting_dataloader.batch.BatchML -> a plain attribute container
  nowcasting_utils WeightedLosses / metrics / validation / visualization -> inert stand-ins
    (get_mse_exp/get_mae_exp restated as documented: weights exp(-ln2*i) normalised to mean 1; LOGGING ONLY)
What the fixture pins is therefore exactly the reference's layer graph, reshape/cat order, target slicing,
loss (F.mse_loss, abs().mean()) and optimiser (torch.optim.Adam(lr=0.0005)) as written in its source,
evaluated by this image's torch CPU ops.  The vectors (inputs, initial parameters, outputs) are data; no
reference source text is stored.
"""
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conv3d_small.npz")

KW = dict(include_pv_yield=False, include_nwp=False, forecast_minutes=30, history_minutes=30,
          number_of_conv3d_layers=4, conv3d_channels=32, image_size_pixels=10, number_sat_channels=11,
          fc1_output_features=16, fc2_output_features=16, fc3_output_features=16)
KW_PV = dict(KW, include_pv_yield=True, include_nwp=True, output_variable="gsp_yield", forecast_minutes=60)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Section(types.SimpleNamespace):
    pass


class BatchML:
    def __init__(self, **sections):
        for k, v in sections.items():
            setattr(self, k, _Section(**v) if isinstance(v, dict) else v)

    def __getitem__(self, key):
        return {"pv_yield": lambda: self.pv.pv_yield, "gsp_yield": lambda: self.gsp.gsp_yield,
                "nwp": lambda: self.nwp.data}[key]()


class WeightedLosses:
    def __init__(self, decay_rate=None, forecast_length=6):
        w = torch.FloatTensor([math.exp(-math.log(2) * i) for i in range(forecast_length)])
        self.weights = w / w.sum() * len(w)

    def get_mse_exp(self, output, target):
        return torch.mean(self.weights * (output - target) ** 2)

    def get_mae_exp(self, output, target):
        return torch.mean(self.weights * torch.abs(output - target))


def install_stubs():
    class LightningModule(torch.nn.Module):
        current_epoch = 0

        def log_dict(self, d, **kw):
            self.logged = {k: float(v) for k, v in d.items()}

    _stub("pytorch_lightning", LightningModule=LightningModule)
    _stub("nowcasting_utils")
    _stub("nowcasting_utils.visualization")
    _stub("nowcasting_utils.visualization.visualization", plot_example=None)
    _stub("nowcasting_utils.visualization.line", plot_batch_results=None)
    _stub("nowcasting_utils.models")
    _stub("nowcasting_utils.models.loss", WeightedLosses=WeightedLosses)
    _stub("nowcasting_utils.models.metrics", mae_each_forecast_horizon=None, mse_each_forecast_horizon=None)
    _stub("nowcasting_utils.metrics")
    _stub("nowcasting_utils.metrics.validation", make_validation_results=None, save_validation_results_to_logger=None)
    _stub("nowcasting_dataset")
    _stub("nowcasting_dataset.data_sources")
    _stub("nowcasting_dataset.data_sources.nwp")
    _stub("nowcasting_dataset.data_sources.nwp.nwp_data_source", NWP_VARIABLE_NAMES=())
    _stub("nowcasting_dataloader")
    _stub("nowcasting_dataloader.batch", BatchML=BatchML)


def checksum(t: torch.Tensor, n=64):
    f = t.detach().double().flatten()
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long()
    return np.concatenate([[f.sum().item(), f.abs().sum().item()], t.detach().flatten()[idx].double().numpy()])


def run_case(Model, kw, tag, out):
    torch.manual_seed(518)
    model = Model(**kw)
    t = kw["history_minutes"] // 5 + kw["forecast_minutes"] // 5 + 1
    g = torch.Generator().manual_seed(1234)
    sat = torch.randn(2, 11, t, kw["image_size_pixels"], kw["image_size_pixels"], generator=g)
    pv = torch.rand(2, t, 128, generator=g)
    t30 = kw["history_minutes"] // 30 + kw["forecast_minutes"] // 30 + 1
    gsp = torch.rand(2, t30, 32, generator=g)
    nwp = torch.randn(2, 10, 19, 2, 2, generator=g)
    batch = {"satellite": {"data": sat}, "pv": {"pv_yield": pv}, "gsp": {"gsp_yield": gsp}, "nwp": {"data": nwp}}
    out[f"{tag}/sat"], out[f"{tag}/pv"], out[f"{tag}/gsp"], out[f"{tag}/nwp"] = sat.numpy(), pv.numpy(), gsp.numpy(), nwp.numpy()
    for k, v in model.state_dict().items():
        out[f"{tag}/init/{k}"] = v.numpy().copy()
    y_hat = model(batch)
    out[f"{tag}/y_hat"] = y_hat.detach().numpy().copy()
    opt = model.configure_optimizers()
    losses = []
    for step in range(3):
        opt.zero_grad()
        loss = model.training_step(batch, 0)
        loss.backward()
        if step == 0:
            out[f"{tag}/logged"] = np.array([model.logged[k] for k in ("MSE/Train", "NMAE/Train", "MSE_EXP/Train", "MAE_EXP/Train")])
            for k, p in model.named_parameters():
                out[f"{tag}/grad/{k}"] = checksum(p.grad)
        opt.step()
        losses.append(float(loss.detach()))
        if step in (0, 2):
            for k, p in model.named_parameters():
                out[f"{tag}/step{step + 1}/{k}"] = checksum(p)
    out[f"{tag}/losses"] = np.array(losses)
    out[f"{tag}/attrs"] = np.array([model.cnn_output_size, model.forecast_len, model.history_len_5, model.forecast_len_5,
                                    model.history_len_30, model.forecast_len_30, model.history_len_60,
                                    model.number_of_samples_per_batch])


SN = dict(include_pv_or_gsp_yield_history=False, include_nwp=True, forecast_minutes=60, history_minutes=60,
          number_of_conv3d_layers=4, conv3d_channels=32, image_size_pixels=10, nwp_image_size_pixels=10,
          number_sat_channels=11, number_nwp_channels=10, fc1_output_features=16, fc2_output_features=16,
          fc3_output_features=16, output_variable="gsp_yield", include_pv_yield_history=True)
SN_PV = dict(SN, include_pv_or_gsp_yield_history=True, output_variable="pv_yield", include_future_satellite=False,
             include_pv_yield_history=False,
             forecast_minutes=30)


def run_case_sat_nwp(Model, kw, tag, out):
    """model_sat_nwp.Model on a 2-example batch: forward, 3 train steps (losses, gradient and parameter checksums)."""
    torch.manual_seed(518)
    model = Model(**kw)
    t5 = kw["history_minutes"] // 5 + kw["forecast_minutes"] // 5 + 1
    t30 = kw["history_minutes"] // 30 + kw["forecast_minutes"] // 30 + 1
    t60 = int(np.ceil(kw["history_minutes"] / 60)) + kw["forecast_minutes"] // 60 + 1
    g = torch.Generator().manual_seed(4321)
    sat = torch.randn(2, 11, t5, kw["image_size_pixels"], kw["image_size_pixels"], generator=g)
    pv = torch.rand(2, t5, 128, generator=g)
    pv[0, 1, 3] = float("nan")                      # nan_to_num path (model_sat_nwp.py:206-226)
    gsp = torch.rand(2, t30, 32, generator=g)
    nwp = torch.randn(2, 10, t60, kw["nwp_image_size_pixels"], kw["nwp_image_size_pixels"], generator=g)
    row = torch.randint(0, 940, (2, 128), generator=g)
    gsp_id = torch.randint(1, 339, (2, 32), generator=g)
    batch = {"satellite": {"data": sat}, "pv": {"pv_yield": pv, "pv_system_row_number": row},
             "gsp": {"gsp_yield": gsp, "gsp_id": gsp_id}, "nwp": {"data": nwp}}
    for k, v in dict(sat=sat, pv=pv, gsp=gsp, nwp=nwp, row=row, gsp_id=gsp_id).items():
        out[f"{tag}/{k}"] = v.numpy()
    for k, v in model.state_dict().items():
        out[f"{tag}/init/{k}"] = v.numpy().copy()
    out[f"{tag}/y_hat"] = model(batch).detach().numpy().copy()
    opt = model.configure_optimizers()
    losses = []
    for step in range(3):
        opt.zero_grad()
        loss = model.training_step(batch, 0)
        loss.backward()
        if step == 0:
            for k, p in model.named_parameters():
                out[f"{tag}/grad/{k}"] = checksum(p.grad)
        opt.step()
        losses.append(float(loss.detach()))
    for k, p in model.named_parameters():
        out[f"{tag}/step3/{k}"] = checksum(p)
    out[f"{tag}/losses"] = np.array(losses)
    out[f"{tag}/attrs"] = np.array([model.cnn_output_size, model.nwp_cnn_output_size, model.forecast_len,
                                    model.fc3.in_features])


def main():
    install_stubs()
    sys.path.insert(0, REF)
    from predict_pv_yield.models.conv3d.model import Model  # the reference's own source
    out = {}
    run_case(Model, KW, "sat_only", out)
    run_case(Model, KW_PV, "pv_nwp", out)
    from predict_pv_yield.models.conv3d.model_sat_nwp import Model as SatNwpModel  # the reference's own source
    run_case_sat_nwp(SatNwpModel, SN, "sat_nwp", out)
    run_case_sat_nwp(SatNwpModel, SN_PV, "sat_nwp_pv", out)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB;", len(out), "arrays")


if __name__ == "__main__":
    main()
