"""BaseModel — host-side mirror of predict_pv_yield/models/base_model.py:27-257.

Same attributes (history_len_5/30/60, forecast_len_5/30/60, forecast_len, number_of_samples_per_batch,
batch_size = 32, results_file_name), the same step / loss / metric-name contract
(`MSE/{tag}`, `NMAE/{tag}`, `MSE_EXP/{tag}`, `MAE_EXP/{tag}`, `MSE_forecast_horizon_{i}/{tag}`) and the
same optimiser (Adam, lr 5e-4).  The arithmetic is the HIP path:
  losses       pv_forecast_losses_f32 (one launch; nmae carries the gradient, base_model.py:99,146)
  optimiser    HipAdam -> pv_adam_step_f32 (torch.optim.Adam order of operations)
Validation plotting / Neptune upload (base_model.py:167-220) is outside the hot path
and not reproduced; the validation results table (base_model.py:223-250) is kept as a CSV.
"""
import logging
import os

import numpy as np
import torch

from ..data.batch import BatchML
from ..lightning import LightningModule

logger = logging.getLogger(__name__)

default_output_variable = "pv_yield"


class BaseModel(LightningModule):
    # default batch_size (base_model.py:30) -- used to slice the target, independent of the loader
    batch_size = 32
    results_file_name = "results_epoch"
    results_dfs = []

    def __init__(self):
        super().__init__()
        self.history_len_5 = self.history_minutes // 5
        self.forecast_len_5 = self.forecast_minutes // 5
        self.history_len_30 = self.history_minutes // 30
        self.forecast_len_30 = self.forecast_minutes // 30
        self.history_len_60 = int(np.ceil(self.history_minutes / 60))
        self.forecast_len_60 = self.forecast_minutes // 60
        if not hasattr(self, "output_variable"):
            self.output_variable = default_output_variable
        if self.output_variable == "pv_yield":
            self.forecast_len = self.forecast_len_5
            self.history_len = self.history_len_5
            self.number_of_samples_per_batch = 128
        else:
            self.forecast_len = self.forecast_len_30
            self.history_len = self.history_len_30
            self.number_of_samples_per_batch = 32
        self.number_of_pv_samples_per_batch = 128

    # ------------------------------------------------------------------------------------------
    def _target(self, batch: BatchML) -> torch.Tensor:
        """y = yield[0:batch_size, -forecast_len:, 0] (base_model.py:91-95): a strided view, no copy."""
        y = batch.gsp.gsp_yield if self.output_variable == "gsp_yield" else batch.pv.pv_yield
        return y[0 : self.batch_size, -self.forecast_len :, 0]

    def _losses(self, y_hat: torch.Tensor, y: torch.Tensor):
        """(mse, nmae, mse_exp, mae_exp); only nmae carries a gradient."""
        if y_hat.is_cuda:
            from ..functional import forecast_losses
            y = y.float()
            return forecast_losses(y_hat.float(), y)
        raise RuntimeError("predict_pv_yield_amd: losses run on the MI355X only (no CPU path); model is on "
                           f"{y_hat.device}")

    def _training_or_validation_step(self, batch, tag: str, return_model_outputs: bool = False):
        if type(batch) == dict:
            batch = BatchML(**batch)
        y_hat = self(batch)
        y = self._target(batch)
        mse_loss, nmae_loss, mse_exp, mae_exp = self._losses(y_hat, y)
        self.log_dict({f"MSE/{tag}": mse_loss, f"NMAE/{tag}": nmae_loss, f"MSE_EXP/{tag}": mse_exp,
                       f"MAE_EXP/{tag}": mae_exp}, on_step=True, on_epoch=True, sync_dist=True)
        if tag != "Train":
            # per-horizon metrics (base_model.py:121-141); the reference's MAE dict reuses the MSE key
            # names and overwrites them -- the logged value under MSE_forecast_horizon_i is the MAE
            d = (y_hat.detach().float() - y.float())
            mae_h = d.abs().mean(dim=0)
            metrics = {f"MSE_forecast_horizon_{i}/{tag}": mae_h[i] for i in range(min(self.forecast_len_30, mae_h.numel()))}
            if metrics:
                self.log_dict(metrics, on_step=True, on_epoch=True, sync_dist=True)
        if return_model_outputs:
            return nmae_loss, y_hat
        return nmae_loss

    def training_step(self, batch, batch_idx):
        return self._training_or_validation_step(batch, tag="Train")

    def validation_step(self, batch, batch_idx):
        if type(batch) == dict:
            batch = BatchML(**batch)
        nmae_loss, model_output = self._training_or_validation_step(batch, tag="Validation", return_model_outputs=True)
        # validation results table (base_model.py:223-239): MW = normalised value * capacity
        gsp = getattr(batch, "gsp", None)
        if gsp is not None and getattr(gsp, "gsp_capacity", None) is not None and self.forecast_len_30 > 0 \
                and model_output.shape[1] >= self.forecast_len_30:
            capacity = gsp.gsp_capacity[:, -self.forecast_len_30 :, 0].cpu().numpy()
            predictions = model_output[:, -self.forecast_len_30 :].detach().float().cpu().numpy() * capacity
            truths = gsp.gsp_yield[:, -self.forecast_len_30 :, 0].cpu().numpy() * capacity
            t0 = batch.metadata.t0_datetime_utc if batch.metadata is not None else None
            if isinstance(t0, torch.Tensor):
                t0 = t0.cpu().numpy()
            elif t0 is not None:
                t0 = np.asarray(t0)
            gsp_ids = gsp.gsp_id[:, 0].cpu().numpy()
            rows = []
            for b in range(predictions.shape[0]):
                for i in range(self.forecast_len_30):
                    t0_b = np.datetime64(int(t0[b]), "ns") if t0 is not None else np.datetime64("NaT")
                    rows.append((t0_b, t0_b + np.timedelta64(30 * (i + 1), "m"), int(gsp_ids[b]),
                                 float(truths[b, i]), float(predictions[b, i])))
            if batch_idx == 0:
                self.results_dfs = []
            self.results_dfs.append(rows)
        return nmae_loss

    def validation_epoch_end(self, outputs):
        """save_validation_results_to_logger (base_model.py:243-250): results_epoch_{epoch}.csv."""
        if not self.results_dfs:
            return
        lg = self.logger
        out_dir = getattr(lg, "log_dir", None) if lg is not None and not isinstance(lg, list) else None
        if out_dir is None and lg is not None and hasattr(lg, "__getitem__"):
            out_dir = getattr(lg[0], "log_dir", None)
        out_dir = out_dir or "."
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, f"{self.results_file_name}_{self.current_epoch}.csv")
        with open(path, "w") as f:
            f.write("t0_datetime_utc,target_datetime_utc,gsp_id,actual_gsp_pv_outturn_mw,forecast_gsp_pv_outturn_mw\n")
            for rows in self.results_dfs:
                for r in rows:
                    f.write(",".join(str(v) for v in r) + "\n")
        self.last_results_path = path

    def test_step(self, batch, batch_idx):
        self._training_or_validation_step(batch, tag="Test")

    def configure_optimizers(self):
        """torch.optim.Adam(self.parameters(), lr=0.0005) (base_model.py:255-257), stepped by pv_adam_step_f32."""
        from ..optim import HipAdam
        return HipAdam(self.parameters(), lr=0.0005)
