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


RESULTS_COLUMNS = ("t0_datetime_utc", "target_datetime_utc", "gsp_id", "actual_gsp_pv_outturn_mw",
                   "forecast_gsp_pv_outturn_mw")


def _weighted_losses_weights(n: int, device) -> torch.Tensor:
    """WeightedLosses(forecast_length=n): w_i = exp(-ln2 i) normalised to mean 1, f32 (base_model.py:76)."""
    w = torch.tensor([float(np.exp(-np.log(2.0) * i)) for i in range(n)], dtype=torch.float32, device=device)
    return w / w.sum() * n


class BaseModel(LightningModule):
    # default batch_size (base_model.py:30) -- used to slice the target, independent of the loader
    batch_size = 32
    results_file_name = "results_epoch"

    def __init__(self):
        super().__init__()
        self.results_dfs = []          # per instance (a class-level list would be shared by every model object)
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

    def _is_parameter_free(self) -> bool:
        return next(self.parameters(), None) is None

    def _losses(self, y_hat: torch.Tensor, y: torch.Tensor, per_horizon: bool = False):
        """(mse, nmae, mse_exp, mae_exp) [, mse per forecast step, mae per forecast step]; only nmae carries a gradient.
        On the MI355X everything comes from ONE launch of pv_forecast_losses_f32.  A module WITHOUT parameters (the
        persistence baseline: pure indexing, nothing to train, no kernels of its own) may also be evaluated on the host --
        that is the reference's CPU plumbing run (BASELINE config 1, tests/models/baseline/*); every trainable model
        refuses the CPU."""
        if y_hat.is_cuda:
            from ..functional import forecast_losses, forecast_losses_with_horizons
            if per_horizon:
                return forecast_losses_with_horizons(y_hat.float(), y.float())
            return forecast_losses(y_hat.float(), y.float())
        if not self._is_parameter_free():
            raise RuntimeError("predict_pv_yield_amd: losses run on the MI355X only (no CPU path); model is on "
                               f"{y_hat.device}")
        d = y_hat.float() - y.float()
        w = _weighted_losses_weights(d.shape[1], d.device)
        four = ((d * d).mean(), d.abs().mean(), (w * d * d).mean(), (w * d.abs()).mean())
        return (four, (d * d).mean(dim=0), d.abs().mean(dim=0)) if per_horizon else four

    def _training_or_validation_step(self, batch, tag: str, return_model_outputs: bool = False):
        if type(batch) == dict:
            batch = BatchML(**batch)
        y_hat = self(batch)
        y = self._target(batch)
        evaluating = tag != "Train"
        if evaluating:
            (mse_loss, nmae_loss, mse_exp, mae_exp), mse_h, mae_h = self._losses(y_hat, y, per_horizon=True)
        else:
            mse_loss, nmae_loss, mse_exp, mae_exp = self._losses(y_hat, y)
        self.log_dict({f"MSE/{tag}": mse_loss, f"NMAE/{tag}": nmae_loss, f"MSE_EXP/{tag}": mse_exp,
                       f"MAE_EXP/{tag}": mae_exp}, on_step=True, on_epoch=True, sync_dist=True)
        if evaluating:
            # per-horizon metrics exactly as base_model.py:121-141 builds them: both dicts use the key
            # `MSE_forecast_horizon_{i}/{tag}` for i in range(forecast_len_30), and `{**metrics_mse, **metrics_mae}`
            # keeps the LATER entry -- what is logged under the MSE name is the MAE of that horizon
            metrics_mse = {f"MSE_forecast_horizon_{i}/{tag}": mse_h[i] for i in range(self.forecast_len_30)}
            metrics_mae = {f"MSE_forecast_horizon_{i}/{tag}": mae_h[i] for i in range(self.forecast_len_30)}
            merged = {**metrics_mse, **metrics_mae}
            if merged:
                self.log_dict(merged, on_step=True, on_epoch=True, sync_dist=True)
        if return_model_outputs:
            return nmae_loss, y_hat
        return nmae_loss

    def training_step(self, batch, batch_idx):
        return self._training_or_validation_step(batch, tag="Train")

    def validation_step(self, batch, batch_idx):
        if type(batch) == dict:
            batch = BatchML(**batch)
        nmae_loss, model_output = self._training_or_validation_step(batch, tag="Validation", return_model_outputs=True)
        # validation results table (base_model.py:223-239; nowcasting_utils.make_validation_results): one row per
        # (example, 30-minute forecast step), MW = normalised value * capacity
        gsp = getattr(batch, "gsp", None)
        n30 = self.forecast_len_30
        if gsp is not None and getattr(gsp, "gsp_capacity", None) is not None and n30 > 0 and model_output.shape[1] >= n30:
            capacity = gsp.gsp_capacity[:, -n30:, 0].cpu().numpy()
            predictions = model_output[:, -n30:].detach().float().cpu().numpy() * capacity
            truths = gsp.gsp_yield[:, -n30:, 0].cpu().numpy() * capacity
            t0 = batch.metadata.t0_datetime_utc if batch.metadata is not None else None
            if isinstance(t0, torch.Tensor):
                t0 = t0.cpu().numpy()
            elif t0 is not None:
                t0 = np.asarray(t0)
            gsp_ids = gsp.gsp_id[:, 0].cpu().numpy()
            rows = []
            for b in range(predictions.shape[0]):
                t0_b = np.datetime64(int(t0[b]), "ns") if t0 is not None else np.datetime64("NaT")
                for i in range(n30):
                    rows.append((t0_b, t0_b + np.timedelta64(30 * (i + 1), "m"), int(gsp_ids[b]),
                                 float(truths[b, i]), float(predictions[b, i])))
            if batch_idx == 0:
                self.results_dfs = []
            self.results_dfs.append(rows)
        return nmae_loss

    def validation_epoch_end(self, outputs):
        """save_validation_results_to_logger (base_model.py:243-250): `{results_file_name}_{current_epoch}.csv`, relative
        to the working directory like the reference (tests set results_file_name to an absolute prefix)."""
        if not self.results_dfs:
            return
        path = f"{self.results_file_name}_{self.current_epoch}.csv"
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        with open(path, "w") as f:
            f.write(",".join(RESULTS_COLUMNS) + "\n")
            for rows in self.results_dfs:
                for r in rows:
                    f.write(",".join(str(v) for v in r) + "\n")
        self.last_results_path = path

    def test_step(self, batch, batch_idx):
        self._training_or_validation_step(batch, tag="Test")

    def configure_optimizers(self):
        """torch.optim.Adam(self.parameters(), lr=0.0005) (base_model.py:255-257), stepped by pv_adam_step_f32."""
        from ..optim import HipAdam
        mark = getattr(self, "_mark_fc1_layout", None)
        if mark is not None:
            mark()        # (parameter-level layout marks do not survive copy.deepcopy: models/conv3d/_fc1_layout.py)
        return HipAdam(self.parameters(), lr=0.0005)
