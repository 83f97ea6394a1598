"""Persistence baseline — mirror of predict_pv_yield/models/baseline/last_value.py:12-45
(the model `run.py experiment=example_simple` drives).  Pure indexing: no kernels of its own."""
import logging

from ...data.batch import BatchML
from ..base_model import BaseModel

_LOG = logging.getLogger("predict_pv_yield_amd")


class Model(BaseModel):
    name = "last_value"

    def __init__(self, forecast_minutes: int = 12, history_minutes: int = 6, output_variable="pv_yield"):
        self.forecast_minutes = forecast_minutes
        self.history_minutes = history_minutes
        self.output_variable = output_variable
        super().__init__()

    def forward(self, x: BatchML):
        if type(x) == dict:
            x = BatchML(**x)
        # Shape: batch_size, seq_length, n_sites
        y = x.gsp.gsp_yield if self.output_variable == "gsp_yield" else x.pv.pv_yield
        # last non-forecast value of the first site, copied forward (last_value.py:38-43)
        y_hat = y[:, -self.forecast_len - 1, 0]
        return y_hat.unsqueeze(1).repeat(1, self.forecast_len)

    def configure_optimizers(self):
        return _NoOpOptimizer()


class _NoOpOptimizer:
    """The baseline has no parameters; Lightning never steps it (validate / test only)."""
    param_groups = []

    def zero_grad(self, set_to_none=True): ...
    def step(self): ...
    def state_dict(self): return {}
    def load_state_dict(self, s): ...
