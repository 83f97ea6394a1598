"""Generates tests/golden/flow_sampler.npz by EXECUTING THE NOTEBOOK'S OWN CELLS (sample_squares, normalise_forecast_horizon,
super_batch_to_example: /root/reference/notebooks/13_3d_conv_with_optical_flow_predictions.ipynb:604-728) on a synthetic
super batch, with the rng the test replays.  xarray is not installed here: `_TimeIndexedImages` stands in for the DataArray
the cell indexes (len / .time[i].values / .loc[t].values / .isel(time=slice).values -- plumbing only, no arithmetic); the
prediction table is a real pandas DataFrame with the notebook's (t0, target) MultiIndex and IMAGE / FORECAST_HORIZON columns.

Run here (the reference tree does not travel to the GPU box):   python tests/golden/make_flow_sampler_golden.py
The fixture holds, per drawn example, the normalised horizon and checksums / corner samples of the three crops."""
import json
import os
import sys
from typing import Dict, Optional  # noqa: F401  (names the notebook cells use)

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = "/root/reference"
NOTEBOOK = os.path.join(REF, "notebooks", "13_3d_conv_with_optical_flow_predictions.ipynb")
OUT = os.path.join(HERE, "flow_sampler.npz")
N_DRAWS, SEED, RNG_SEED = 25, 0, 42


def fake_super_batch(t=20, h=150, w=170, seed=SEED, nan_border=9):
    """The generator tests/test_flow_examples.py uses: normalised images + (n-1)n/2 predictions whose NaN frame grows."""
    rng = np.random.default_rng(seed)
    sat = rng.standard_normal((t, h, w)).astype(np.float32)
    preds, index = [], []
    for flow_i in range(t - 1):
        for step in range(1, t - flow_i):
            p = rng.standard_normal((h, w)).astype(np.float32)
            m = min(nan_border + step, 40)
            p[:m] = np.nan
            p[:, -m:] = np.nan
            preds.append(p)
            index.append((flow_i, flow_i + step))
    return sat, np.stack(preds), np.array(index, dtype=np.int64)


class _Values:
    def __init__(self, v):
        self.values = v


class _TimeIndexedImages:
    """What the cell needs of xr.DataArray(dims=(time, y, x))."""

    def __init__(self, data, times):
        self.data, self._times = data, list(times)
        self.time = [_Values(np.datetime64(t)) for t in times]
        outer = self

        class _Loc:
            def __getitem__(self, t):
                return _Values(outer.data[outer._times.index(pd.Timestamp(t))])
        self.loc = _Loc()

    def __len__(self):
        return len(self.data)

    def isel(self, time):
        return _Values(self.data[time])


def load_cells():
    nb = json.load(open(NOTEBOOK))
    cells = ["".join(c["source"]) for c in nb["cells"] if c["cell_type"] == "code"]
    ns = dict(np=np, pd=pd, Dict=Dict, Optional=Optional, MINUTES_PER_TIMESTEP=5, SAT_IMAGES="SAT_IMAGES",
              FORECAST_HORIZON="FORECAST_HORIZON", HISTORICAL_SAT_IMAGES="HISTORICAL_SAT_IMAGES",
              OPTICAL_FLOW_PREDICTIONS="OPTICAL_FLOW_PREDICTIONS", TARGET_SAT_IMAGE="TARGET_SAT_IMAGE")
    for marker in ("def sample_squares", "def normalise_forecast_horizon", "def super_batch_to_example"):
        (src,) = [c for c in cells if marker in c]
        exec(compile(src, NOTEBOOK, "exec"), ns)
    return ns


def main():
    ns = load_cells()
    sat, preds, index = fake_super_batch()
    times = pd.date_range("2021-06-01 10:00", periods=len(sat), freq="5min")
    rows = pd.MultiIndex.from_arrays([times[index[:, 0]], times[index[:, 1]]])
    table = pd.DataFrame({"IMAGE": list(preds), "FORECAST_HORIZON": times[index[:, 1]] - times[index[:, 0]]}, index=rows)
    super_batch = {"SAT_IMAGES": _TimeIndexedImages(sat, times), "OPTICAL_FLOW_PREDICTIONS": table}
    rng = np.random.default_rng(RNG_SEED)
    out = {k: [] for k in ("horizon", "hist_sum", "hist_corner", "pred_sum", "pred_corner", "target_sum", "target_corner")}
    for _ in range(N_DRAWS):
        ex = ns["super_batch_to_example"](super_batch, rng=rng)
        h, p, t = ex["HISTORICAL_SAT_IMAGES"], np.asarray(ex["OPTICAL_FLOW_PREDICTIONS"]), ex["TARGET_SAT_IMAGE"]
        assert h.shape == (4, 128, 128) and p.shape == (128, 128) and t.shape == (64, 64)
        out["horizon"].append(np.float32(ex["FORECAST_HORIZON"]))
        for name, a in (("hist", h), ("pred", p), ("target", t)):
            out[f"{name}_sum"].append(a.astype(np.float64).sum())
            out[f"{name}_corner"].append(np.float32(a.reshape(-1)[0]))
    out = {k: np.array(v) for k, v in out.items()}
    out["rng_probe"] = np.array(rng.integers(0, 1 << 30))        # the stream position after the draws
    out["params"] = np.array([N_DRAWS, SEED, RNG_SEED])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
