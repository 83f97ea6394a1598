"""Generates tests/golden/flow_models_small.npz by EXECUTING THE REFERENCE'S OWN SOURCE on the CPU:

  * `LitAutoEncoder` — the code cell of /root/reference/notebooks/13_3d_conv_with_optical_flow_predictions.ipynb that
    defines CHANNELS / KERNEL / PADDING and the class, exec'd as written (pytorch_lightning.LightningModule is replaced
    by torch.nn.Module + a recording log_dict, as in make_conv3d_golden.py; the dict-key constants are the notebook's);
  * `Conv3dMaxPool` — imported from /root/reference/predict_pv_yield/models/perceiver/perceiver_conv3d_nwp_sat.py
    (perceiver_pytorch / nowcasting_* are import stubs; none contributes arithmetic to this class).

Run here (the reference tree does not travel to the GPU box):   python tests/golden/make_flow_model_golden.py
Inputs are reduced (32x32 images instead of 128x128) so the fixture stays small; the vectors are data only.
"""
import json
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_conv3d_golden import REF, _stub, checksum, install_stubs  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "flow_models_small.npz")
NOTEBOOK = os.path.join(REF, "notebooks", "13_3d_conv_with_optical_flow_predictions.ipynb")


def load_notebook_class():
    install_stubs()
    import pytorch_lightning as pl
    import torch.nn.functional as F
    from torch import nn
    nb = json.load(open(NOTEBOOK))
    cells = ["".join(c["source"]) for c in nb["cells"] if c["cell_type"] == "code"]
    (src,) = [c for c in cells if "class LitAutoEncoder" in c]
    ns = dict(torch=torch, nn=nn, F=F, pl=pl, FORECAST_HORIZON="FORECAST_HORIZON",
              HISTORICAL_SAT_IMAGES="HISTORICAL_SAT_IMAGES", OPTICAL_FLOW_PREDICTIONS="OPTICAL_FLOW_PREDICTIONS",
              TARGET_SAT_IMAGE="TARGET_SAT_IMAGE")
    exec(compile(src, NOTEBOOK, "exec"), ns)
    return ns["LitAutoEncoder"]


def autoencoder_case(out, size=32, batch=3):
    LitAutoEncoder = load_notebook_class()
    torch.manual_seed(42)
    model = LitAutoEncoder()
    g = torch.Generator().manual_seed(99)
    batch_d = {"HISTORICAL_SAT_IMAGES": torch.randn(batch, 4, size, size, generator=g),
               "OPTICAL_FLOW_PREDICTIONS": torch.randn(batch, size, size, generator=g),
               "FORECAST_HORIZON": torch.randn(batch, generator=g),
               "TARGET_SAT_IMAGE": torch.randn(batch, size // 2, size // 2, generator=g)}
    for k, v in batch_d.items():
        out[f"ae/{k}"] = v.numpy()
    for k, v in model.state_dict().items():
        out[f"ae/init/{k}"] = v.numpy().copy()
    out["ae/y_hat"] = model(batch_d).detach().numpy().copy()
    opt = model.configure_optimizers()
    losses = []
    for step in range(3):
        opt.zero_grad()
        loss = model.training_step(batch_d, 0)
        loss.backward()
        if step == 0:
            for k, p in model.named_parameters():
                out[f"ae/grad/{k}"] = p.grad.numpy().copy()
        opt.step()
        losses.append(float(loss.detach()))
    for k, p in model.named_parameters():
        out[f"ae/step3/{k}"] = p.detach().numpy().copy()
    out["ae/losses"] = np.array(losses)


def maxpool_case(out):
    install_stubs()
    _stub("perceiver_pytorch", Perceiver=object)
    _stub("nowcasting_dataset.consts", NWP_VARIABLE_NAMES=tuple("abcdefghijkl"), SAT_VARIABLE_NAMES=tuple("abcdefghijkl"))
    sys.path.insert(0, REF)
    from predict_pv_yield.models.perceiver.perceiver_conv3d_nwp_sat import Conv3dMaxPool  # the reference's own source
    torch.manual_seed(7)
    block = Conv3dMaxPool(out_channels=8, in_channels=3)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 3, 5, 13, 16, generator=g).requires_grad_(True)       # odd height: floor in the output extent
    y = block(x)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    out["mp/x"], out["mp/dy"], out["mp/y"], out["mp/dx"] = x.detach().numpy(), dy.numpy(), y.detach().numpy(), x.grad.numpy()
    for k, v in block.state_dict().items():
        out[f"mp/init/{k}"] = v.numpy().copy()
    for k, p in block.named_parameters():
        out[f"mp/grad/{k}"] = p.grad.numpy().copy()


def main():
    out = {}
    autoencoder_case(out)
    maxpool_case(out)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB;", len(out), "arrays")


if __name__ == "__main__":
    main()
