"""Generates tests/golden/conv3d_nwp_small.npz by EXECUTING THE REFERENCE'S OWN MODULE SOURCE
(/root/reference/predict_pv_yield/models/conv3d/model_nwp.py + base_model.py) on the CPU, under the same import stubs
as make_conv3d_golden.py (they replace packages that are absent from this image and contribute no arithmetic).

Run here (the reference tree does not travel to the GPU box):   python tests/golden/make_conv3d_nwp_golden.py

The fixture holds inputs, initial parameters, the forward output, gradient / parameter checksums and the losses of
three Adam steps: data only, no reference source text.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_conv3d_golden import REF, checksum, install_stubs  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conv3d_nwp_small.npz")

NWP = dict(forecast_minutes=120, history_minutes=30, number_of_conv3d_layers=4, conv3d_channels=32, nwp_image_size_pixels=12,
           number_nwp_channels=10, fc1_output_features=16, fc2_output_features=16, fc3_output_features=16,
           output_variable="gsp_yield")
NWP_1CH = dict(NWP, number_nwp_channels=1, number_of_conv3d_layers=2, nwp_image_size_pixels=6, output_variable="pv_yield",
               forecast_minutes=60, embedding_dem=0, include_pv_yield_history=False)


def run_case(Model, kw, tag, out):
    torch.manual_seed(518)
    model = Model(**kw)
    t5 = kw["history_minutes"] // 5 + kw["forecast_minutes"] // 5 + 1
    t30 = kw["history_minutes"] // 30 + kw["forecast_minutes"] // 30 + 1
    t60 = int(np.ceil(kw["history_minutes"] / 60)) + kw["forecast_minutes"] // 60 + 1
    g = torch.Generator().manual_seed(9876)
    pv = torch.rand(3, t5, 128, generator=g)
    gsp = torch.rand(3, t30, 32, generator=g)
    nwp = torch.randn(3, kw["number_nwp_channels"], t60, kw["nwp_image_size_pixels"], kw["nwp_image_size_pixels"], generator=g)
    batch = {"pv": {"pv_yield": pv}, "gsp": {"gsp_yield": gsp}, "nwp": {"data": nwp}}
    for k, v in dict(pv=pv, gsp=gsp, nwp=nwp).items():
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
                if p.grad is not None:
                    out[f"{tag}/grad/{k}"] = checksum(p.grad)
        opt.step()
        losses.append(float(loss.detach()))
    for k, p in model.named_parameters():
        out[f"{tag}/step3/{k}"] = checksum(p)
    out[f"{tag}/losses"] = np.array(losses)
    out[f"{tag}/attrs"] = np.array([model.nwp_cnn_output_size, model.forecast_len, model.fc3.in_features])


def main():
    install_stubs()
    sys.path.insert(0, REF)
    from predict_pv_yield.models.conv3d.model_nwp import Model  # the reference's own source
    out = {}
    run_case(Model, NWP, "nwp", out)
    run_case(Model, NWP_1CH, "nwp_1ch", out)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB;", len(out), "arrays")


if __name__ == "__main__":
    main()
