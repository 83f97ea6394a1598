"""Super-batch → training-example pipeline of notebooks/13_3d_conv_with_optical_flow_predictions.ipynb, on the device.

  load_super_batch         SatelliteLoader.load_super_batch (13_…ipynb:446-474): Farnebäck fields on the raw 10-bit
                           counts, normalise with the HRV mean/std, all (n-1)n/2 flow predictions (NaN border)
  pick_example_indices     the random choices of super_batch_to_example (:675-728): history start, target
  sample_squares           (:604-652) random 128² crop of history + prediction with the centred 64² target crop;
                           crops containing NaN are rejected
  super_batch_to_example   the two retry loops
  collate                  default_collate of B examples -> the batch dict LitAutoEncoder.forward reads

The images never leave the MI355X: flow, normalisation and warps are the HIP kernels of optical_flow.py, crops are
views of device tensors; only the random index choices (numpy Generator, as in the notebook) run on the host.
One deliberate difference: the notebook's `_crop` overwrites the example dict while it iterates, so a retry after a
half-finished crop samples from an already-cropped 128² image (and raises ValueError from rng.integers(0, 0));
here every retry samples from the full-extent images, which is what the surrounding comments describe.
"""
from typing import Dict, Optional

import numpy as np
import torch

from .. import optical_flow as of
from ..models.conv3d.flow_autoencoder import (FORECAST_HORIZON, HISTORICAL_SAT_IMAGES, OPTICAL_FLOW_PREDICTIONS,
                                               SECONDS_PER_TIMESTEP, TARGET_SAT_IMAGE, normalise_forecast_horizon)

SAT_IMAGES = "SAT_IMAGES"
OPTICAL_FLOW_FIELDS = "OPTICAL_FLOW_FIELDS"
PREDICTION_INDEX = "PREDICTION_INDEX"   # [P, 2] (t0 index, index of the frame the forecast is about): the DataFrame index

SAT_IMAGE_MEAN = np.float32(93.23458)   # 13_…ipynb:345-346
SAT_IMAGE_STD = np.float32(115.34247)


class ImageHasNansError(Exception):
    pass


def load_super_batch(raw_counts: torch.Tensor, include_optical_flow: bool = True) -> Dict[str, torch.Tensor]:
    """raw_counts: [T, H, W] 10-bit counts (int16 or float32) of consecutive 5-minute HRV images, on the device."""
    if not raw_counts.is_cuda:
        raise RuntimeError("load_super_batch: the satellite stack must be on the MI355X (there is no CPU fallback)")
    from .. import hip_ops as K
    counts = raw_counts if raw_counts.dtype in (torch.int16, torch.float32) else raw_counts.float()
    mean = torch.tensor([SAT_IMAGE_MEAN], device=counts.device)
    std = torch.tensor([SAT_IMAGE_STD], device=counts.device)
    super_batch = {}
    if include_optical_flow:
        super_batch[OPTICAL_FLOW_FIELDS] = of.compute_optical_flow(counts)          # flow on the RAW counts
    sat = K.normalise(counts.contiguous(), mean, std, inner=counts.numel())        # then normalise (:460-464)
    super_batch[SAT_IMAGES] = sat
    if include_optical_flow:
        preds, index = of.compute_optical_flow_predictions(sat, super_batch[OPTICAL_FLOW_FIELDS])
        super_batch[OPTICAL_FLOW_PREDICTIONS] = preds
        super_batch[PREDICTION_INDEX] = index
    return super_batch


def pick_example_indices(rng: np.random.Generator, n_sat_images: int, prediction_index: np.ndarray,
                         n_historical_images: int = 4, history_stride: int = 3):
    """Host half of super_batch_to_example: (hist_start_idx, hist_end_idx, t0_idx, prediction row, target idx)."""
    total_hist_length = n_historical_images * history_stride
    max_hist_start_idx = n_sat_images - total_hist_length - 1   # -1: predict at least one timestep ahead
    hist_start_idx = int(rng.integers(low=0, high=max_hist_start_idx))
    hist_end_idx = hist_start_idx + total_hist_length
    t0_idx = hist_end_idx - 1
    rows = np.flatnonzero(prediction_index[:, 0] == t0_idx)     # all predictions made at t0, sorted by target
    row = int(rng.choice(rows))
    return hist_start_idx, hist_end_idx, t0_idx, row, int(prediction_index[row, 1])


def sample_squares(example: Dict[str, torch.Tensor], rng: np.random.Generator, n_pixels_per_side_large: int = 128,
                   n_pixels_per_side_small: int = 64) -> Dict[str, torch.Tensor]:
    height, width = example[OPTICAL_FLOW_PREDICTIONS].shape[-2:]
    large_max_x = width - n_pixels_per_side_large
    large_max_y = height - n_pixels_per_side_large
    border = (n_pixels_per_side_large - n_pixels_per_side_small) // 2
    large_top = int(rng.integers(low=0, high=large_max_y))
    large_left = int(rng.integers(low=0, high=large_max_x))
    out = dict(example)

    def crop(names, top, bottom, left, right):
        for name in names:
            image = example[name][..., top:bottom, left:right]
            if bool(torch.isnan(image).any()):
                raise ImageHasNansError(f"{name} has NaNs!")
            out[name] = image

    # predictions first: they are the most likely to have NaNs (smeared edges)
    crop((OPTICAL_FLOW_PREDICTIONS, HISTORICAL_SAT_IMAGES), large_top, large_top + n_pixels_per_side_large,
         large_left, large_left + n_pixels_per_side_large)
    crop((TARGET_SAT_IMAGE,), large_top + border, large_top + n_pixels_per_side_large - border,
         large_left + border, large_left + n_pixels_per_side_large - border)
    return out


def super_batch_to_example(super_batch: Dict[str, torch.Tensor], rng: Optional[np.random.Generator] = None,
                           n_historical_images: int = 4, history_stride: int = 3, max_retries: int = 128,
                           n_pixels_per_side_large: int = 128, n_pixels_per_side_small: int = 64):
    rng = rng if rng is not None else np.random.default_rng()
    sat = super_batch[SAT_IMAGES]
    index = super_batch[PREDICTION_INDEX]
    index = index.cpu().numpy() if isinstance(index, torch.Tensor) else np.asarray(index)
    for _ in range(max_retries):
        start, end, t0_idx, row, target_idx = pick_example_indices(rng, len(sat), index, n_historical_images,
                                                                   history_stride)
        horizon_s = (target_idx - t0_idx) * SECONDS_PER_TIMESTEP
        example = {
            TARGET_SAT_IMAGE: sat[target_idx],
            FORECAST_HORIZON: torch.tensor(normalise_forecast_horizon(horizon_s), device=sat.device),
            HISTORICAL_SAT_IMAGES: sat[start:end:history_stride],
            OPTICAL_FLOW_PREDICTIONS: super_batch[OPTICAL_FLOW_PREDICTIONS][row],
        }
        for _ in range(max_retries):
            try:
                return sample_squares(example, rng, n_pixels_per_side_large, n_pixels_per_side_small)
            except ImageHasNansError:
                pass
    raise ImageHasNansError(f"Cropped images still have NaNs, even after {max_retries ** 2} retries!")


def collate(examples) -> Dict[str, torch.Tensor]:
    return {k: torch.stack([e[k] for e in examples]).contiguous() for k in examples[0]}
