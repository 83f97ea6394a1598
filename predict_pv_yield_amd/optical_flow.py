"""Optical-flow advection of satellite tiles on the MI355X — the notebook helpers of the reference,
batched and device-resident.

Mirrors (same names, argument meaning and error behaviour):
  convert_10bpp_to_uint8            notebooks/13_3d_conv_with_optical_flow_predictions.ipynb:112-119
  calcOpticalFlowFarneback          cv.calcOpticalFlowFarneback as called at 13_...ipynb:133-135
  compute_optical_flow              13_...ipynb:175-240   (process pool -> one batched launch sequence)
  weighted_average                  notebooks/optical_flow_1.ipynb:293-294
  remap_image                       13_...ipynb:259-281 (BORDER_CONSTANT, NaN) / optical_flow_1.ipynb:415-430
  compute_optical_flow_predictions  13_...ipynb:284-333   ((n-1)n/2 predictions from n frames)
  structural_similarity             skimage.metrics.structural_similarity as called at optical_flow_1.ipynb cells 31, 35, 38
  compute_opt_flow_and_score, search_farneback_params   optical_flow_1.ipynb cells 38-42 (the one-at-a-time parameter search)
and the join the reference leaves as a TODO (`# TODO: Use optical flow, not actual sat images of the
future!`, predict_pv_yield/models/perceiver/perceiver.py:118):
  advect_future_frames / replace_future_frames_with_flow   
Tensors stay on the GPU end to end; NumPy inputs are accepted for the cv2-style single-image calls and
are copied to the device (the compute never runs on the host).
"""
from typing import Optional, Sequence, Union

import numpy as np
import torch

from . import hip_ops as K
from ._lib import PV_BORDER_CONSTANT, PV_BORDER_REPLICATE, PV_OPTFLOW_FARNEBACK_GAUSSIAN

BORDER_CONSTANT = PV_BORDER_CONSTANT
BORDER_REPLICATE = PV_BORDER_REPLICATE
OPTFLOW_FARNEBACK_GAUSSIAN = PV_OPTFLOW_FARNEBACK_GAUSSIAN
INTER_LINEAR = 1

# Farnebäck arguments used everywhere in the reference (13_...ipynb:133-135)
REFERENCE_FARNEBACK_KWARGS = dict(pyr_scale=0.5, levels=2, winsize=40, iterations=3, poly_n=5, poly_sigma=0.7,
                                  flags=OPTFLOW_FARNEBACK_GAUSSIAN)

# per-channel normalisation constants (predict_pv_yield/netcdf_dataset.py:19-32); index 0 = HRV
SAT_MEAN = np.array([93.23458, 131.71373, 843.7779, 736.6148, 771.1189, 589.66034, 862.29816, 927.69586,
                     90.70885, 107.58985, 618.4583, 532.47394], np.float32)
SAT_STD = np.array([115.34247, 139.92636, 36.99538, 57.366386, 30.346825, 149.68007, 51.70631, 35.872967,
                    115.77212, 120.997154, 98.57828, 99.76469], np.float32)


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("predict_pv_yield_amd.optical_flow needs an MI355X (no CPU path is provided)")
    return torch.device("cuda", torch.cuda.current_device())


def _as_cuda(a) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        return a.contiguous() if a.is_cuda else a.contiguous().to(_device())
    return torch.from_numpy(np.ascontiguousarray(a)).to(_device())


def _like_input(t: torch.Tensor, template):
    return t if isinstance(template, torch.Tensor) else t.cpu().numpy()


def convert_10bpp_to_uint8(array, check_range: bool = True):
    """Convert 10 bit per pixel to uint8: round_half_even(array / 4); asserts 0 <= result <= 255."""
    x = _as_cuda(array)
    if x.dtype not in (torch.int16, torch.float32):
        x = x.to(torch.float32)
    out, flag = K.u8_from_10bit(x, 0, return_flag=True)
    if check_range:
        assert int(flag.item()) == 0, "convert_10bpp_to_uint8: values outside [0, 255] after conversion"
    return _like_input(out, array)


def calcOpticalFlowFarneback(prev, next, flow=None, pyr_scale=0.5, levels=2, winsize=40, iterations=3, poly_n=5,
                             poly_sigma=0.7, flags=OPTFLOW_FARNEBACK_GAUSSIAN):
    """cv.calcOpticalFlowFarneback signature; prev/next uint8 [H,W] -> float32 [H,W,2]."""
    if flow is not None:
        raise ValueError("an initial flow (OPTFLOW_USE_INITIAL_FLOW) is not supported")
    p, n = _as_cuda(prev), _as_cuda(next)
    out = K.farneback_pairs(p[None], n[None], pyr_scale=pyr_scale, levels=levels, winsize=winsize,
                            iterations=iterations, poly_n=poly_n, poly_sigma=poly_sigma, flags=flags)[0]
    return _like_input(out, prev)


def compute_optical_flow(sat_data, **farneback_kwargs):
    """[..., T, H, W] 10-bit counts -> optical flow fields [..., T-1, H, W, 2] float32, one per consecutive pair
    (the field's time coordinate is that of the second image of the pair, as in the reference)."""
    x = _as_cuda(sat_data)
    if x.dtype != torch.uint8:
        x = convert_10bpp_to_uint8(x)
    kw = dict(REFERENCE_FARNEBACK_KWARGS)
    kw.update(farneback_kwargs)
    return _like_input(K.farneback_stack(x, **kw), sat_data)


def weighted_average(flows, weights: Optional[Sequence[float]] = None):
    """np.average(flows, axis=0, weights=range(1, N+1)).astype(np.float32) for flows [N, H, W, 2]."""
    f = _as_cuda(flows)
    return _like_input(K.flow_weighted_mean(f[None], weights)[0], flows)


def remap_image(image, flow, border_mode: int = BORDER_CONSTANT, border_value: float = float("nan")):
    """Takes an image and warps it forwards in time according to the flow field (cv.remap, INTER_LINEAR).
    Default border = the nb-13 behaviour (BORDER_CONSTANT, NaN); pass BORDER_REPLICATE for optical_flow_1's."""
    img, fl = _as_cuda(image), _as_cuda(flow).float()
    if img.dtype not in (torch.float32, torch.uint8):
        img = img.float()
    out = K.remap_bilinear(img[None], fl[None], n_steps=1, step0=1.0, border_mode=border_mode,
                           border_value=border_value)[0, 0]
    return _like_input(out, image)


def compute_optical_flow_predictions(sat_data, flows, border_mode: int = BORDER_CONSTANT,
                                     border_value: float = float("nan")):
    """For n source images and n-1 flows, the (n-1)n/2 linear-extrapolation predictions of nb-13:
    prediction(flow_i, step) = remap(sat_data[flow_i], flows[flow_i] * step), step = 1 .. n-1-flow_i.
    Returns (predictions [P, H, W], index [P, 2] int64 = (t0 index, index of the frame the forecast is about)),
    sorted by (t0, target) like the reference's DataFrame index."""
    imgs, fl = _as_cuda(sat_data).float(), _as_cuda(flows).float()
    n = imgs.shape[0]
    num_flows = n - 1
    if fl.shape[0] != num_flows:
        raise ValueError("compute_optical_flow_predictions: need exactly len(sat_data) - 1 flows")
    h, w = imgs.shape[1:]
    n_pred = num_flows * n // 2
    preds = torch.empty((n_pred, h, w), dtype=torch.float32, device=imgs.device)
    index = []
    off = 0
    for flow_i in range(num_flows):
        steps = num_flows - flow_i
        K.remap_bilinear(imgs[flow_i:flow_i + 1], fl[flow_i:flow_i + 1], n_steps=steps, step0=1.0,
                         border_mode=border_mode, border_value=border_value, out=preds[off:off + steps][None])
        index += [(flow_i, flow_i + s) for s in range(1, steps + 1)]
        off += steps
    idx = torch.tensor(index, dtype=torch.int64)
    return _like_input(preds, sat_data), idx


# ------------------------------------------------------------------------------------------------
# scoring and the parameter search of notebooks/optical_flow_1.ipynb (cells 31, 35, 38-42)
# ------------------------------------------------------------------------------------------------
def structural_similarity(im1, im2, data_range: Optional[float] = None):
    """skimage.metrics.structural_similarity(im1, im2) with its defaults (the reference's call: optical_flow_1.ipynb cells 31,
    35, 38), on the device: two uint8 or two float32 images [H, W] -> float; stacks [N, H, W] -> float64 tensor [N]."""
    a, b = _as_cuda(im1), _as_cuda(im2)
    if a.dtype != b.dtype or a.dtype not in (torch.uint8, torch.float32):
        a, b = a.float(), b.float()
    if a.dim() == 2:
        return float(K.ssim_mean(a[None], b[None], data_range)[0])
    return K.ssim_mean(a, b, data_range)


def compute_opt_flow_and_score(images, num_timesteps: int = 23, **farneback_kwargs):
    """optical_flow_1.ipynb cell 38: the flow between images[0] and images[1] with the given Farneback arguments, then for
    i = 1 .. num_timesteps - 1 the prediction remap_image(images[1], flow * i) (BORDER_REPLICATE) scored against images[i + 1]
    by structural similarity.  images: uint8 [T >= num_timesteps + 1, H, W].  -> list of num_timesteps - 1 scores."""
    img = _as_cuda(images)
    if img.dtype != torch.uint8 or img.dim() != 3:
        raise TypeError("compute_opt_flow_and_score: a uint8 stack [T, H, W] is expected (convert_10bpp_to_uint8 first)")
    n = num_timesteps - 1
    if img.shape[0] < n + 2:
        raise ValueError(f"compute_opt_flow_and_score: {num_timesteps} time steps need {n + 2} images, got {img.shape[0]}")
    kw = dict(REFERENCE_FARNEBACK_KWARGS)
    kw.update(farneback_kwargs)
    flow = K.farneback_pairs(img[0:1], img[1:2], **kw)                                             # [1, H, W, 2]
    pred = K.remap_bilinear(img[1:2], flow, n_steps=n, step0=1.0, border_mode=BORDER_REPLICATE)     # [1, n, H, W]: flow * 1 .. n
    return [float(v) for v in K.ssim_mean(img[2:2 + n], pred[0]).tolist()]


# the grid of cell 40 (`flags` has one value there and is left out)
REFERENCE_PARAM_RANGES = dict(pyr_scale=[round(0.1 * i, 1) for i in range(1, 10)], levels=list(range(1, 10)),
                              winsize=[5, 35, 37, 40, 42, 45, 50, 60], iterations=[1, 3, 5, 10, 20, 50], poly_n=[1, 2, 3, 5, 9],
                              poly_sigma=[0.1, 0.5, 0.6, 0.7, 1, 1.1, 1.2, 1.9])


def search_farneback_params(images, param_defaults: Optional[dict] = None, param_ranges: Optional[dict] = None,
                            num_timesteps: int = 23):
    """optical_flow_1.ipynb cells 40-42: one parameter at a time is varied around `param_defaults` (default: the arguments the
    search arrived at, REFERENCE_FARNEBACK_KWARGS); the score of a setting is the mean structural similarity of its
    num_timesteps - 1 forecasts.  -> (all_scores {(name, value): mean score}, durations {(name, value): seconds},
    best {name: the value with the highest score}).  A value this implementation does not take (poly_n other than 5 or 7, a
    pyramid deeper than the smoothing kernel allows: the library says so) scores NaN and never wins."""
    import time
    defaults = dict(REFERENCE_FARNEBACK_KWARGS if param_defaults is None else param_defaults)
    ranges = REFERENCE_PARAM_RANGES if param_ranges is None else param_ranges
    all_scores, durations = {}, {}
    for name, values in ranges.items():
        for val in values:
            params = dict(defaults)
            params[name] = val
            t0 = time.time()
            try:
                scores = compute_opt_flow_and_score(images, num_timesteps, **params)
                score = float(np.mean(scores))
            except (RuntimeError, ValueError):      # the library refused the setting (check() raises with its message)
                score = float("nan")
            durations[(name, val)] = time.time() - t0
            all_scores[(name, val)] = score
    best = {}
    for name, values in ranges.items():
        scored = [(all_scores[(name, v)], v) for v in values if all_scores[(name, v)] == all_scores[(name, v)]]
        best[name] = max(scored, key=lambda sv: sv[0])[1] if scored else None
    return all_scores, durations, best


# ------------------------------------------------------------------------------------------------
# the join: advected future frames feeding the Conv3D model
# ------------------------------------------------------------------------------------------------
_MEAN_STD_ON_DEVICE = {}


def _default_mean_std(c: int, dev) -> tuple:
    """The reference's per-channel normalisation constants as device tensors, uploaded once per (channel count, device): two
    host-to-device copies from pageable memory per call cost the pipeline 60 us per batch (two copy kernels and the host
    synchronisation around them)."""
    key = (c, str(dev))
    if key not in _MEAN_STD_ON_DEVICE:
        _MEAN_STD_ON_DEVICE[key] = (torch.from_numpy(SAT_MEAN[1:1 + c] if c < 12 else SAT_MEAN[:c]).to(dev),
                                    torch.from_numpy(SAT_STD[1:1 + c] if c < 12 else SAT_STD[:c]).to(dev))
    return _MEAN_STD_ON_DEVICE[key]


def advect_future_frames(raw: torch.Tensor, n_future: int, mean: Optional[torch.Tensor] = None,
                         std: Optional[torch.Tensor] = None, border_mode: int = BORDER_REPLICATE,
                         border_value: float = float("nan"), out_dtype=torch.float32, **farneback_kwargs) -> torch.Tensor:
    """Config-3 pipeline for raw 10-bit counts [B, T, C, H, W] (int16 or float32, time-major as
    in BASELINE.json):  u8 = round(raw/4) -> T-1 Farnebäck fields per (b, c) -> weighted mean (weights 1..T-1)
    -> normalise (raw - mean_c)/std_c -> n_future frames remap(frame_t0, k * flow), k = 1..n_future.
    Returns [B, C, T + n_future, H, W] float32 in the model's NCDHW layout; the advected frames are written by the
    remap kernel straight into the future time slices (no host round trip, no intermediate copy)."""
    if not raw.is_cuda:
        raise RuntimeError("advect_future_frames: input must be on the MI355X")
    b, t, c, h, w = raw.shape
    dev = raw.device
    if mean is None:
        mean, std = _default_mean_std(c, dev)
    kw = dict(REFERENCE_FARNEBACK_KWARGS)
    kw.update(farneback_kwargs)
    if raw.dtype not in (torch.int16, torch.float32):
        raw = raw.float()
    if (h * w) % 8 == 0:
        # one pass over the raw counts: channel-major u8 stacks (each (b, c) is one Farnebäck frame stack) and the
        # normalised observed frames written straight into out[:, :, :t]
        u8, out = K.prepare_stacks(raw.contiguous(), mean, std, t + n_future)
    else:
        stacks = raw.permute(0, 2, 1, 3, 4).contiguous()
        u8 = K.u8_from_10bit(stacks, 0)
        out = torch.empty((b, c, t + n_future, h, w), dtype=torch.float32, device=dev)
        out[:, :, :t] = K.normalise(stacks, mean, std, inner=t * h * w)
    flows = K.farneback_stack(u8, **kw)                               # [B, C, T-1, H, W, 2]
    mean_flow = K.flow_weighted_mean(flows.view(b * c, t - 1, h, w, 2))  # [B*C, H, W, 2]
    # advected frames: src = normalised frame t0 = out[b, c, t-1], dst = out[b, c, t + k - 1]
    frame = h * w
    img_stride = (t + n_future) * frame
    K.remap_bilinear_strided(out.data_ptr() + (t - 1) * frame * 4, img_stride, mean_flow,
                             out.data_ptr() + t * frame * 4, img_stride, frame, b * c, n_future, 1.0, h, w,
                             border_mode, border_value)
    return out if out_dtype == torch.float32 else out.to(out_dtype)


def replace_future_frames_with_flow(sat_data: torch.Tensor, n_future: int, border_mode: int = BORDER_REPLICATE,
                                    counts_scale: float = 255.0 / 6.0, **farneback_kwargs) -> torch.Tensor:
    """For a NORMALISED model input [B, C, T, H, W]: recompute the last n_future time slices by advecting the
    last observed frame along the weighted-mean Farnebäck flow of the observed frames.  The u8 images Farnebäck
    needs are obtained by mapping the normalised values (about N(0,1)) affinely onto 0..255
    (u8 = clip(round(128 + x * counts_scale)))."""
    if not sat_data.is_cuda:
        raise RuntimeError("replace_future_frames_with_flow: input must be on the MI355X")
    b, c, t, h, w = sat_data.shape
    t_obs = t - n_future
    if t_obs < 2:
        raise ValueError("need at least two observed frames to estimate a flow")
    obs = sat_data[:, :, :t_obs].contiguous()
    # affine map onto 10-bit-like counts, then the reference's u8 conversion
    counts = ((obs * (4.0 * counts_scale)) + 512.0).clamp_(0.0, 1020.0)
    u8 = K.u8_from_10bit(counts, 0)
    kw = dict(REFERENCE_FARNEBACK_KWARGS)
    kw.update(farneback_kwargs)
    flows = K.farneback_stack(u8, **kw)
    mean_flow = K.flow_weighted_mean(flows.view(b * c, t_obs - 1, h, w, 2))
    out = sat_data.contiguous().clone()
    frame = h * w
    K.remap_bilinear_strided(out.data_ptr() + (t_obs - 1) * frame * 4, t * frame, mean_flow,
                             out.data_ptr() + t_obs * frame * 4, t * frame, frame, b * c, n_future, 1.0, h, w,
                             border_mode, float("nan"))
    return out


class AdvectingLoader:
    """Wraps an iterable of config-3 batches (dicts whose batch["satellite"]["data"] holds the OBSERVED frames as raw
    int16 counts [B, T_obs, C, H, W]) and yields the same batches with that entry replaced by the advected model input
    [B, C, T_obs + n_future, H, W] (the reference fans `compute_optical_flow` out to a process pool in front of training,
    notebooks/13_...ipynb:175-240).  The tensor that is handed over is tagged so that
    Model(future_frames="optical_flow") takes it as is; results are bit-identical to the inline call.

    The advection runs on the CONSUMER's stream, right before the batch is yielded.  Rounds 2-3 ran it one batch ahead on a
    side HIP stream under the train step of the previous batch: both stages fill the chip, so the overlap bought 3 % at best
    (3.04-3.16 against 3.03-3.24 ms per batch) and on the round-3 driver's device it cost 50 % (4.97 against 3.24 ms:
    the side stream's allocator pool, record_stream's deferred reuse).  A mode that cannot beat the plain call by 5 % is not
    worth a second stream's failure modes: removed in round 4."""

    def __init__(self, loader, n_future: int, **advect_kwargs):
        self.loader, self.n_future, self.kw = loader, n_future, advect_kwargs

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for batch in self.loader:
            out = advect_future_frames(batch["satellite"]["data"], self.n_future, **self.kw)
            out._pv_advected = True
            new = dict(batch)
            new["satellite"] = dict(batch["satellite"], data=out)
            yield new
