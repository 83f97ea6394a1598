"""Generates tests/golden/ssim_skimage.npz: structural-similarity scores of optical-flow forecasts, computed by scikit-image
itself (the third-party package the reference scores its forecasts with: notebooks/optical_flow_1.ipynb cells 31, 35, 38).

Two stages, because no interpreter of this image has both halves:
  1. python3 tests/golden/make_ssim_golden.py inputs          (system python: the repo's oracle)
       a synthetic advection sequence -> uint8 frames (convert mode 1, optical_flow_1's), Farneback flows of the warm-up
       pairs (oracle/pv_oracle.c), their weighted average, and the forecasts remap(image_t0, flow * i) for i = 0..N-1
       (uint8, BORDER_REPLICATE) -> tests/golden/_ssim_inputs.npz (scratch, not committed)
  2. /opt/conda/bin/python3.9 tests/golden/make_ssim_golden.py scores      (scikit-image 0.18.3)
       metrics.structural_similarity(ground_truth, forecast) and (image_t0, ground_truth) [persistence] with default
       arguments -> tests/golden/ssim_skimage.npz = the inputs + the scores + the scikit-image version.
The committed fixture holds arrays only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SCRATCH = os.path.join(HERE, "_ssim_inputs.npz")
OUT = os.path.join(HERE, "ssim_skimage.npz")
WARM_UP, STEPS, H, W = 6, 6, 128, 160
VELOCITY = (1.3, -0.8)
FARNEBACK = dict(pyr_scale=0.5, levels=2, winsize=40, iterations=3, poly_n=5, poly_sigma=0.7)


def stage_inputs():
    sys.path.insert(0, ROOT)
    from oracle import flow_oracle as fo
    from predict_pv_yield_amd.data.synthetic import blob_texture_sequence
    rng = np.random.default_rng(20211005)
    seq = blob_texture_sequence(rng, WARM_UP + STEPS, H, W, VELOCITY)
    counts = np.clip(np.rint(seq), 0, 1023).astype(np.int16)
    frames, flag = fo.convert_10bpp_to_uint8(counts, 1)             # optical_flow_1.ipynb:129-134 (trunc(x / 1023 * 255))
    assert not flag
    flows = np.stack([fo.calc_optical_flow_farneback(frames[i], frames[i + 1], **FARNEBACK) for i in range(WARM_UP - 1)])
    flow = fo.weighted_average(flows)                               # optical_flow_1.ipynb:293-294
    image_t0 = frames[WARM_UP - 1]
    forecasts = np.stack([fo.remap_image(image_t0, flow, float(i), fo.BORDER_REPLICATE, 0) for i in range(STEPS + 1)])
    np.savez(SCRATCH, frames=frames, flows=flows, flow=flow, forecasts=forecasts)
    print("wrote", SCRATCH, frames.shape, forecasts.shape)


def stage_scores():
    import skimage
    from skimage import metrics
    d = np.load(SCRATCH)
    frames, forecasts = d["frames"], d["forecasts"]
    image_t0 = frames[WARM_UP - 1]
    ssim_flow = np.array([metrics.structural_similarity(frames[WARM_UP - 1 + i], forecasts[i]) for i in range(STEPS + 1)])
    ssim_persistence = np.array([metrics.structural_similarity(image_t0, frames[WARM_UP - 1 + i]) for i in range(STEPS + 1)])
    np.savez_compressed(OUT, frames=frames, flow=d["flow"], forecasts=forecasts, ssim_flow=ssim_flow,
                        ssim_persistence=ssim_persistence, warm_up=np.int64(WARM_UP),
                        skimage_version=np.array([int(v) for v in skimage.__version__.split(".")[:3]]))
    print("wrote", OUT, "skimage", skimage.__version__)
    print("flow        ", np.round(ssim_flow, 4))
    print("persistence ", np.round(ssim_persistence, 4))


if __name__ == "__main__":
    {"inputs": stage_inputs, "scores": stage_scores}[sys.argv[1]]()
