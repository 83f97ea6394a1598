"""Generates tests/golden/flow_cv2.npz: inputs and OpenCV's own outputs for the optical-flow half of the path, to PIN
oracle/pv_oracle.c against the library the reference actually calls
(cv.calcOpticalFlowFarneback at notebooks/13_3d_conv_with_optical_flow_predictions.ipynb:133-135, cv.remap at :275-281).

OpenCV is not installed in the build image (and the reference neither declares nor pins it), so this script cannot run
there: the flow oracle is "parity unpinned" until someone runs

    pip install opencv-python-headless        # any 4.5+ wheel
    python tests/golden/make_flow_golden.py

on a machine that has it and commits the resulting .npz (inputs + expected outputs only: a fixture, no source).
tests/test_oracle_flow_cv2.py consumes the file when it exists and is skipped otherwise.

Cases: K1 identical frames; K2 translated dense textures at 64x64 (one coarse level) and 160x200 (two);
R1 integer flow, R2 multiples of 1/32 px, R3 cvRound ties + NaN coordinates, R4 NaN border, R5 u8 fixed point; both border
modes the notebooks use (BORDER_CONSTANT with NaN, BORDER_REPLICATE).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "flow_cv2.npz")

FB_ARGS = dict(pyr_scale=0.5, levels=2, winsize=40, iterations=3, poly_n=5, poly_sigma=0.7)


def farneback_cases():
    from predict_pv_yield_amd.data.synthetic import blob_texture_sequence
    cases = {}
    rng = np.random.default_rng(5)
    for hw in ((64, 64), (160, 200)):
        for v in ((0.0, 0.0), (1.0, 0.0), (0.0, -2.0), (1.5, 0.75), (3.0, -2.0)):
            seq = blob_texture_sequence(rng, 2, hw[0], hw[1], v)
            counts = np.clip(np.rint(seq), 0, 1023)
            u8 = (counts / 4.0).round().astype(np.uint8)          # convert_10bpp_to_uint8, 13_...ipynb:112-119
            cases[f"fb_{hw[0]}x{hw[1]}_v{v[0]}_{v[1]}"] = u8
    return cases


def remap_cases():
    rng = np.random.default_rng(2)
    cases = {}
    img = rng.normal(size=(20, 24)).astype(np.float32)
    cases["R1"] = (img, np.stack([np.full((20, 24), 2, np.float32), np.full((20, 24), -1, np.float32)], -1))
    cases["R2"] = (img, (rng.integers(-64, 64, (20, 24, 2)) / 32.0).astype(np.float32))
    fl = np.zeros((20, 24, 2), np.float32)
    fl[0, 1, 0], fl[0, 2, 0] = -0.5 / 32, -1.5 / 32
    fl[1, 1] = [np.nan, 0]
    cases["R3"] = (img, fl)
    cases["R4"] = (np.ones((6, 7), np.float32), np.zeros((6, 7, 2), np.float32))
    cases["Rrand"] = (img, rng.normal(0, 3, (20, 24, 2)).astype(np.float32))
    u8 = rng.integers(0, 256, (20, 24)).astype(np.uint8)
    cases["R5"] = (u8, rng.normal(0, 2, (20, 24, 2)).astype(np.float32))
    return cases


def main():
    try:
        import cv2
    except ImportError:
        print("cv2 is not importable here: install opencv-python-headless and re-run (see the module docstring)")
        return 2
    out = {"cv2_version": np.array(cv2.__version__)}
    for name, pair in farneback_cases().items():
        flow = cv2.calcOpticalFlowFarneback(pair[0], pair[1], None, flags=cv2.OPTFLOW_FARNEBACK_GAUSSIAN, **FB_ARGS)
        out[name + "/frames"] = pair
        out[name + "/flow"] = flow.astype(np.float32)
    for name, (img, flow) in remap_cases().items():
        h, w = img.shape
        # the reference's map construction (13_...ipynb:268-272): -flow, then += arange along each axis, in float32
        remap = -flow.copy()
        remap[..., 0] += np.arange(w)
        remap[..., 1] += np.arange(h)[:, np.newaxis]
        out[name + "/image"] = img
        out[name + "/flow"] = flow
        if img.dtype == np.uint8:
            out[name + "/replicate"] = cv2.remap(img, remap, None, cv2.INTER_LINEAR, borderMode=cv2.BORDER_REPLICATE)
        else:
            out[name + "/replicate"] = cv2.remap(img, remap, None, cv2.INTER_LINEAR, borderMode=cv2.BORDER_REPLICATE)
            out[name + "/constant_nan"] = cv2.remap(img, remap, None, cv2.INTER_LINEAR, borderMode=cv2.BORDER_CONSTANT,
                                                    borderValue=np.nan)
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} with OpenCV {cv2.__version__}: {len(out)} arrays")
    return 0


if __name__ == "__main__":
    sys.exit(main())
