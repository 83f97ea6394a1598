"""CPU: pins oracle/pv_oracle.c against OpenCV's OWN outputs when the fixture tests/golden/flow_cv2.npz exists
(written by tests/golden/make_flow_golden.py on a machine that has cv2; the build image has not, so until then this
module is skipped and the flow oracle stays "parity unpinned" -- analytic pins only, tests/test_oracle_flow.py)."""
import os

import numpy as np
import pytest

from oracle import flow_oracle as fo

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "flow_cv2.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(FIXTURE), reason="cv2-verified fixture not generated yet (no OpenCV here)")


@pytest.fixture(scope="module")
def golden():
    return np.load(FIXTURE)


def test_fixture_names_its_opencv_version(golden):
    assert str(golden["cv2_version"]).split(".")[0] in ("3", "4", "5")


def test_farneback_matches_cv2(golden):
    names = sorted({k.split("/")[0] for k in golden.files if k.startswith("fb_")})
    assert len(names) == 10
    for name in names:
        frames, ref = golden[name + "/frames"], golden[name + "/flow"]
        flow = fo.calc_optical_flow_farneback(frames[0], frames[1])
        # same algorithm, possibly different summation order inside OpenCV's SIMD paths: sub-milli-pixel agreement
        assert np.abs(flow - ref).max() <= 2e-3, (name, float(np.abs(flow - ref).max()))


def test_remap_matches_cv2_bit_for_bit(golden):
    for name in ("R1", "R2", "R3", "R4", "Rrand"):
        img, flow = golden[name + "/image"], golden[name + "/flow"]
        rep = fo.remap_image(img, flow, 1.0, fo.BORDER_REPLICATE)
        assert np.array_equal(rep, golden[name + "/replicate"]), name
        con = fo.remap_image(img, flow, 1.0, fo.BORDER_CONSTANT, np.nan)
        ref = golden[name + "/constant_nan"]
        assert np.array_equal(np.isnan(con), np.isnan(ref)) and np.array_equal(con[~np.isnan(ref)], ref[~np.isnan(ref)]), name
    img, flow = golden["R5/image"], golden["R5/flow"]
    assert np.array_equal(fo.remap_image(img, flow, 1.0, fo.BORDER_REPLICATE), golden["R5/replicate"])


# ---- the HIP path against the same fixture (the day flow_cv2.npz lands, rows a-10 / a-13 are pinned without new code) ----
@pytest.mark.gpu
def test_hip_farneback_matches_cv2(golden, device):
    import torch
    from predict_pv_yield_amd import hip_ops as K
    for name in sorted({k.split("/")[0] for k in golden.files if k.startswith("fb_")}):
        frames, ref = golden[name + "/frames"], golden[name + "/flow"]
        prev, nxt = (torch.from_numpy(np.ascontiguousarray(f[None])).to(device) for f in frames)
        flow = K.farneback_pairs(prev, nxt)[0].cpu().numpy()
        assert np.abs(flow - ref).max() <= 2e-3, (name, float(np.abs(flow - ref).max()))


@pytest.mark.gpu
def test_hip_remap_matches_cv2_bit_for_bit(golden, device):
    import torch
    from predict_pv_yield_amd import optical_flow as of
    for name in ("R1", "R2", "R3", "R4", "Rrand", "R5"):
        img, flow = torch.from_numpy(golden[name + "/image"]).to(device), torch.from_numpy(golden[name + "/flow"]).to(device)
        # remap_image builds the reference's map (meshgrid - flow, 13_...ipynb:268-272) itself
        rep = of.remap_image(img, flow, border_mode=of.BORDER_REPLICATE).cpu().numpy()
        assert np.array_equal(rep, golden[name + "/replicate"]), name
        if name != "R5":
            con = of.remap_image(img, flow, border_mode=of.BORDER_CONSTANT, border_value=float("nan")).cpu().numpy()
            ref = golden[name + "/constant_nan"]
            assert np.array_equal(np.isnan(con), np.isnan(ref)) and np.array_equal(con[~np.isnan(ref)], ref[~np.isnan(ref)]), name
