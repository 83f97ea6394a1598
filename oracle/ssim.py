"""skimage.metrics.structural_similarity with the arguments the reference uses -- `metrics.structural_similarity(ground_truth,
remapped_image)` on uint8 images, every option at its default (notebooks/optical_flow_1.ipynb cells 31, 35, 38: the only
QUALITY number the reference computes for its optical-flow forecasts).

TEST INFRASTRUCTURE ONLY.  scikit-image is a third-party dependency of the reference's notebooks (unpinned; 0.18.3 is the version
importable in this image under /opt/conda/bin/python3.9 and the one the goldens were made with:
tests/golden/make_ssim_golden.py).  Restated from its published definition (Wang, Bovik, Sheikh, Simoncelli, "Image quality
assessment: from error visibility to structural similarity", IEEE TIP 2004) with that function's defaults:
  7 x 7 uniform window (scipy.ndimage.uniform_filter, mode "reflect"), images as float64, sample covariance
  (x N / (N - 1), N = 49), K1 = 0.01, K2 = 0.03, data_range = 255 for uint8 (the dtype's range), mean of the SSIM map with a
  border of (7 - 1) / 2 = 3 pixels cropped.
Pinned by tests/test_oracle_flow.py::test_ssim_restatement_reproduces_skimage against tests/golden/ssim_skimage.npz."""
import numpy as np
from scipy.ndimage import uniform_filter


def structural_similarity(im1: np.ndarray, im2: np.ndarray, win_size: int = 7, data_range=None) -> float:
    if im1.shape != im2.shape or im1.ndim != 2:
        raise ValueError("structural_similarity: two 2-D images of one shape")
    if data_range is None:
        if im1.dtype != im2.dtype:
            raise ValueError("structural_similarity: give data_range for images of different dtypes")
        if im1.dtype == np.uint8:
            data_range = 255.0
        elif np.issubdtype(im1.dtype, np.floating):
            data_range = 2.0        # skimage's dtype range of float images is [-1, 1]
        else:
            info = np.iinfo(im1.dtype)
            data_range = float(info.max) - float(info.min)
    a, b = im1.astype(np.float64), im2.astype(np.float64)
    n = win_size * win_size
    cov_norm = n / (n - 1.0)
    ux, uy = uniform_filter(a, size=win_size), uniform_filter(b, size=win_size)
    uxx, uyy, uxy = uniform_filter(a * a, size=win_size), uniform_filter(b * b, size=win_size), uniform_filter(a * b, size=win_size)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    pad = (win_size - 1) // 2
    return float(s[pad:-pad, pad:-pad].mean(dtype=np.float64))
