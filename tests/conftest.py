import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def ref_order(param, tensor=None):
    """`tensor` (the parameter itself by default; a gradient, a moment ...) in the reference's column order: the bf16 Conv3D
    model stores fc1.weight channels-last along its input axis (predict_pv_yield_amd/models/conv3d/_fc1_layout.py)."""
    from predict_pv_yield_amd.models.conv3d._fc1_layout import reference_layout
    return reference_layout(param, tensor)
