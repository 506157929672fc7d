import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_golden.npz"))


@pytest.fixture(scope="session")
def hip():
    """The ctypes binding of liblsfa_hip.so on cuda:0 (GPU tests only)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from lsfa_amd import hip as _hip
    return _hip
