import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "sparse-view-3dgs-pack_amd")
for p in (PKG, os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: larger CPU cases")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle bound through the same ctypes prototypes as the product library."""
    import oracle_lib
    return oracle_lib.get()


@pytest.fixture(scope="session")
def hip():
    """The product backend (libgsplat_hip.so).  Raises if it cannot be loaded - GPU tests must never
    pass on a fallback."""
    import torch
    assert torch.cuda.is_available(), "GPU test running without a GPU"
    from gsplat_amd import hip_backend
    return hip_backend()
