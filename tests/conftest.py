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
    # torch's CPU pool like the oracle's (oracle_lib.host_cpu_share): the CPUs this process is GRANTED, not the ones it sees
    try:
        import torch
        import oracle_lib
        if "OMP_NUM_THREADS" not in os.environ:
            torch.set_num_threads(oracle_lib.host_cpu_share())
    except Exception:
        pass


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle bound through the same ctypes prototypes as the product library."""
    import oracle_lib
    return oracle_lib.get()


@pytest.fixture(autouse=True)
def _gpu_parity_uses_the_exact_chain(request):
    """GPU parity tests hold EVERY gradient tensor to 1e-4.  For dL_dscales / dL_drotations / dL_dcov3D the oracle's fp32
    transcription of the reference formula is not a meaningful arbiter at that bar (it is 2e-4 ... 2e-3 away from the exact
    image of its own inputs), so while a test marked `gpu` runs the oracle evaluates that chain in
    double (Oracle.exact_chain).  tests/test_gpu_fullsize.py switches it per call and reports the fp32 distance too; the
    CPU suite keeps testing the fp32 transcription (against tests/dense_reference.py at its own, wider, bars)."""
    if request.node.get_closest_marker("gpu") is None or "oracle" not in request.fixturenames:
        yield
        return
    with request.getfixturevalue("oracle").exact_chain():
        yield


@pytest.fixture(scope="session")
def hip():
    """The product backend (libgsplat_hip.so).  Raises if it cannot be loaded - GPU tests must never
    pass on a fallback."""
    import torch
    assert torch.cuda.is_available(), "GPU test running without a GPU"
    from gsplat_amd import hip_backend
    return hip_backend()


def pytest_sessionfinish(session, exitstatus):
    """Measured threshold-flip counts of the GPU parity tests (test_gpu_raster_parity.FLIPS) -> gpurun_out/flip_counts.json."""
    mod = sys.modules.get("test_gpu_raster_parity")
    flips = getattr(mod, "FLIPS", None) if mod else None
    if flips:
        import json
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            json.dump(flips, open(os.path.join(ROOT, "gpurun_out", "flip_counts.json"), "w"), indent=1)
        except OSError:
            pass
