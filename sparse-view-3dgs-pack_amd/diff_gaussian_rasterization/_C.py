"""Stand-in for the reference's compiled extension module `diff_gaussian_rasterization._C`
(diff-gaussian-rasterization/ext.cpp:15-19): the same three entry points, same positional
arguments and return tuples, served by libgsplat_hip.so.

Deliberately NOT exported: `fusedssim` / `fusedssim_backward` (LGDWT-GS/utils/loss_utils.py:16-19
probes for them in a try/except; the fused SSIM lives in the separate `fused_ssim` package here).
"""
from gsplat_amd import hip_backend


class _Backend:
    """`_C.backend` -> the process-wide RasterBackend (lazy: loading the library needs torch + the .so)."""

    def __getattr__(self, name):
        return getattr(hip_backend(), name)

    def __setattr__(self, name, value):
        setattr(hip_backend(), name, value)


backend = _Backend()


def rasterize_gaussians(*args, **kw):
    return hip_backend().rasterize_gaussians(*args, **kw)


def rasterize_gaussians_backward(*args, **kw):
    return hip_backend().rasterize_gaussians_backward(*args, **kw)


def mark_visible(*args):
    return hip_backend().mark_visible(*args)
