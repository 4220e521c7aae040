"""gsplat_amd - MI355X-native core behind the reference's operator API.

Sub-modules:
  capi    ctypes description of include/gsplat.h
  _lib    loads csrc/libgsplat_hip.so (no fallback)
  raster  torch <-> C-ABI glue for the rasterizer (the reference's `_C` entry points)
"""
from .capi import CApi, GsError  # noqa: F401

__all__ = ["CApi", "GsError", "hip_backend"]

_backend = None


def hip_backend():
    """The process-wide RasterBackend bound to libgsplat_hip.so."""
    global _backend
    if _backend is None:
        from ._lib import hip_api
        from .raster import RasterBackend
        _backend = RasterBackend(hip_api(), "cuda")
    return _backend
