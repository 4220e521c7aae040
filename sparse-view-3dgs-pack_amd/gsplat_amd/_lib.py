"""Loads the product library csrc/libgsplat_hip.so (prefix ``gs_``).

There is deliberately no fallback: if the HIP library has not been built, or the process has no
GPU, every operator raises.
"""
import os
import threading

# torch must be loaded BEFORE the library: both link libamdhip64 and have to share ONE HIP runtime (the
# streams and device pointers handed across the C ABI come from torch).  Loading the library first would
# bind it to a second copy of the runtime and every launch would fail with hipErrorNoDevice.
import torch  # noqa: F401,E402

from .capi import CApi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libgsplat_hip.so")

_lock = threading.Lock()
_api = None


def hip_api():
    global _api
    if _api is None:
        with _lock:
            if _api is None:
                _api = CApi(LIB_PATH, "gs_")
    return _api
