"""The C-ABI library loads (no GPU needed) and exports exactly what include/gsplat.h declares; the
ctypes prototype table covers the same set; the product path fails loudly without a GPU."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gsplat.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", src)))


def exported(path, prefix):
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    return sorted({l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith(prefix)})


def test_header_declares_what_the_prototype_table_binds():
    from gsplat_amd.capi import PROTOTYPES
    assert header_functions() == sorted("gs_" + n for n in PROTOTYPES)


def test_hip_library_exports_every_declared_symbol_and_loads():
    from gsplat_amd._lib import LIB_PATH, hip_api
    assert os.path.exists(LIB_PATH), "libgsplat_hip.so not built (python __graft_entry__.py)"
    assert exported(LIB_PATH, "gs_") == header_functions()
    api = hip_api()  # dlopen + bind every prototype; no device call
    assert api.raw("abi_version")() == 7
    assert b"gfx950" in api.raw("build_info")()
    # the ctypes mirrors have the layout the library was compiled with
    from gsplat_amd import capi
    for which, cls in enumerate((capi.GsView, capi.GsGaussians, capi.GsScratch, capi.GsGrads, capi.GsStepState,
                                 capi.GsLgdwtParams, capi.GsAdamSeg)):
        assert api.raw("struct_bytes")(which) == ctypes.sizeof(cls), cls.__name__
    assert api.raw("struct_bytes")(99) == 0
    # pure host entry: scratch sizing
    out = (ctypes.c_size_t * 3)()
    ws = ctypes.c_size_t(0)
    api.call("scratch_bytes", 1000, 1920, 1080, 50000, out, ctypes.byref(ws))
    assert out[0] >= 1000 * 64 and out[1] >= 1920 * 1080 * 8 and out[2] >= 50000 * 16 and ws.value >= 1000 * 128
    assert api.raw("scratch_bytes")(-1, 10, 10, 0, out, None) == -2  # GS_E_SHAPE
    assert api.raw("scratch_bytes")(1, 10, 10, 0, None, None) == -1  # GS_E_NULL


def test_oracle_exports_the_same_abi_under_its_own_prefix(oracle):
    from gsplat_amd.capi import DEVICE_ONLY, PROTOTYPES
    import oracle_lib
    have = set(exported(oracle_lib.ORACLE_SO, "gso_"))
    want = {"gso_" + n for n in PROTOTYPES if n not in DEVICE_ONLY}
    assert want <= have


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_a_gpu():
    """No silent CPU fallback: the drop-in packages refuse CPU tensors."""
    import diff_gaussian_rasterization as dgr
    from simple_knn._C import distCUDA2
    from gsplat_amd import synthetic
    from helpers import settings_for
    cam = synthetic.look_at_camera((3.0, 0.0, 0.0), 32, 32)
    rs = settings_for(dgr.GaussianRasterizationSettings, cam, torch.zeros(3), 0, torch.device("cpu"))
    m = torch.zeros((4, 3))
    with pytest.raises(RuntimeError, match="no fallback"):
        dgr.GaussianRasterizer(rs)(means3D=m, means2D=m, opacities=torch.ones((4, 1)), colors_precomp=torch.ones((4, 3)),
                                   scales=torch.ones((4, 3)), rotations=torch.ones((4, 4)))
    with pytest.raises(RuntimeError, match="no CPU path"):
        distCUDA2(torch.zeros((8, 3)))


def test_product_sources_never_reference_the_oracle():
    """The shipped package must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "sparse-view-3dgs-pack_amd")
    bad = []
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"oracle_lib|libgs_oracle|gso_|oracle/", txt):
                    bad.append(os.path.join(dp, f))
    assert bad == []
