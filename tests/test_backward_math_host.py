"""The product's per-Gaussian backward math (csrc/gs_backward_math.h: SH, cov2D / anti-aliasing, cov3D -> scale /
quaternion, projection - written from the derivation, not from the reference's expansion) run on the HOST through
tests/tools/backward_math_host.hip, against the oracle's transcription of backward.cu:23-449 fed the SAME per-Gaussian
sums.  No GPU involved: the functions are __host__ __device__ and the kernel calls exactly these."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from gsplat_amd import synthetic
from gsplat_amd.capi import GsGrads
from test_oracle_dense import small_scene

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "tools", "backward_math_host.hip")
SO = os.path.join(HERE, "tools", "_build", "libbackward_math_host.so")
CSRC = os.path.join(os.path.dirname(HERE), "sparse-view-3dgs-pack_amd", "csrc")


@pytest.fixture(scope="module")
def host_math():
    deps = [SRC] + [os.path.join(CSRC, f) for f in ("gs_backward_math.h", "gs_math.h", "gs_common.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(SO) < os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off",
                               "-fPIC", "-shared", "-Wno-unused-function", "-o", SO, SRC])
    lib = C.CDLL(SO)
    lib.bm_backward_from_rows.restype = C.c_int
    return lib


CASES = [
    dict(P=300, seed=11, W=96, H=64, aa=False, depth_mode=0, deg=3),
    dict(P=300, seed=12, W=80, H=72, aa=True, depth_mode=1, deg=3),
    dict(P=200, seed=13, W=64, H=64, aa=True, depth_mode=0, deg=1),
    dict(P=200, seed=14, W=64, H=48, aa=False, depth_mode=2, deg=2),
    dict(P=150, seed=15, W=64, H=48, aa=True, depth_mode=1, deg=0, precomp=True, mod=0.8),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "s%d_aa%d_d%d_deg%d" % (c["seed"], c["aa"], c["depth_mode"], c["deg"]))
def test_host_run_of_the_kernel_math_matches_the_oracle_stage_two(oracle, host_math, case):
    P, W, H = case["P"], case["W"], case["H"]
    sc = synthetic.trained_like(P, seed=case["seed"], sh_degree=case["deg"], scale_mult=1.5)
    sc["scale_modifier"] = case.get("mod", 1.0)
    if case.get("precomp"):
        sc = small_scene(P, case["seed"], precomp_color=True, precomp_cov=True)
    cam = synthetic.look_at_camera((2.6, 1.1, 0.9), W, H)
    be = oracle.backend
    cpu = torch.device("cpu")
    e = torch.empty(0)
    g = lambda k: sc[k] if sc.get(k) is not None else e  # noqa: E731
    bg = torch.zeros(3)
    R, color, radii, geom, binning, img, invd = be.rasterize_gaussians(
        bg, sc["means3D"], g("colors_precomp"), sc["opacities"], g("scales"), g("rotations"), sc.get("scale_modifier", 1.0),
        g("cov3D_precomp"), cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W, g("shs"),
        sc.get("sh_degree", 0), cam.camera_center, False, case["aa"], False)
    assert int((radii > 0).sum()) > P // 3
    st = be.export_state(P, W, H, R, geom, binning, img)
    gen = torch.Generator().manual_seed(case["seed"])
    rows = torch.zeros((P, 16), dtype=torch.float64)  # the row layout of gs_backward_from_rows: float64 slots
    rows[:, :10] = (torch.randn((P, 10), generator=gen) * torch.tensor([3.0, 3.0, 50.0, 50.0, 50.0, 1.0, 1.0, 1.0, 1.0, 2.0])).double()
    rows[radii <= 0] = 0  # the blend backward never touches a culled Gaussian
    args = (rows, bg, sc["means3D"], radii, g("colors_precomp"), sc["opacities"], g("scales"), g("rotations"),
            sc.get("scale_modifier", 1.0), g("cov3D_precomp"), cam.world_view_transform, cam.full_proj_transform,
            cam.tanfovx, cam.tanfovy, H, W, g("shs"), sc.get("sh_degree", 0), cam.camera_center, geom, case["aa"])
    want = be.backward_from_rows(*args, depth_mode=case["depth_mode"])

    # the same through the product's math on the host
    keep = []
    view = be._view(keep, cpu, bg, cam.world_view_transform, cam.full_proj_transform, cam.camera_center, cam.tanfovx,
                    cam.tanfovy, H, W, sc.get("scale_modifier", 1.0), sc.get("sh_degree", 0), False, case["aa"], False)
    gg = be._gauss(keep, cpu, sc["means3D"], g("shs"), g("colors_precomp"), sc["opacities"], g("scales"), g("rotations"),
                   g("cov3D_precomp"))
    names = ("means2D", "colors", "opacity", "means3D", "cov3D", "sh", "scales", "rotations")
    got = {n: (None if w is None else torch.zeros_like(w)) for n, w in zip(names, want)}
    grads = GsGrads()
    p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    grads.dL_dmeans3D, grads.dL_dmeans2D, grads.dL_dsh = p(got["means3D"]), p(got["means2D"]), p(got["sh"])
    grads.dL_dcolors, grads.dL_dopacity = p(got["colors"]), p(got["opacity"])
    grads.dL_dscales, grads.dL_drotations, grads.dL_dcov3D = p(got["scales"]), p(got["rotations"]), p(got["cov3D"])
    cov3D, clamped = st["cov3D"].contiguous(), st["clamped"].contiguous()
    rc = host_math.bm_backward_from_rows(C.byref(view), C.byref(gg), C.c_void_p(radii.data_ptr()),
                                         C.c_void_p(cov3D.data_ptr()), C.c_void_p(clamped.data_ptr()),
                                         C.c_void_p(rows.data_ptr()), case["depth_mode"], C.byref(grads))
    assert rc == 0
    report = {}
    for n, w in zip(names, want):
        if w is None:
            continue
        scale = max(float(w.abs().max()), 1e-20)
        report[n] = float((got[n] - w).abs().max()) / scale
        assert float(w.abs().max()) > 0 or n in ("scales", "rotations"), n
    print(report)
    # two fp32 evaluation orders of the same algebra: everything agrees to a few ulps of the tensor's largest entry
    # (measured <= 4e-6) except single entries of the conic -> covariance -> scale / rotation chain, which amplifies
    # rounding for needle-shaped footprints (measured 2.8e-5 on one of these cases): the stated
    # bar, 1e-4 of the tensor's max, is asserted for those
    for n, e_ in report.items():
        assert e_ <= (1e-4 if n in ("scales", "rotations", "cov3D", "means3D") else 2e-6), (n, e_, report)
