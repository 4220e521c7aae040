"""The FSGS rasterizer generation (dgr_fsgs: colour + depth + alpha outputs, confidence-scaled gradients) on the
CPU: the oracle's restatement of FSGS/submodules/diff-gaussian-rasterization-confidence against the independent
dense float64 autograd formulation, and the Python-side contract of dgr_fsgs/__init__.py."""
import pytest
import torch

import dgr_fsgs
from gsplat_amd import synthetic
from test_oracle_dense import small_scene
import dense_reference

LEAVES = ("means3D", "opacities", "shs", "scales", "rotations")


def fsgs_run(Rast, Settings, sc, cam, bg, device, dL_c, dL_d, dL_a, confidence=None):
    p = {k: sc[k].detach().clone().to(device).requires_grad_(True) for k in LEAVES}
    P = p["means3D"].shape[0]
    conf = torch.ones((P, 1), device=device) if confidence is None else confidence.to(device)
    rs = Settings(image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                  bg=bg.to(device), scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(device),
                  projmatrix=cam.full_proj_transform.to(device), sh_degree=sc.get("sh_degree", 3),
                  campos=cam.camera_center.to(device), prefiltered=False, debug=False, confidence=conf)
    m2 = torch.zeros_like(p["means3D"], requires_grad=True)
    color, radii, depth, alpha = Rast(rs)(means3D=p["means3D"], means2D=m2, opacities=p["opacities"], shs=p["shs"],
                                          scales=p["scales"], rotations=p["rotations"])
    ((color * dL_c.to(device)).sum() + (depth * dL_d.to(device)).sum() + (alpha * dL_a.to(device)).sum()).backward()
    g = {k: v.grad.detach().cpu() for k, v in p.items()}
    g["means2D"] = m2.grad.detach().cpu()
    return dict(color=color.detach().cpu(), depth=depth.detach().cpu(), alpha=alpha.detach().cpu(), radii=radii.cpu(),
                grads=g)


def dense_fsgs(sc, cam, bg, dL_c, dL_d, dL_a):
    d, leaves = {}, {}
    for k, v in sc.items():
        if torch.is_tensor(v):
            leaves[k] = v.double().clone().requires_grad_(True)
            d[k] = leaves[k]
        else:
            d[k] = v
    P = sc["means3D"].shape[0]
    leaves["ndc_probe"] = torch.zeros((P, 2), dtype=torch.float64, requires_grad=True)
    d["ndc_probe"] = leaves["ndc_probe"]
    out = dense_reference.render(d, cam, bg, False)
    ((out["color"] * dL_c.double()).sum() + (out["depth"] * dL_d.double()).sum() + (out["alpha"] * dL_a.double()).sum()).backward()
    return out, {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}


@pytest.mark.parametrize("seed,eye,bgv,big", [(1, (3.2, 1.0, 1.5), (0.0, 0.0, 0.0), False),
                                             (3, (0.6, -1.4, 0.4), (0.2, 0.9, 0.1), True),
                                             (5, (2.0, 2.0, 2.0), (1.0, 1.0, 1.0), False)])
def test_oracle_fsgs_matches_dense_float64(oracle, seed, eye, bgv, big):
    sc = small_scene(110, seed, False, False, 3, big)
    W, H = 44, 36
    cam = synthetic.look_at_camera(eye, W, H, FoVx=0.9)
    bg = torch.tensor(bgv)
    g = torch.Generator().manual_seed(seed)
    dL_c, dL_d, dL_a = (torch.randn((3, H, W), generator=g), torch.randn((1, H, W), generator=g) * 0.3,
                        torch.randn((1, H, W), generator=g))
    o = fsgs_run(oracle.FsgsRasterizer, oracle.FsgsSettings, sc, cam, bg, torch.device("cpu"), dL_c, dL_d, dL_a)
    dn, dg = dense_fsgs(sc, cam, bg, dL_c, dL_d, dL_a)
    assert torch.equal(o["radii"].long(), dn["radii"].long())
    for k in ("color", "depth", "alpha"):
        err = (o[k].double() - dn[k]).abs()
        assert int((err > 2e-5 * max(1.0, float(dn[k].abs().max()))).sum()) <= 2, k
    assert float(o["alpha"].max()) <= 1.0 + 1e-6 and float(o["alpha"].min()) >= 0.0
    names = {"means3D": "means3D", "opacities": "opacities", "shs": "shs", "scales": "scales", "rotations": "rotations"}
    # The reference's backward reads T_final back as 1 - alpha_image (-confidence backward.cu:461): on saturated
    # pixels (T ~ 1e-4) that subtraction keeps only ~3 digits of T, so its gradients carry up to ~1e-3 of fp32
    # noise that the product form does not have.  With the probe GSO_FSGS_T=1 (exact T kept by the oracle) the
    # same comparison holds at 2e-4 - run below in a subprocess for the saturated scene.
    import os
    tol = 2e-4 if (not big or os.environ.get("GS_FSGS_TIGHT")) else 1.5e-3
    for k, dk in names.items():
        a, b = o["grads"][k].double(), dg[dk].reshape(o["grads"][k].shape)
        s = max(1e-12, float(b.abs().max()))
        assert float((a - b).abs().max()) <= tol * s, (k, float((a - b).abs().max()) / s)


def test_saturated_scene_gap_is_the_alpha_readback():
    import os
    import subprocess
    import sys
    if os.environ.get("GSO_FSGS_T"):
        pytest.skip("already inside the probe run")
    env = dict(os.environ, GSO_FSGS_T="1", GS_FSGS_TIGHT="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", __file__, "-k", "dense_float64 and 3-eye1"], env=env,
                       capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:]


def test_confidence_scales_every_gradient_but_means2D(oracle):
    sc = small_scene(80, 2, False, False, 3, False)
    cam = synthetic.look_at_camera((3.0, 0.5, 1.0), 40, 32, FoVx=0.9)
    bg = torch.zeros(3)
    g = torch.Generator().manual_seed(0)
    dL = (torch.randn((3, 32, 40), generator=g), torch.randn((1, 32, 40), generator=g), torch.randn((1, 32, 40), generator=g))
    conf = torch.rand((80, 1), generator=g)
    a = fsgs_run(oracle.FsgsRasterizer, oracle.FsgsSettings, sc, cam, bg, torch.device("cpu"), *dL)
    b = fsgs_run(oracle.FsgsRasterizer, oracle.FsgsSettings, sc, cam, bg, torch.device("cpu"), *dL, confidence=conf)
    assert torch.equal(a["color"], b["color"]) and torch.equal(a["grads"]["means2D"], b["grads"]["means2D"])
    for k in LEAVES:
        c = conf if a["grads"][k].dim() == 2 else conf[..., None]
        assert torch.allclose(b["grads"][k], a["grads"][k] * c, rtol=0, atol=0), k


def test_fsgs_settings_and_errors(oracle):
    assert dgr_fsgs.GaussianRasterizationSettings._fields[-1] == "confidence" and len(dgr_fsgs.GaussianRasterizationSettings._fields) == 13
    sc = small_scene(20, 1, False, False, 3, False)
    cam = synthetic.look_at_camera((3.0, 0.5, 1.0), 32, 32, FoVx=0.9)
    rs = oracle.FsgsSettings(32, 32, cam.tanfovx, cam.tanfovy, torch.zeros(3), 1.0, cam.world_view_transform,
                             cam.full_proj_transform, 3, cam.camera_center, False, False, torch.ones((20, 1)))
    r = oracle.FsgsRasterizer(rs)
    with pytest.raises(Exception):
        r(means3D=sc["means3D"], means2D=None, opacities=sc["opacities"], scales=sc["scales"], rotations=sc["rotations"])
    with pytest.raises(Exception):
        r(means3D=sc["means3D"], means2D=None, opacities=sc["opacities"], shs=sc["shs"], scales=sc["scales"])
    assert r.markVisible(sc["means3D"]).dtype == torch.bool
