"""Oracle (fp32 restatement of the CUDA kernels) vs the independent dense float64 autograd model."""
import math

import pytest
import torch

import dense_reference
from gsplat_amd import synthetic
from helpers import run_scene


def small_scene(P, seed, precomp_color=False, precomp_cov=False, sh_degree=3, big=False):
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand((P, 3), generator=g) * 2.0 - 1.0
    scales = torch.exp(torch.log(torch.tensor(0.25 if big else 0.12)) + 0.5 * torch.randn((P, 3), generator=g))
    q = torch.randn((P, 4), generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    op = torch.sigmoid(1.5 * torch.randn((P, 1), generator=g)).clamp(0.02, 0.98)
    sc = dict(means3D=xyz, opacities=op, sh_degree=sh_degree)
    if precomp_color:
        sc["colors_precomp"] = torch.rand((P, 3), generator=g)
    else:
        sh = torch.zeros((P, 16, 3))
        sh[:, 0] = torch.randn((P, 3), generator=g) * 0.8
        sh[:, 1:] = torch.randn((P, 15, 3), generator=g) * 0.3
        sc["shs"] = sh
    if precomp_cov:
        R = dense_reference.quat_to_rot(q.double())
        S = torch.diag_embed(scales.double())
        Sig = (R @ S @ S @ R.transpose(1, 2)).float()
        sc["cov3D_precomp"] = torch.stack([Sig[:, 0, 0], Sig[:, 0, 1], Sig[:, 0, 2], Sig[:, 1, 1], Sig[:, 1, 2],
                                           Sig[:, 2, 2]], dim=1).contiguous()
    else:
        sc["scales"], sc["rotations"] = scales, q
    return sc


def dense_run(scene, cam, bg, antialiasing, dL_dcolor, dL_dinvdepth):
    d = {}
    leaves = {}
    for k, v in scene.items():
        if torch.is_tensor(v):
            leaves[k] = v.double().clone().requires_grad_(True)
            d[k] = leaves[k]
        else:
            d[k] = v
    P = scene["means3D"].shape[0]
    leaves["ndc_probe"] = torch.zeros((P, 2), dtype=torch.float64, requires_grad=True)
    d["ndc_probe"] = leaves["ndc_probe"]
    out = dense_reference.render(d, cam, bg, antialiasing)
    loss = (out["color"] * dL_dcolor.double()).sum()
    if dL_dinvdepth is not None:
        loss = loss + (out["invdepth"] * dL_dinvdepth.double()).sum()
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return out, grads


CASES = [
    dict(P=120, seed=1, W=48, H=40, aa=False, bg=(0.0, 0.0, 0.0), eye=(3.2, 1.0, 1.5)),
    dict(P=160, seed=2, W=37, H=53, aa=True, bg=(1.0, 0.5, 0.25), eye=(-2.5, 2.8, -0.7)),
    dict(P=90, seed=3, W=64, H=32, aa=False, bg=(0.2, 0.9, 0.1), eye=(0.6, -1.4, 0.4), big=True),  # camera inside the cloud: culling + clamping
    dict(P=100, seed=4, W=40, H=40, aa=True, bg=(0.0, 0.0, 0.0), eye=(3.0, 0.2, 2.0), precomp_color=True, precomp_cov=True),
    dict(P=100, seed=5, W=33, H=47, aa=False, bg=(0.3, 0.3, 0.3), eye=(2.0, 2.0, 2.0), sh_degree=1),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "P%d_%dx%d_s%d" % (c["P"], c["W"], c["H"], c["seed"]))
def test_oracle_matches_dense_float64(oracle, case):
    sc = small_scene(case["P"], case["seed"], case.get("precomp_color", False), case.get("precomp_cov", False),
                     case.get("sh_degree", 3), case.get("big", False))
    cam = synthetic.look_at_camera(case["eye"], case["W"], case["H"], FoVx=0.9)
    bg = torch.tensor(case["bg"])
    g = torch.Generator().manual_seed(100 + case["seed"])
    dL_dcolor = torch.randn((3, case["H"], case["W"]), generator=g)
    dL_dinv = torch.randn((1, case["H"], case["W"]), generator=g) * 0.3
    o = run_scene(oracle.Rasterizer, oracle.Settings, sc, cam, torch.device("cpu"), bg=bg, antialiasing=case["aa"],
                  dL_dcolor=dL_dcolor, dL_dinvdepth=dL_dinv)
    dn, dg = dense_run(sc, cam, bg, case["aa"], dL_dcolor, dL_dinv)

    assert torch.equal(o["radii"].long(), dn["radii"].long()), "radii differ"
    assert int((o["radii"] > 0).sum()) > case["P"] // 4
    err = (o["color"].double() - dn["color"]).abs().max().item()
    assert err < 2e-5, "colour max abs err %.3e" % err
    err = (o["invdepth"].double() - dn["invdepth"]).abs().max().item()
    assert err < 2e-5 * max(1.0, dn["invdepth"].abs().max().item()), "invdepth err %.3e" % err

    def chk(name, a, b, tol=2e-4):
        a, b = a.double(), b.double()
        scale = max(b.abs().max().item(), 1e-12)
        e = (a - b).abs().max().item() / scale
        assert e < tol, "%s: rel err %.3e (scale %.3e)" % (name, e, scale)

    for k in ("means3D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp"):
        if k in o["grads"]:
            chk("dL_d" + k, o["grads"][k], dg[k])
    chk("dL_dmeans2D", o["grads"]["means2D"][:, :2], dg["ndc_probe"])
    assert float(o["grads"]["means2D"][:, 2].abs().max()) == 0.0
