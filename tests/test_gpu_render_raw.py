"""gsplat_amd.render_raw.render - the reference's render() (LGDWT-GS/gaussian_renderer/__init__.py:18-128) for its own GaussianModel
with the six RAW leaf tensors handed to the library - against the drop-in render() that activates them with torch (exp / normalize /
sigmoid / cat) first: same image, same radii, the gradients on the six leaves within 1e-4 of their largest entry, and the
reference's densification statistic from `viewspace_points.grad`."""
import pytest
import torch

from gsplat_amd import synthetic
from gsplat_amd.dropin import _PIPE, DropInLoop, DropInModel, render
from gsplat_amd.trainer import camera_to

pytestmark = pytest.mark.gpu


def leaves(pc):
    return dict(xyz=pc._xyz, f_dc=pc._features_dc, f_rest=pc._features_rest, opacity=pc._opacity, scaling=pc._scaling,
                rotation=pc._rotation)


@pytest.mark.parametrize("P,W,H,deg", [(20000, 480, 320, 3), (3000, 131, 75, 1)])
def test_raw_row_render_equals_the_drop_in_render(hip, P, W, H, deg):
    from gsplat_amd.render_raw import render as render_raw
    dev = torch.device("cuda")
    sc = synthetic.trained_like(P, seed=3, sh_degree=deg)
    cam = camera_to(synthetic.orbit_cameras(W, H)[5], dev)
    bg = torch.tensor([0.1, 0.3, 0.2], device=dev)
    g = torch.Generator().manual_seed(0)
    cot = torch.randn((3, H, W), generator=g).to(dev)
    cot_d = (torch.randn((1, H, W), generator=g) * 0.2).to(dev)
    out = {}
    for name in ("torch", "raw"):
        pc = DropInModel(sc, dev)
        pkg = render(cam, pc, bg) if name == "torch" else render_raw(cam, pc, _PIPE, bg)
        assert set(pkg) == {"render", "viewspace_points", "visibility_filter", "radii", "depth"}
        ((pkg["render"] * cot).sum() + (pkg["depth"] * cot_d).sum()).backward()
        vis = pkg["visibility_filter"].squeeze(1)
        stat = torch.norm(pkg["viewspace_points"].grad[vis, :2], dim=-1)      # gaussian_model.py:472
        out[name] = dict(image=pkg["render"].detach(), radii=pkg["radii"], depth=pkg["depth"].detach(), stat=stat, vis=vis,
                         grads={k: v.grad.clone() for k, v in leaves(pc).items()})
    a, b = out["torch"], out["raw"]
    assert torch.equal(a["radii"], b["radii"]) and torch.equal(a["vis"], b["vis"])
    assert float((a["image"] - b["image"]).abs().max()) <= 1e-5 and float((a["depth"] - b["depth"]).abs().max()) <= 1e-5
    for k in a["grads"]:
        ga, gb = a["grads"][k], b["grads"][k]
        assert ga.shape == gb.shape
        err = float((ga - gb).abs().max()) / max(float(ga.abs().max()), 1e-20)
        print("dL/d%-9s rel err %.2e" % (k, err))
        assert err <= 1e-4, (k, err)
    assert float((a["stat"] - b["stat"]).abs().max()) <= 1e-4 * float(a["stat"].abs().max())


def test_drop_in_loop_with_the_raw_row_render_trains(hip):
    dev = torch.device("cuda")
    sc = synthetic.trained_like(30000, seed=1)
    cams = [camera_to(c, dev) for c in synthetic.orbit_cameras(320, 240)[:3]]
    g = torch.Generator().manual_seed(2)
    gts = [torch.rand((3, 240, 320), generator=g).to(dev) for _ in cams]
    loops = {kind: DropInLoop(sc, cams, gts, dev, dwt=True, patch=True, optimizer="fused", use_camera_key=True,
                              fused_criterion=True, raw_render=kind == "raw") for kind in ("torch", "raw")}
    losses = {k: [lp.iteration(j % 3) for j in range(9)] for k, lp in loops.items()}
    for k in losses:
        assert losses[k][-1] < losses[k][0]
    assert all(abs(x - y) <= 2e-3 * abs(x) for x, y in zip(losses["torch"], losses["raw"])), losses
    pa, pb = loops["torch"].pc, loops["raw"].pc
    assert torch.equal(pa.denom, pb.denom) and torch.equal(pa.max_radii2D, pb.max_radii2D)
    assert float((pa.xyz_gradient_accum - pb.xyz_gradient_accum).abs().max()) <= 1e-3 * float(pa.xyz_gradient_accum.abs().max())
