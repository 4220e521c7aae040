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
        assert set(pkg) - {"render_unclamped"} == {"render", "viewspace_points", "visibility_filter", "radii", "depth"}
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


def _raw_forward_backward(backend, rs, pc, cot, cot_d, split):
    """The node of render_raw by hand, with the SH rows either as ONE [P,16,3] array (torch.cat copy) or as the model's two
    tensors (GsGaussians.shs_rest / GsStepState.grad_out_rest)."""
    from gsplat_amd.capi import GsStepState
    dev, P = pc._xyz.device, int(pc._xyz.shape[0])
    f32 = dict(dtype=torch.float32, device=dev)
    e = torch.Tensor([])
    with torch.no_grad():
        f_dc, f_rest = pc._features_dc.detach(), pc._features_rest.detach()
        sh = f_dc if split else torch.cat((f_dc, f_rest), dim=1).contiguous()
        xyz, op, sc, rot = pc._xyz.detach(), pc._opacity.detach(), pc._scaling.detach(), pc._rotation.detach()
        backend.raw_activations = True
        backend.sh_rest = f_rest if split else None
        out = backend.rasterize_gaussians(rs.bg, xyz, e, op, sc, rot, rs.scale_modifier, e, rs.viewmatrix, rs.projmatrix,
                                          rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos,
                                          rs.prefiltered, rs.antialiasing, rs.debug)
        num_rendered, color, radii, geom, binning, img, invdepth = out
        gx, gop, gsc, grot = (torch.empty((P, n), **f32) for n in (3, 1, 3, 4))
        gsh = [torch.empty((P, 1, 3), **f32), torch.empty((P, 15, 3), **f32)] if split else [torch.empty((P, 16, 3), **f32)]
        stats = torch.zeros((3, P), **f32)
        st = GsStepState()
        st.xyz, st.features, st.opacity, st.scaling, st.rotation = (t.data_ptr() for t in (xyz, sh, op, sc, rot))
        for k, t in enumerate((gx, gsh[0], gop, gsc, grot)):
            st.grad_out[k] = t.data_ptr()
            st.step[k] = 1
        if split:
            st.grad_out_rest = gsh[1].data_ptr()
        st.beta1, st.beta2, st.eps = 0.9, 0.999, 1e-15
        st.xyz_gradient_accum, st.denom, st.max_radii2D = (stats[k].data_ptr() for k in range(3))
        backend._raw_backward = True
        backend.sh_rest = f_rest if split else None
        backend.fused_step = st
        backend.rasterize_gaussians_backward(rs.bg, xyz, radii, e, op, sc, rot, rs.scale_modifier, e, rs.viewmatrix, rs.projmatrix,
                                             rs.tanfovx, rs.tanfovy, cot, cot_d, sh, rs.sh_degree, rs.campos, geom, num_rendered,
                                             binning, img, rs.antialiasing, rs.debug)
        torch.cuda.synchronize()
    gsh_all = torch.cat(gsh, dim=1)
    return dict(color=color, radii=radii, invdepth=invdepth, gx=gx, gsh=gsh_all, gop=gop, gsc=gsc, grot=grot, stats=stats)


@pytest.mark.parametrize("P,W,H,deg", [(20000, 480, 320, 3), (3001, 131, 75, 2), (700, 64, 48, 0), (1_000_000, 1920, 1080, 3)])
def test_split_sh_rows_are_bit_identical_to_the_one_row_layout(hip, P, W, H, deg):
    """GsGaussians.shs_rest + GsStepState.grad_out_rest (the model's _features_dc / _features_rest read and written in place)
    against the same call on the concatenated [P,16,3] rows: every output and every gradient bit for bit."""
    import math
    from diff_gaussian_rasterization import GaussianRasterizationSettings, _RasterizeGaussians
    dev = torch.device("cuda")
    knn = None
    if P > 100000:   # (BASELINE C3 size: the scene's scales from the product's own kNN, as bench.py builds it)
        from simple_knn._C import distCUDA2
        knn = lambda x: distCUDA2(x.to(dev)).cpu()  # noqa: E731
    sc = synthetic.trained_like(P, seed=11, sh_degree=deg, knn=knn)
    cam = camera_to(synthetic.orbit_cameras(W, H)[2], dev)
    pc = DropInModel(sc, dev)
    rs = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=torch.tensor([0.2, 0.1, 0.4], device=dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=pc.active_sh_degree, campos=cam.camera_center, prefiltered=False,
        debug=False, antialiasing=False)
    backend = _RasterizeGaussians._impl.backend
    g = torch.Generator().manual_seed(4)
    cot = torch.randn((3, H, W), generator=g).to(dev)
    cot_d = (torch.randn((1, H, W), generator=g) * 0.2).to(dev)
    a = _raw_forward_backward(backend, rs, pc, cot, cot_d, split=False)
    b = _raw_forward_backward(backend, rs, pc, cot, cot_d, split=True)
    assert int((a["radii"] > 0).sum()) > P // 10
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert float(a["gsh"].abs().max()) > 0


def test_split_sh_rows_are_refused_where_they_are_not_served(hip):
    """gs_backward (one dL_dsh array) and the Adam form of gs_backward_step return GS_E_UNSUPPORTED / GS_E_SHAPE; the python
    backend refuses them before the call."""
    dev = torch.device("cuda")
    from diff_gaussian_rasterization import _RasterizeGaussians
    backend = _RasterizeGaussians._impl.backend
    sc = synthetic.trained_like(500, seed=1)
    pc = DropInModel(sc, dev)
    cam = camera_to(synthetic.orbit_cameras(64, 48)[0], dev)
    import math
    e = torch.Tensor([])
    backend.sh_rest = pc._features_rest.detach()
    with pytest.raises(RuntimeError, match="raw activations"):
        backend.rasterize_gaussians(torch.zeros(3, device=dev), pc._xyz.detach(), e, torch.sigmoid(pc._opacity.detach()),
                                    torch.exp(pc._scaling.detach()), pc._rotation.detach(), 1.0, e, cam.world_view_transform,
                                    cam.full_proj_transform, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), 48, 64,
                                    pc._features_dc.detach(), 3, cam.camera_center, False, False, False)
    assert backend.sh_rest is None


def test_an_empty_model_renders_nothing_and_leaves_no_request_armed(hip):
    """P = 0 (rasterize_points.cu:88: the reference returns zero images and empty buffers): render_raw returns the same, its
    backward returns empty gradients, and the one-shot requests it made of the backend (raw activations, split SH rows, camera
    key) are gone - the next, ordinary render is not affected."""
    from gsplat_amd.render_raw import render as render_raw
    from diff_gaussian_rasterization import _RasterizeGaussians
    dev = torch.device("cuda")
    cam = camera_to(synthetic.orbit_cameras(96, 64)[0], dev)
    bg = torch.tensor([0.5, 0.25, 0.125], device=dev)

    class Empty:
        active_sh_degree = 3

    pc = Empty()
    for name, shape in (("_xyz", (0, 3)), ("_features_dc", (0, 1, 3)), ("_features_rest", (0, 15, 3)), ("_opacity", (0, 1)),
                        ("_scaling", (0, 3)), ("_rotation", (0, 4))):
        setattr(pc, name, torch.nn.Parameter(torch.zeros(shape, device=dev)))
    pkg = render_raw(cam, pc, _PIPE, bg, camera_key="empty")
    assert pkg["render"].shape == (3, 64, 96) and float(pkg["render"].abs().max()) == 0.0
    assert pkg["radii"].numel() == 0 and pkg["visibility_filter"].numel() == 0
    (pkg["render"].sum() + pkg["depth"].sum()).backward()
    assert pc._xyz.grad is None or pc._xyz.grad.numel() == 0
    backend = _RasterizeGaussians._impl.backend
    assert not backend.raw_activations and backend.sh_rest is None and backend.camera_key is None and not backend._raw_backward
    # an ordinary render right behind it: activated values, one SH array - must not be read as raw rows
    full = DropInModel(synthetic.trained_like(2000, seed=2), dev)
    a = render(cam, full, bg)["render"]
    b = render(cam, full, bg)["render"]
    assert torch.equal(a, b) and float(a.max()) > 0


def test_an_evaluation_render_on_raw_rows_between_a_forward_and_its_backward(hip):
    """train.py renders test views under no_grad while a training graph may still be alive: a raw-row forward without a backward
    of its own must not be taken for the forward of the NEXT backward the backend sees."""
    from gsplat_amd.render_raw import render as render_raw
    dev = torch.device("cuda")
    sc = synthetic.trained_like(5000, seed=4)
    cams = [camera_to(c, dev) for c in synthetic.orbit_cameras(160, 120)[:2]]
    bg = torch.zeros(3, device=dev)
    grads = []
    for evaluate in (False, True):
        pc = DropInModel(sc, dev)
        pkg = render(cams[0], pc, bg)                      # the drop-in render on activated values: a training forward
        if evaluate:
            with torch.no_grad():
                img = render_raw(cams[1], pc, _PIPE, bg)["render"]
            assert float(img.max()) > 0
        pkg["render"].square().sum().backward()
        grads.append({k: v.grad.clone() for k, v in leaves(pc).items()})
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]), k
