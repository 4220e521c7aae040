"""GPU parity of the fused RGB+NIR pass: ONE 4-channel pass vs the reference's TWO
3-channel passes (render + render_nir, mult-dwtgs/gaussian_renderer/__init__.py:18-149,151-258), the second
with colors_precomp = nir.repeat(1, 3) and channel 0 kept.  Checked against both the HIP two-pass result and the
oracle's two passes."""
import pytest
import torch

import diff_gaussian_rasterization as dgr
from gsplat_amd import synthetic
from gsplat_amd.nir import GaussianRasterizerX, nir_colors
from helpers import settings_for

pytestmark = pytest.mark.gpu
TOL = 1e-4


def leaves(sc, device):
    names = ("means3D", "opacities", "shs", "scales", "rotations")
    return {k: sc[k].detach().clone().to(device).requires_grad_(True) for k in names}


def two_pass(Rast, Settings, sc, cam, bg, nir, device, dL_rgb, dL_nir, aa):
    p = leaves(sc, device)
    nirp = nir.detach().clone().to(device).requires_grad_(True)
    rs = settings_for(Settings, cam, bg, sc.get("sh_degree", 0), device, aa, 1.0)
    rast = Rast(raster_settings=rs)
    m2a = torch.zeros_like(p["means3D"], requires_grad=True)
    rgb, radii, invd = rast(means3D=p["means3D"], means2D=m2a, opacities=p["opacities"], shs=p["shs"],
                            scales=p["scales"], rotations=p["rotations"])
    m2b = torch.zeros_like(p["means3D"], requires_grad=True)
    img3, _, _ = rast(means3D=p["means3D"], means2D=m2b, opacities=p["opacities"], shs=None,
                      colors_precomp=nirp[:, None].repeat(1, 3), scales=p["scales"], rotations=p["rotations"])
    nir_img = img3[0:1]
    ((rgb * dL_rgb.to(device)).sum() + (nir_img * dL_nir.to(device)).sum()).backward()
    g = {k: v.grad.detach().cpu() for k, v in p.items()}
    g["nir"] = nirp.grad.detach().cpu()
    g["means2D"] = (m2a.grad + m2b.grad).detach().cpu()
    return rgb.detach().cpu(), nir_img.detach().cpu(), radii.cpu(), g


def fused(sc, cam, bg, nir, dL_rgb, dL_nir, aa):
    device = torch.device("cuda")
    p = leaves(sc, device)
    nirp = nir.detach().clone().to(device).requires_grad_(True)
    rs = settings_for(dgr.GaussianRasterizationSettings, cam, bg, sc.get("sh_degree", 0), device, aa, 1.0)
    m2 = torch.zeros_like(p["means3D"], requires_grad=True)
    rgb, radii, invd, nir_img = GaussianRasterizerX(rs)(means3D=p["means3D"], means2D=m2, opacities=p["opacities"],
                                                       extra=nirp, shs=p["shs"], scales=p["scales"],
                                                       rotations=p["rotations"])
    ((rgb * dL_rgb.to(device)).sum() + (nir_img * dL_nir.to(device)).sum()).backward()
    g = {k: v.grad.detach().cpu() for k, v in p.items()}
    g["nir"] = nirp.grad.detach().cpu()
    g["means2D"] = m2.grad.detach().cpu()
    return rgb.detach().cpu(), nir_img.detach().cpu(), radii.cpu(), g


@pytest.mark.parametrize("kind,P,W,H,deg,aa,bgv", [("trained", 8000, 320, 240, 3, False, (0.0, 0.0, 0.0)),
                                                  ("trained", 20000, 400, 400, 1, True, (0.7, 0.2, 0.4)),
                                                  ("init", 10000, 400, 400, 0, False, (1.0, 1.0, 1.0))])
def test_fused_pass_equals_two_reference_passes(hip, oracle, kind, P, W, H, deg, aa, bgv):
    gen = synthetic.init_like if kind == "init" else synthetic.trained_like
    sc = gen(P, seed=4, sh_degree=deg)
    cam = synthetic.orbit_cameras(W, H)[2]
    bg = torch.tensor(bgv)
    g = torch.Generator().manual_seed(9)
    nir = nir_colors(torch.sigmoid(torch.randn((P, 1, 1), generator=g)), torch.tensor(1.7))
    dL_rgb = torch.randn((3, H, W), generator=g)
    dL_nir = torch.randn((1, H, W), generator=g)
    f_rgb, f_nir, f_radii, f_g = fused(sc, cam, bg, nir, dL_rgb, dL_nir, aa)
    h_rgb, h_nir, h_radii, h_g = two_pass(dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, sc, cam, bg, nir,
                                          torch.device("cuda"), dL_rgb, dL_nir, aa)
    # same kernels, same lists, same blend order: colour is bit-identical to the HIP two-pass result, the 4th
    # channel up to the contraction of its final X + T*bg
    assert torch.equal(f_rgb, h_rgb) and torch.equal(f_radii, h_radii)
    assert float((f_nir - h_nir).abs().max()) <= 1e-6
    from helpers import check_grads
    tag = "nir_%s_%d_%dx%d" % (kind, P, W, H)
    # (the fused pass rounds the sum of both passes' contributions once, the two passes round each and add)
    check_grads({k: v.cpu() for k, v in f_g.items()}, {k: v.cpu() for k, v in h_g.items()}, tag + "_vs_hip_two_pass", scene=sc)
    o_rgb, o_nir, o_radii, o_g = two_pass(oracle.Rasterizer, oracle.Settings, sc, cam, bg, nir, torch.device("cpu"),
                                          dL_rgb, dL_nir, aa)
    assert torch.equal(f_radii, o_radii)
    bad = ((f_nir - o_nir).abs() > TOL).sum() + ((f_rgb - o_rgb).abs().amax(0) > TOL * max(1.0, float(o_rgb.abs().max()))).sum()
    assert int(bad) <= max(2, W * H // 20000)
    keep = (((f_nir - o_nir).abs()[0] <= 0.2 * TOL) & ((f_rgb - o_rgb).abs().amax(0) <= 0.2 * TOL)).float()
    if float(keep.min()) == 0.0:  # a threshold pixel took the other branch: compare gradients without it
        dL_rgb, dL_nir = dL_rgb * keep, dL_nir * keep
        _, _, _, f_g = fused(sc, cam, bg, nir, dL_rgb, dL_nir, aa)
        _, _, _, o_g = two_pass(oracle.Rasterizer, oracle.Settings, sc, cam, bg, nir, torch.device("cpu"), dL_rgb,
                                dL_nir, aa)
    check_grads({k: v.cpu() for k, v in f_g.items()}, o_g, tag + "_vs_oracle_two_pass", scene=sc)


def test_fused_pass_at_c5_size_against_the_oracle_two_pass(hip, oracle):
    """BASELINE configs[4] geometry - 1 M Gaussians, 1920x1080, SH degree 3 - through the fused 4-channel pass, against
    the reference's two passes on the CPU oracle (about 10 s): radii equal, both images at 1e-4 up to a bounded number
    of threshold pixels, every gradient at 1e-4 of its tensor's maximum - dL_dscales / dL_drotations against the oracle's
    double evaluation of the reference's chain (conftest._gpu_parity_uses_the_exact_chain; rounds 1-3 bounded them at 1e-2
    here because the oracle's fp32 formula is ~1e-3 from the float64 result at this size)."""
    P, W, H = 1_000_000, 1920, 1080
    from simple_knn._C import distCUDA2
    sc = synthetic.trained_like(P, seed=0, sh_degree=3, knn=lambda x: distCUDA2(x.cuda()).cpu())
    cam = synthetic.orbit_cameras(W, H)[5]
    bg = torch.zeros(3)
    g = torch.Generator().manual_seed(9)
    nir = nir_colors(torch.sigmoid(torch.randn((P, 1, 1), generator=g)), torch.tensor(1.3))
    dL_rgb = torch.randn((3, H, W), generator=g)
    dL_nir = torch.randn((1, H, W), generator=g)
    f_rgb, f_nir, f_radii, f_g = fused(sc, cam, bg, nir, dL_rgb, dL_nir, False)
    o_rgb, o_nir, o_radii, o_g = two_pass(oracle.Rasterizer, oracle.Settings, sc, cam, bg, nir, torch.device("cpu"),
                                          dL_rgb, dL_nir, False)
    assert torch.equal(f_radii, o_radii)
    bad = ((f_nir - o_nir).abs() > TOL).sum() + ((f_rgb - o_rgb).abs().amax(0) > TOL * max(1.0, float(o_rgb.abs().max()))).sum()
    print("pixels beyond 1e-4:", int(bad), "of", W * H)
    assert int(bad) <= W * H // 20000
    keep = (((f_nir - o_nir).abs()[0] <= 0.2 * TOL) & ((f_rgb - o_rgb).abs().amax(0) <= 0.2 * TOL)).float()
    if float(keep.min()) == 0.0:  # threshold pixels took the other branch somewhere: compare gradients without them
        dL_rgb, dL_nir = dL_rgb * keep, dL_nir * keep
        _, _, _, f_g = fused(sc, cam, bg, nir, dL_rgb, dL_nir, False)
        _, _, _, o_g = two_pass(oracle.Rasterizer, oracle.Settings, sc, cam, bg, nir, torch.device("cpu"), dL_rgb, dL_nir,
                                False)
    from helpers import check_grads
    check_grads({k: v.cpu() for k, v in f_g.items()}, o_g, "nir_c5_size_vs_oracle_two_pass")
    # ... and on the lists the C5 bench times: the camera's second visit renders from depth-limited, region-binned lists
    # (a fifth of the instances) - same images, and every gradient again within 1e-4 of the oracle's two passes
    old = (hip.tile_cull, hip.depth_limit_on)
    hip._cam_cache.clear()
    try:
        hip.tile_cull, hip.depth_limit_on = True, True
        used0, failed0 = hip.depth_limit_stats["used"], hip.depth_limit_stats["failed"]
        fused(sc, cam, bg, nir, dL_rgb, dL_nir, False)                       # first visit: measures where every tile stops
        l_rgb, l_nir, l_radii, l_g = fused(sc, cam, bg, nir, dL_rgb, dL_nir, False)
        assert hip.depth_limit_stats["used"] - used0 == 1 and hip.depth_limit_stats["failed"] == failed0
        assert torch.equal(l_rgb, f_rgb) and torch.equal(l_nir, f_nir) and torch.equal(l_radii, f_radii)
        check_grads({k: v.cpu() for k, v in l_g.items()}, o_g, "nir_c5_size_limited_lists_vs_oracle_two_pass")
        for k in f_g:   # the same pairs, float64 sums: the limited run IS the un-limited one
            x, y = f_g[k].double().cpu(), l_g[k].double().cpu()
            assert float((x - y).abs().max()) <= 1e-6 * max(1e-12, float(x.abs().max())), k
    finally:
        hip.tile_cull, hip.depth_limit_on = old
        hip._cam_cache.clear()


def test_four_channel_pass_on_depth_limited_lists(hip):
    """The per-camera depth limits (csrc/gs_tilecull.h) under the fused RGB + NIR pass: the second and third visit of a
    camera render from cut, region-binned lists - all four channels, radii and every gradient the bits of the un-limited
    pass (the 4th channel's blend stops where the colour's does: same T test, forward.cu:326-328) - and stale limits are
    detected and the view rendered again."""
    P, W, H = 40000, 640, 480
    sc = synthetic.trained_like(P, seed=5, sh_degree=2)
    cam = synthetic.orbit_cameras(W, H)[7]
    cam = cam._replace(world_view_transform=cam.world_view_transform.cuda(), full_proj_transform=cam.full_proj_transform.cuda(),
                       camera_center=cam.camera_center.cuda())
    bg = torch.tensor([0.1, 0.2, 0.3])
    g = torch.Generator().manual_seed(3)
    nir = torch.rand((P,), generator=g)
    dL_rgb, dL_nir = torch.randn((3, H, W), generator=g), torch.randn((1, H, W), generator=g)
    old = (hip.tile_cull, hip.depth_limit_on)
    hip._cam_cache.clear()
    try:
        hip.tile_cull, hip.depth_limit_on = True, False
        ref = fused(sc, cam, bg, nir, dL_rgb, dL_nir, False)
        hip.depth_limit_on = True
        used0, failed0 = hip.depth_limit_stats["used"], hip.depth_limit_stats["failed"]
        first = fused(sc, cam, bg, nir, dL_rgb, dL_nir, False)     # measures the stop depths
        second = fused(sc, cam, bg, nir, dL_rgb, dL_nir, False)    # limited
        third = fused(sc, cam, bg, nir, dL_rgb, dL_nir, False)
        assert hip.depth_limit_stats["used"] - used0 == 2 and hip.depth_limit_stats["failed"] == failed0
        for run in (first, second, third):
            assert torch.equal(run[0], ref[0]) and torch.equal(run[1], ref[1]) and torch.equal(run[2], ref[2])
            for k in ref[3]:
                x, y = ref[3][k].double(), run[3][k].double()
                # (the same pairs summed into float64 rows: which tiles' totals arrive first no longer shows; rounds 1-3: 5e-4)
                assert float((x - y).abs().max()) <= 1e-6 * max(1e-12, float(x.abs().max())), k
        faint = dict(sc, opacities=sc["opacities"] * 0.3)            # tiles now saturate far deeper than the limits allow
        hip.depth_limit_on = False
        ref_f = fused(faint, cam, bg, nir, dL_rgb, dL_nir, False)
        hip.depth_limit_on = True
        got = fused(faint, cam, bg, nir, dL_rgb, dL_nir, False)
        assert hip.depth_limit_stats["failed"] == failed0 + 1
        assert torch.equal(got[0], ref_f[0]) and torch.equal(got[1], ref_f[1])
    finally:
        hip.tile_cull, hip.depth_limit_on = old
        hip._cam_cache.clear()


def test_extra_channel_argument_errors(hip):
    dev = torch.device("cuda")
    sc = synthetic.trained_like(100, seed=1, sh_degree=0)
    cam = synthetic.orbit_cameras(64, 64)[0]
    rs = settings_for(dgr.GaussianRasterizationSettings, cam, torch.zeros(3), 0, dev, False, 1.0)
    p = {k: sc[k].to(dev) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    with pytest.raises(Exception):
        GaussianRasterizerX(rs)(means3D=p["means3D"], means2D=None, opacities=p["opacities"], extra=torch.zeros(7, device=dev),
                                shs=p["shs"], scales=p["scales"], rotations=p["rotations"])
    with pytest.raises(Exception):
        GaussianRasterizerX(rs)(means3D=p["means3D"], means2D=None, opacities=p["opacities"],
                                extra=torch.zeros(100, device=dev), scales=p["scales"], rotations=p["rotations"])
