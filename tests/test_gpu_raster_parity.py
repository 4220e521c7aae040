"""GPU parity: libgsplat_hip.so (through the C ABI) vs the CPU oracle on the same seeded inputs.

Bit-exact: radii, tiles_touched, point_offsets, num_rendered, sorted keys, point_list, ranges and
the whole per-Gaussian geometry state (depth, pixel mean, conic, opacity, colour, cov3D, clamp flags):
the per-Gaussian kernels are built without FMA contraction and follow the oracle's evaluation order.
Tolerance 1e-4 (relative to the tensor's max magnitude, fp32): rendered colour / inverse depth /
final_T and all gradients - the blend kernels use FMA + v_exp_f32 and sum in a different order.
A pixel whose alpha or transmittance sits within rounding distance of a hard threshold
(alpha < 1/255, T < 1e-4) may legitimately flip; such pixels are counted and bounded separately.
The same holds for gradients: one flipped decision changes that pixel's transmittance chain by a whole
alpha quantum (1/255), i.e. ~4e-3 of everything the pixel back-propagates.  (Measured on the CPU side alone:
the oracle compiled with and without FMA contraction - the freedom nvcc has with the reference - differs by
4e-4 of max|dL_dscales| on the init-10k/400x400 scene because of ONE such pixel.)  The gradient tests
therefore find the flipped pixels from the two forward states (`flip_mask`), bound their number, zero the
image cotangent there for BOTH implementations and compare everything else at the tight tolerance.
"""
import math

import pytest
import torch

import diff_gaussian_rasterization as dgr
from gsplat_amd import synthetic
from helpers import run_scene
from test_oracle_dense import small_scene

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(autouse=True)
def reference_lists(hip):
    """This module pins the binning state bit for bit, so it runs with the reference's bounding-square
    instance lists (GsView.tile_cull = 0); tests/test_gpu_tilecull.py covers the culled lists."""
    old = hip.tile_cull
    hip.tile_cull = False
    yield
    hip.tile_cull = old


def forward_state(backend, scene, cam, device, bg, antialiasing):
    def dev(t):
        return None if t is None else t.to(device)
    e = torch.empty(0)
    sh = dev(scene.get("shs"))
    args = (bg.to(device), dev(scene["means3D"]), dev(scene.get("colors_precomp")) if scene.get("colors_precomp") is not None else e,
            dev(scene["opacities"]), dev(scene.get("scales")) if scene.get("scales") is not None else e,
            dev(scene.get("rotations")) if scene.get("rotations") is not None else e, scene.get("scale_modifier", 1.0),
            dev(scene.get("cov3D_precomp")) if scene.get("cov3D_precomp") is not None else e,
            cam.world_view_transform.to(device), cam.full_proj_transform.to(device), cam.tanfovx, cam.tanfovy,
            cam.image_height, cam.image_width, sh if sh is not None else e, scene.get("sh_degree", 0),
            cam.camera_center.to(device), False, antialiasing, False)
    R, color, radii, geom, binning, img, invd = backend.rasterize_gaussians(*args)
    P = scene["means3D"].shape[0]
    st = backend.export_state(P, cam.image_width, cam.image_height, R, geom, binning, img)
    st = {k: v.cpu() for k, v in st.items()}
    st.update(num_rendered=R, color=color.cpu(), radii=radii.cpu(), invdepth=invd.cpu())
    return st


FLIPS = {}   # name -> measured counts (written to gpurun_out/flip_counts.json by conftest at session end)


def flip_bound(npix):
    """Pixels whose last contributor / threshold decision may differ from the oracle's (log2-domain alpha + FMA in
    csrc/gs_blend.h against the oracle's expf form: a pixel within rounding of alpha = 1/255 or T = 1e-4 takes the other branch).
    Measured (round 4, profiles/r04_parity_fullsize_c*.json): 2 of 0.64 M pixels at C2, 6 of 2.07 M at C3, 10 at C4 - about 3-5 per
    million.  The bar is 4x that rate, with a floor for small images (where a handful is the measured worst: round 5,
    profiles/r05_flip_counts.json)."""
    return max(4, npix // 80000)


BINNING_STATE = ("tiles_touched", "point_offsets", "keys_sorted", "point_list", "ranges")


def last_contributor_id(st, W, H):
    """[H, W] int64: Gaussian id of each pixel's last contributor (-1: none) - n_contrib made list-independent."""
    gx = (W + 15) // 16
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    tile = (ys // 16) * gx + xs // 16
    start = st["ranges"].reshape(-1, 2)[:, 0].long()[tile]
    n = st["n_contrib"].reshape(H, W).long()
    pl = st["point_list"].long()
    if pl.numel() == 0:
        return torch.full((H, W), -1, dtype=torch.long)
    ids = pl[(start + n - 1).clamp(0, pl.numel() - 1)]
    return torch.where(n > 0, ids, torch.full_like(ids, -1))


def compare_forward(h, o, name, skip=(), culled=False):
    if culled:
        skip = tuple(skip) + BINNING_STATE
        assert h["num_rendered"] <= o["num_rendered"], name
    else:
        assert h["num_rendered"] == o["num_rendered"], name
    for k in ("radii", "clamped") + BINNING_STATE:
        if k in skip:
            continue
        assert torch.equal(h[k], o[k]), "%s: %s not bit-exact" % (name, k)
    for k in ("depths", "means2D", "conic_opacity", "rgb", "cov3D"):
        if k in skip:  # state the reference never materialises in this input mode (precomputed colour / cov3D)
            continue
        assert torch.equal(h[k].view(torch.int32), o[k].view(torch.int32)), "%s: %s not bit-exact" % (name, k)
    # image
    dc = (h["color"] - o["color"]).abs().amax(dim=0)
    scale = max(1.0, float(o["color"].abs().max()))
    bad = dc > TOL * scale
    H_, W_ = dc.shape
    nflip = int((last_contributor_id(h, W_, H_) != last_contributor_id(o, W_, H_)).sum())
    npix = dc.numel()
    # every out-of-tolerance pixel must be explained by a threshold flip and stay below one
    # quantisation step of alpha (1/255) times the colour range
    assert int(bad.sum()) <= max(2, npix // 20000), "%s: %d/%d pixels beyond %.0e" % (name, int(bad.sum()), npix, TOL)
    assert float(dc.max()) <= 1.5 / 255.0 * max(1.0, float(o["rgb"].abs().max())), "%s: max colour err %.3e" % (name, float(dc.max()))
    assert nflip <= flip_bound(npix), "%s: last contributor differs on %d pixels (bound %d)" % (name, nflip, flip_bound(npix))
    FLIPS[name] = dict(nflip=nflip, npix=npix, bad=int(bad.sum()))
    dT = (h["final_T"] - o["final_T"]).abs()
    assert int((dT > TOL).sum()) <= max(2, npix // 20000)
    di = (h["invdepth"] - o["invdepth"]).abs()
    assert int((di > TOL * max(1.0, float(o["invdepth"].abs().max()))).sum()) <= max(2, npix // 20000)
    return dict(bad=int(bad.sum()), nflip=nflip, maxerr=float(dc.max()))


def flip_mask(h, o):
    """[H, W] bool: pixels whose forward state shows a threshold decision that went the other way."""
    scale = max(1.0, float(o["color"].abs().max()))
    m = (h["color"] - o["color"]).abs().amax(dim=0) > 0.2 * TOL * scale
    m |= (h["final_T"] - o["final_T"]).abs().reshape(m.shape) > 0.2 * TOL
    m |= last_contributor_id(h, m.shape[1], m.shape[0]) != last_contributor_id(o, m.shape[1], m.shape[0])
    assert int(m.sum()) <= flip_bound(m.numel()), "%d flipped pixels (bound %d)" % (int(m.sum()), flip_bound(m.numel()))
    return m


SMALL = [
    dict(P=120, seed=1, W=48, H=40, aa=False, bg=(0.0, 0.0, 0.0), eye=(3.2, 1.0, 1.5)),
    dict(P=160, seed=2, W=37, H=53, aa=True, bg=(1.0, 0.5, 0.25), eye=(-2.5, 2.8, -0.7)),
    dict(P=90, seed=3, W=64, H=32, aa=False, bg=(0.2, 0.9, 0.1), eye=(0.6, -1.4, 0.4), big=True),
    dict(P=100, seed=4, W=40, H=40, aa=True, bg=(0.0, 0.0, 0.0), eye=(3.0, 0.2, 2.0), precomp_color=True, precomp_cov=True),
    dict(P=100, seed=5, W=33, H=47, aa=False, bg=(0.3, 0.3, 0.3), eye=(2.0, 2.0, 2.0), sh_degree=1),
    dict(P=700, seed=6, W=160, H=96, aa=False, bg=(0.0, 0.0, 0.0), eye=(2.9, -1.0, 0.5), big=True),  # >256 entries per tile
]


def grads_close(hg, og, name):
    """every tensor at TOL, dL_dscales / dL_drotations at helpers.CHAIN_TOL (why: helpers.py); prints what it measured"""
    from helpers import check_grads
    check_grads(hg, og, name)


@pytest.mark.parametrize("case", SMALL, ids=lambda c: "P%d_%dx%d_s%d" % (c["P"], c["W"], c["H"], c["seed"]))
def test_small_scenes_forward_state_and_gradients(hip, oracle, case):
    sc = small_scene(case["P"], case["seed"], case.get("precomp_color", False), case.get("precomp_cov", False),
                     case.get("sh_degree", 3), case.get("big", False))
    cam = synthetic.look_at_camera(case["eye"], case["W"], case["H"], FoVx=0.9)
    bg = torch.tensor(case["bg"])
    name = "P%d_s%d" % (case["P"], case["seed"])
    h = forward_state(hip, sc, cam, torch.device("cuda"), bg, case["aa"])
    o = forward_state(oracle.backend, sc, cam, torch.device("cpu"), bg, case["aa"])
    skip = (("rgb", "clamped") if case.get("precomp_color") else ()) + (("cov3D",) if case.get("precomp_cov") else ())
    compare_forward(h, o, name, skip)

    g = torch.Generator().manual_seed(100 + case["seed"])
    dL_dcolor = torch.randn((3, case["H"], case["W"]), generator=g)
    dL_dinv = torch.randn((1, case["H"], case["W"]), generator=g) * 0.3
    keep = (~flip_mask(h, o)).float()
    dL_dcolor, dL_dinv = dL_dcolor * keep, dL_dinv * keep
    ho = run_scene(dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, sc, cam, torch.device("cuda"), bg=bg,
                   antialiasing=case["aa"], dL_dcolor=dL_dcolor, dL_dinvdepth=dL_dinv)
    oo = run_scene(oracle.Rasterizer, oracle.Settings, sc, cam, torch.device("cpu"), bg=bg, antialiasing=case["aa"],
                   dL_dcolor=dL_dcolor, dL_dinvdepth=dL_dinv)
    grads_close(ho["grads"], oo["grads"], name)


@pytest.mark.parametrize("kind,P,W,H,deg", [("init", 10000, 400, 400, 0), ("trained", 10000, 400, 400, 3),
                                           ("trained", 60000, 800, 800, 3), ("trained", 30000, 1920, 1080, 2),
                                           ("trained", 20000, 3840, 2160, 1), ("trained", 15000, 1921, 1081, 3)])
def test_config_sized_scenes(hip, oracle, kind, P, W, H, deg):
    """BASELINE config-1 size (10 k Gaussians, 400x400) and larger images, incl. a 1080p tile grid
    (120x68 tiles, last tile row half empty, 45-bit sort keys), a 4K grid (240x135 = 32 400 tiles: 15 tile bits, the
    XCD-padded launch) and a size that is a multiple of nothing (1921x1081: partial tiles on both edges)."""
    gen = synthetic.init_like if kind == "init" else synthetic.trained_like
    sc = gen(P, seed=0, sh_degree=deg)
    cam = synthetic.orbit_cameras(W, H)[3]
    bg = torch.zeros(3)
    name = "%s_P%d_%dx%d" % (kind, P, W, H)
    h = forward_state(hip, sc, cam, torch.device("cuda"), bg, False)
    o = forward_state(oracle.backend, sc, cam, torch.device("cpu"), bg, False)
    info = compare_forward(h, o, name)
    print(name, "R=%d" % h["num_rendered"], info)
    g = torch.Generator().manual_seed(5)
    dL_dcolor = torch.randn((3, H, W), generator=g) * (~flip_mask(h, o)).float()
    ho = run_scene(dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, sc, cam, torch.device("cuda"), bg=bg,
                   dL_dcolor=dL_dcolor)
    oo = run_scene(oracle.Rasterizer, oracle.Settings, sc, cam, torch.device("cpu"), bg=bg, dL_dcolor=dL_dcolor)
    grads_close(ho["grads"], oo["grads"], name)


def test_empty_and_all_culled(hip):
    dev = torch.device("cuda")
    cam = synthetic.look_at_camera((3.0, 0.0, 0.0), 64, 48)
    S = dgr.GaussianRasterizationSettings
    from helpers import settings_for
    rs = settings_for(S, cam, torch.tensor([0.1, 0.2, 0.3]), 0, dev)
    rast = dgr.GaussianRasterizer(rs)
    # P == 0: zero outputs (rasterize_points.cu:88)
    z = torch.zeros((0, 3), device=dev)
    color, radii, invd = rast(means3D=z, means2D=z, opacities=torch.zeros((0, 1), device=dev),
                              colors_precomp=torch.zeros((0, 3), device=dev), scales=z, rotations=torch.zeros((0, 4), device=dev))
    assert color.shape == (3, 48, 64) and float(color.abs().max()) == 0 and radii.numel() == 0
    # everything behind the camera: background only, zero gradients
    m = torch.tensor([[10.0, 0.0, 0.0], [12.0, 1.0, 0.0]], device=dev, requires_grad=True)
    color, radii, invd = rast(means3D=m, means2D=torch.zeros_like(m), opacities=torch.full((2, 1), 0.5, device=dev),
                              colors_precomp=torch.rand((2, 3), device=dev), scales=torch.full((2, 3), 0.1, device=dev),
                              rotations=torch.tensor([[1.0, 0, 0, 0]] * 2, device=dev))
    assert int(radii.abs().sum()) == 0
    assert torch.allclose(color[:, 0, 0].cpu(), torch.tensor([0.1, 0.2, 0.3]))
    color.sum().backward()
    assert float(m.grad.abs().max()) == 0.0
    vis = rast.markVisible(torch.tensor([[10.0, 0, 0], [0.0, 0, 0]], device=dev))
    assert vis.tolist() == [False, True]


def test_mark_visible_vs_oracle_around_the_near_plane(hip, oracle):
    """markVisible (rasterizer_impl.cu:54-66, in_frustum of auxiliary.h:139-160: view-space z > 0.2): HIP against the oracle on a
    million points whose depths straddle 0.2 - a slab of them within a few ulps of the plane, where the fp32 evaluation order
    of the 4 x 4 transform decides - for three cameras; and against the float64 value away from the plane."""
    import numpy as np
    P = 1_000_000
    rng = np.random.RandomState(11)
    for ci in (0, 9, 20):
        cam = synthetic.orbit_cameras(640, 480)[ci]
        V = cam.world_view_transform.double()          # (row-vector convention: p_view = [p, 1] @ V)
        pts = torch.from_numpy(rng.uniform(-6.0, 6.0, size=(P, 3)))
        # move the first 600 k points onto the plane z_view = 0.2 (+- a spread from 1e-8 to 1e-2), along the view axis
        z = (torch.cat((pts, torch.ones((P, 1), dtype=torch.float64)), dim=1) @ V)[:, 2]
        axis = V[:3, 2] / (V[:3, 2] @ V[:3, 2])
        n = 600_000
        spread = torch.from_numpy(10.0 ** rng.uniform(-8, -2, size=n) * rng.choice([-1.0, 1.0], size=n))
        pts[:n] += ((0.2 + spread) - z[:n])[:, None] * axis[None, :]
        pts[n:n + 1000] += (0.2 - z[n:n + 1000])[:, None] * axis[None, :]      # as exactly on it as float64 gets
        p32 = pts.float()
        h = hip.mark_visible(p32.cuda(), cam.world_view_transform.cuda(), cam.full_proj_transform.cuda()).cpu()
        o = oracle.backend.mark_visible(p32, cam.world_view_transform, cam.full_proj_transform)
        assert torch.equal(h, o), (ci, int((h != o).sum()))
        z32 = (torch.cat((p32.double(), torch.ones((P, 1), dtype=torch.float64)), dim=1) @ V)[:, 2]
        clear = (z32 - 0.2).abs() > 1e-5
        assert torch.equal(h[clear], (z32 > 0.2)[clear])
        assert 0.2 < float(h.float().mean()) < 0.8 and int((~clear).sum()) > 100_000


@pytest.mark.parametrize("P,seed", [(5, 0), (1000, 1), (4097, 2), (100000, 3), (1000000, 4), (1000000, 5)])
def test_knn_bit_exact_vs_oracle(hip, oracle, P, seed):
    import numpy as np
    from gsplat_amd.knn import dist2
    from simple_knn._C import distCUDA2
    rng = np.random.RandomState(seed)
    pts = torch.from_numpy((rng.random_sample((P, 3)) * 2.6 - 1.3).astype(np.float32))
    if P == 4097:
        pts[100:140] = pts[7]  # duplicates
    if seed == 5:  # the size the metric is quoted on, clustered like a point cloud (dense blobs + sparse halo + repeats)
        c = rng.random_sample((64, 3)).astype(np.float32) * 2.0 - 1.0
        blob = c[rng.randint(0, 64, P)] + (rng.standard_normal((P, 3)) * 0.02).astype(np.float32)
        pts = torch.where(torch.from_numpy(rng.random_sample((P, 1)) < 0.8), torch.from_numpy(blob), pts)
        pts[5000:5200] = pts[17]
    ref = dist2(oracle.api, pts)
    out = distCUDA2(pts.cuda()).cpu()
    assert torch.equal(out.view(torch.int32), ref.view(torch.int32))


def test_optimistic_capacity_path_equals_synchronous_path(hip):
    """The no-host-bubble forward (binning capacity predicted from earlier calls) must give the same
    bytes as the synchronous path, including when the prediction overflows and the phase is re-run."""
    sc = synthetic.trained_like(20000, seed=3)
    cam = synthetic.orbit_cameras(640, 360)[5]
    dev = torch.device("cuda")
    bg = torch.zeros(3)
    old = (hip.optimistic, hip._capacity_hint)
    try:
        hip.optimistic = False
        ref = forward_state(hip, sc, cam, dev, bg, False)
        hip.optimistic = True
        for hint in (64, ref["num_rendered"] - 1, ref["num_rendered"], 10 * ref["num_rendered"]):
            hip._capacity_hint = hint
            got = forward_state(hip, sc, cam, dev, bg, False)
            assert got["num_rendered"] == ref["num_rendered"]
            for k in ("point_list", "keys_sorted", "ranges", "n_contrib"):
                assert torch.equal(got[k], ref[k]), (hint, k)
            assert torch.equal(got["color"], ref["color"]), hint
        g = torch.Generator().manual_seed(0)
        dL = torch.randn((3, 360, 640), generator=g)
        hip._capacity_hint = 3 * ref["num_rendered"]
        a = run_scene(dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, sc, cam, dev, bg=bg, dL_dcolor=dL)
        hip.optimistic = False
        b = run_scene(dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, sc, cam, dev, bg=bg, dL_dcolor=dL)
        for k in b["grads"]:
            scale = max(float(b["grads"][k].abs().max()), 1e-12)
            assert float((a["grads"][k] - b["grads"][k]).abs().max()) / scale < 1e-4, k
    finally:
        hip.optimistic, hip._capacity_hint = old


def test_two_contexts_alive_rgb_plus_nir_pass(hip, oracle):
    """train_nir.py renders RGB and then a second pass with precomputed (NIR) colours before any backward runs
    (mult-dwtgs/gaussian_renderer/__init__.py:151-258): two forward contexts coexist and their gradients add."""
    sc = small_scene(600, 21, big=True)
    cam = synthetic.look_at_camera((3.0, 0.4, 0.8), 96, 64, FoVx=0.9)
    outs = {}
    for name, dev, Rast, Settings in (("hip", torch.device("cuda"), dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings),
                                      ("oracle", torch.device("cpu"), oracle.Rasterizer, oracle.Settings)):
        from helpers import settings_for
        p = {k: sc[k].to(dev).requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
        nir_albedo = torch.linspace(-1, 1, 600).reshape(600, 1).to(dev).requires_grad_(True)
        rs = settings_for(Settings, cam, torch.zeros(3), 3, dev)
        rast = Rast(rs)
        m2d = torch.zeros_like(p["means3D"], requires_grad=True)
        rgb, _, _ = rast(means3D=p["means3D"], means2D=m2d, opacities=p["opacities"], shs=p["shs"], scales=p["scales"],
                         rotations=p["rotations"])
        nir_col = torch.sigmoid(nir_albedo).repeat(1, 3)
        nir, _, _ = rast(means3D=p["means3D"], means2D=m2d, opacities=p["opacities"], colors_precomp=nir_col,
                         scales=p["scales"], rotations=p["rotations"])
        g = torch.Generator().manual_seed(8)
        w1, w2 = torch.randn((3, 64, 96), generator=g).to(dev), torch.randn((64, 96), generator=g).to(dev)
        ((rgb * w1).sum() + (nir[0] * w2).sum()).backward()
        outs[name] = {k: v.grad.cpu() for k, v in p.items()}
        outs[name]["nir"] = nir_albedo.grad.cpu()
        outs[name]["img"] = nir.detach().cpu()
    assert float((outs["hip"]["img"] - outs["oracle"]["img"]).abs().max()) < 1e-5
    grads_close({k: v for k, v in outs["hip"].items() if k != "img"}, {k: v for k, v in outs["oracle"].items() if k != "img"},
                "rgb+nir")


def test_reference_python_geometry_fixture_on_the_gpu(hip):
    """tests/golden/geometry.npz (the reference's python covariance path and geom_transform_points, see
    make_golden.gen_geometry) against the HIP library's own state: cov3D of gs_export_geom == the python covariance,
    the pixel mean == ndc2Pix of geom_transform_points, and the python-cov render path (cov3D_precomp,
    gaussian_renderer/__init__.py:64-68) == the scale / rotation path."""
    import os

    import numpy as np
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "geometry.npz"))
    dev = torch.device("cuda")
    P = z["scales"].shape[0]
    W, H, FoVx = 1237, 822, 0.69
    FoVy = synthetic.focal2fov(synthetic.fov2focal(FoVx, W), H)
    wvt, full = torch.tensor(z["world_view_transform"]), torch.tensor(z["full_proj_transform"])
    cam = synthetic.Camera(H, W, FoVx, FoVy, wvt, full, wvt.inverse()[3, :3].contiguous())
    g = torch.Generator().manual_seed(5)
    base = dict(means3D=torch.tensor(z["points"]), opacities=torch.sigmoid(torch.randn((P, 1), generator=g)),
                colors_precomp=torch.rand((P, 3), generator=g), sh_degree=0)
    sr = dict(base, scales=torch.tensor(z["scales"]) * 6.0, rotations=torch.tensor(z["quats"]))
    st = forward_state(hip, sr, cam, dev, torch.zeros(3), False)
    vis = st["radii"] > 0
    assert int(vis.sum()) > P // 2
    ref_cov = torch.tensor(z["cov6_m10"]) * 36.0
    assert float((st["cov3D"][vis] - ref_cov[vis]).abs().max()) <= 2e-6 * float(ref_cov.abs().max())
    pp = torch.tensor(z["p_proj"]).double()
    px = torch.stack([((pp[:, 0] + 1.0) * W - 1.0) * 0.5, ((pp[:, 1] + 1.0) * H - 1.0) * 0.5], dim=1)
    assert float((st["means2D"][vis].double() - px[vis]).abs().max()) <= 5e-4  # pixels; fp32 at |x| ~ 1e3
    assert float((st["depths"][vis].double() - torch.tensor(z["p_view"]).double()[vis, 2]).abs().max()) <= 1e-5
    pc = forward_state(hip, dict(base, cov3D_precomp=ref_cov), cam, dev, torch.zeros(3), False)
    assert int((pc["radii"] != st["radii"]).sum()) <= 1
    assert float((pc["color"] - st["color"]).abs().max()) < TOL  # (measured 3e-5: the covariance arrives rounded to fp32)
