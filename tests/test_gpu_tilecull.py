"""GPU parity of the culled instance lists (GsView.tile_cull = 1, csrc/gs_tilecull.h) - the product default.

The reference emits one instance per tile of a Gaussian's bounding square (rasterizer_impl.cu:70-111) and
rejects per pixel (alpha < 1/255, forward.cu:352-356).  The HIP path drops the (tile, Gaussian) pairs on
which no pixel can pass that test.  What is checked here, against the oracle's reference lists:
  1. the kept pairs are exactly a SUBSEQUENCE of the reference's sorted list (same tile, same order,
     nothing added);
  2. every dropped pair is dead: on all 256 pixels of the tile the reference's own fp32 expression gives
     power > 0 or alpha < 1/255 (so colour, depth, final_T, the last contributor and every gradient are
     sums over exactly the same pairs);
  3. hence the pixels of the two HIP modes are bit-identical, and images / gradients match the oracle at the
     same tolerance as with the reference lists.
"""
import numpy as np
import pytest
import torch

import diff_gaussian_rasterization as dgr
from gsplat_amd import synthetic
from helpers import run_scene
from test_gpu_raster_parity import (TOL, compare_forward, flip_mask, forward_state, grads_close,
                                    last_contributor_id)
from test_oracle_dense import small_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def culled_lists(hip):
    old = hip.tile_cull
    hip.tile_cull = True
    yield
    hip.tile_cull = old


def pair_keys(st):
    """(tile << 32 | gaussian id) of every list entry, tile after tile, in list order (helpers.canonical_lists)."""
    from helpers import canonical_lists
    return canonical_lists(st)[1]


def check_lists(h, o, W, H, name):
    hk, ok = pair_keys(h), pair_keys(o)
    kept = np.isin(ok, hk)
    assert np.array_equal(ok[kept], hk), "%s: culled list is not a subsequence of the reference list" % name
    dropped = ok[~kept]
    gx = (W + 15) // 16
    tile, gid = dropped >> 32, dropped & 0xFFFFFFFF
    xy = o["means2D"].numpy().astype(np.float32)
    co = o["conic_opacity"].numpy().astype(np.float32)
    off = np.arange(16, dtype=np.float32)
    worst = 0.0
    f = np.float32
    for a in range(0, len(dropped), 200000):
        t, g = tile[a:a + 200000], gid[a:a + 200000]
        px = ((t % gx) * 16).astype(np.float32)[:, None, None] + off[None, None, :]
        py = ((t // gx) * 16).astype(np.float32)[:, None, None] + off[None, :, None]
        dx = xy[g, 0][:, None, None] - px
        dy = xy[g, 1][:, None, None] - py
        A, B, Cc, op = (co[g, i][:, None, None] for i in range(4))
        # forward.cu:343-356, evaluated op by op in fp32 like the oracle (no FMA)
        power = f(-0.5) * (A * dx * dx + Cc * dy * dy) - B * dx * dy
        alpha = np.minimum(f(0.99), op * np.exp(power))
        live = (power <= 0) & (alpha >= f(1.0 / 255.0))
        assert not live.any(), "%s: %d dropped pairs would have contributed" % (name, int(live.any(axis=(1, 2)).sum()))
        worst = max(worst, float(np.where(power <= 0, alpha, 0).max()))
    return dict(kept=len(hk), reference=len(ok), ratio=len(hk) / max(1, len(ok)), closest_alpha_x255=worst * 255)


SCENES = [
    ("init", 10000, 400, 400, 0, False),      # large isotropic, opacity 0.1
    ("trained", 10000, 400, 400, 3, False),   # anisotropic, all opacities
    ("trained", 60000, 800, 800, 3, True),    # anti-aliasing rescales the opacity the cull sees
    ("trained", 30000, 1920, 1080, 2, False),
]


@pytest.mark.parametrize("kind,P,W,H,deg,aa", SCENES)
def test_culled_lists_drop_only_dead_pairs_and_keep_order(hip, oracle, kind, P, W, H, deg, aa):
    gen = synthetic.init_like if kind == "init" else synthetic.trained_like
    sc = gen(P, seed=0, sh_degree=deg)
    cam = synthetic.orbit_cameras(W, H)[3]
    bg = torch.zeros(3)
    name = "%s_P%d_%dx%d" % (kind, P, W, H)
    h = forward_state(hip, sc, cam, torch.device("cuda"), bg, aa)
    o = forward_state(oracle.backend, sc, cam, torch.device("cpu"), bg, aa)
    info = check_lists(h, o, W, H, name)
    print(name, info)
    assert info["ratio"] < 0.8  # the cull is doing something
    compare_forward(h, o, name, culled=True)
    g = torch.Generator().manual_seed(5)
    dL = torch.randn((3, H, W), generator=g) * (~flip_mask(h, o)).float()
    ho = run_scene(dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, sc, cam, torch.device("cuda"), bg=bg,
                   antialiasing=aa, dL_dcolor=dL)
    oo = run_scene(oracle.Rasterizer, oracle.Settings, sc, cam, torch.device("cpu"), bg=bg, antialiasing=aa, dL_dcolor=dL)
    grads_close(ho["grads"], oo["grads"], name)


@pytest.mark.parametrize("seed,big,aa", [(1, False, False), (3, True, False), (6, True, True)])
def test_small_scenes_culled(hip, oracle, seed, big, aa):
    """Camera close to / inside the cloud: huge splats, rectangles clipped by the grid, near-degenerate conics."""
    sc = small_scene(300, seed, False, False, 3, big)
    cam = synthetic.look_at_camera((1.2, 0.4, 0.3), 96, 80, FoVx=1.1)
    bg = torch.tensor([0.1, 0.2, 0.3])
    name = "small_s%d" % seed
    h = forward_state(hip, sc, cam, torch.device("cuda"), bg, aa)
    o = forward_state(oracle.backend, sc, cam, torch.device("cpu"), bg, aa)
    print(name, check_lists(h, o, 96, 80, name))
    compare_forward(h, o, name, culled=True)


def test_low_opacity_and_degenerate_splats(hip, oracle):
    """opacity below 1/255 (never blends: zero instances, radii still reported), opacity exactly at the
    threshold, needle-shaped splats (cond(Q) large -> full rectangle fallback)."""
    P = 64
    g = torch.Generator().manual_seed(11)
    means = (torch.rand((P, 3), generator=g) - 0.5) * 1.2
    scales = torch.full((P, 3), 0.05)
    scales[:16] = torch.tensor([0.6, 0.0005, 0.0005])  # needles
    rot = torch.nn.functional.normalize(torch.randn((P, 4), generator=g), dim=1)
    op = torch.rand((P, 1), generator=g)
    op[16:32] = 0.0039  # < 1/255 = 0.003921...
    op[32:40] = 1.0 / 255.0
    op[40:44] = 0.0
    sc = dict(means3D=means, scales=scales, rotations=rot, opacities=op, colors_precomp=torch.rand((P, 3), generator=g),
              sh_degree=0)
    cam = synthetic.look_at_camera((2.5, 0.3, 0.2), 128, 96, FoVx=0.9)
    bg = torch.zeros(3)
    h = forward_state(hip, sc, cam, torch.device("cuda"), bg, False)
    o = forward_state(oracle.backend, sc, cam, torch.device("cpu"), bg, False)
    info = check_lists(h, o, 128, 96, "lowop")
    print(info)
    assert torch.equal(h["radii"], o["radii"])  # visibility reporting is the reference's
    tt = h["tiles_touched"]
    assert int(tt[16:32].sum()) == 0 and int(tt[40:44].sum()) == 0
    assert int(o["tiles_touched"][16:32].sum()) > 0
    compare_forward(h, o, "lowop", skip=("rgb", "clamped"), culled=True)


def test_both_list_modes_give_identical_pixels(hip):
    sc = synthetic.trained_like(40000, seed=2, sh_degree=3)
    cam = synthetic.orbit_cameras(960, 540)[7]
    dev = torch.device("cuda")
    bg = torch.tensor([0.3, 0.1, 0.2])
    hip.tile_cull = False
    a = forward_state(hip, sc, cam, dev, bg, False)
    hip.tile_cull = True
    b = forward_state(hip, sc, cam, dev, bg, False)
    assert b["num_rendered"] < 0.8 * a["num_rendered"]
    for k in ("color", "invdepth", "final_T", "radii"):
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(last_contributor_id(a, 960, 540), last_contributor_id(b, 960, 540))
    g = torch.Generator().manual_seed(1)
    dL = torch.randn((3, 540, 960), generator=g)
    hip.tile_cull = False
    ga = run_scene(dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, sc, cam, dev, bg=bg, dL_dcolor=dL)
    hip.tile_cull = True
    gb = run_scene(dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, sc, cam, dev, bg=bg, dL_dcolor=dL)
    for k in ga["grads"]:  # same pairs, different atomic order
        x, y = ga["grads"][k].double(), gb["grads"][k].double()
        assert float((x - y).abs().max()) <= 5e-4 * max(1e-12, float(x.abs().max())), k  # atomic-order noise only
