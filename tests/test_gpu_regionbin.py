"""Region binning (GsView.tile_cull = 2, csrc/gs_regionbin.hip) against the LSD binning (tile_cull = 1,
csrc/gs_binning.hip: depth sort of the Gaussians -> instance emission -> stable partition by tile id), which
tests/test_gpu_tilecull.py pins against the oracle's reference lists.  Replaces duplicateWithKeys +
cub::DeviceRadixSort::SortPairs + identifyTileRanges (rasterizer_impl.cu:70-138, 280-321) for the culled lists.

  1. every tile's list holds the same Gaussians in the same order (only the place of a list inside point_list differs),
     num_rendered is the same number, every pixel output is bit-identical, gradients agree to float-atomic order;
  2. a capacity that is too small (lists or region buckets) is detected and the view rendered again - same lists;
  3. depth-limited lists: the region path cuts per tile exactly (the LSD path keeps a span-trimmed superset), both
     contain every entry the blend visits and render the un-limited bits;
  4. a region that holds more Gaussians than one workgroup sorts falls back to the LSD path.
"""
import numpy as np
import pytest
import torch

import diff_gaussian_rasterization as dgr
from gsplat_amd import synthetic
from helpers import canonical_lists, run_scene
from test_gpu_raster_parity import forward_state, last_contributor_id

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


@pytest.fixture(autouse=True)
def culled_lists(hip):
    old = (hip.tile_cull, hip.binning, hip.depth_limit_on, hip._capacity_hint, hip._capacity_hint_limited)
    hip.tile_cull, hip.depth_limit_on = True, False
    hip._cam_cache.clear()
    hip._region_off.clear()
    yield
    hip.tile_cull, hip.binning, hip.depth_limit_on, hip._capacity_hint, hip._capacity_hint_limited = old
    hip._cam_cache.clear()
    hip._region_off.clear()


def both(hip, sc, cam, bg, aa=False):
    hip.binning = "lsd"
    a = forward_state(hip, sc, cam, DEV, bg, aa)
    hip.binning = "region"
    b = forward_state(hip, sc, cam, DEV, bg, aa)
    return a, b


SCENES = [
    ("init", 10000, 400, 400, 0, False),       # large isotropic splats: many regions per Gaussian
    ("trained", 60000, 800, 800, 3, True),
    ("trained", 30000, 1920, 1080, 2, False),  # regions cut by the image border (68 tile rows = 17 regions)
    ("trained", 5000, 250, 130, 1, False),     # partial tiles and partial regions on both axes
]


@pytest.mark.parametrize("kind,P,W,H,deg,aa", SCENES)
def test_region_lists_equal_the_lsd_lists_tile_by_tile(hip, kind, P, W, H, deg, aa):
    gen = synthetic.init_like if kind == "init" else synthetic.trained_like
    sc = gen(P, seed=0, sh_degree=deg)
    cam = synthetic.orbit_cameras(W, H)[3]
    bg = torch.tensor([0.3, 0.1, 0.2])
    a, b = both(hip, sc, cam, bg, aa)
    assert a["num_rendered"] == b["num_rendered"] > 0
    ca, ka = canonical_lists(a)
    cb, kb = canonical_lists(b)
    assert np.array_equal(ca, cb), "tile list lengths differ"
    assert np.array_equal(ka, kb), "a tile's list differs"
    for k in ("color", "invdepth", "final_T", "radii", "n_contrib"):
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(last_contributor_id(a, W, H), last_contributor_id(b, W, H))
    # the exported 64-bit keys: every entry carries its tile, every list is sorted by depth
    keys = b["keys_sorted"]
    tiles = keys >> 32
    assert torch.equal(keys[torch.sort(tiles, stable=True).indices], torch.sort(keys).values)
    # gradients: same pairs, same per-tile order; across tiles the float atomics arrive in another order
    g = torch.Generator().manual_seed(5)
    dL = torch.randn((3, H, W), generator=g)
    hip.binning = "lsd"
    ga = run_scene(dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, sc, cam, DEV, bg=bg, antialiasing=aa, dL_dcolor=dL)
    hip.binning = "region"
    gb = run_scene(dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, sc, cam, DEV, bg=bg, antialiasing=aa, dL_dcolor=dL)
    assert torch.equal(ga["color"], gb["color"])
    for k in ga["grads"]:
        x, y = ga["grads"][k].double(), gb["grads"][k].double()
        assert float((x - y).abs().max()) <= 5e-4 * max(1e-12, float(x.abs().max())), k


def test_capacity_overflow_is_detected_and_the_view_rendered_again(hip):
    sc = synthetic.trained_like(20000, seed=3)
    cam = synthetic.orbit_cameras(640, 360)[5]
    bg = torch.zeros(3)
    hip.binning = "region"
    ref = forward_state(hip, sc, cam, DEV, bg, False)
    R = ref["num_rendered"]
    kref = canonical_lists(ref)[1]
    regions = ((640 // 16 + 3) // 4) * (((360 + 15) // 16 + 3) // 4)
    # too few instances of room for the lists / for the region buckets / plenty
    for hint in (regions, R // 3, R - 1, 10 * R):
        hip._capacity_hint = hint
        got = forward_state(hip, sc, cam, DEV, bg, False)
        assert got["num_rendered"] == R, hint
        assert np.array_equal(canonical_lists(got)[1], kref), hint
        assert torch.equal(got["color"], ref["color"]) and torch.equal(got["n_contrib"], ref["n_contrib"]), hint
        assert hip._capacity_hint >= R


def test_depth_limited_region_lists(hip):
    """Second visit of a camera: per tile the region path keeps exactly the entries within the tile's bound - a subset of
    the LSD path's cut that still holds every entry the blend visits - and renders the un-limited bits."""
    P, W, H = 60000, 800, 800
    sc = synthetic.trained_like(P, seed=0, sh_degree=3)
    cam = synthetic.orbit_cameras(W, H)[3]
    bg = torch.tensor([0.2, 0.1, 0.3])
    full = {}
    cut = {}
    for mode in ("lsd", "region"):
        hip.binning = mode
        hip.depth_limit_on = True
        hip._cam_cache.clear()
        used0 = hip.depth_limit_stats["used"]
        full[mode] = forward_state(hip, sc, cam, DEV, bg, False)   # first visit measures the stop depths
        cut[mode] = forward_state(hip, sc, cam, DEV, bg, False)    # second visit: cut lists
        assert hip.depth_limit_stats["used"] == used0 + 1 and hip.last_status()[2] == 0
        third = forward_state(hip, sc, cam, DEV, bg, False)
        assert third["num_rendered"] == cut[mode]["num_rendered"]
        for k in ("color", "invdepth", "final_T", "radii", "n_contrib"):
            assert torch.equal(full[mode][k], cut[mode][k]), (mode, k)
            assert torch.equal(full[mode][k], third[k]), (mode, k)
    assert torch.equal(full["lsd"]["color"], full["region"]["color"])
    kf = canonical_lists(full["region"])[1]
    kl, kr = canonical_lists(cut["lsd"])[1], canonical_lists(cut["region"])[1]
    assert np.array_equal(kf[np.isin(kf, kr)], kr), "limited region list is not a subsequence of the full list"
    assert np.isin(kr, kl).all(), "the exact per-tile cut must lie inside the LSD path's span-trimmed cut"
    print("instances: full", len(kf), "lsd cut", len(kl), "region cut", len(kr))
    assert len(kr) <= len(kl) < 0.9 * len(kf)
    # every entry the blend visits is still there (prefix of each tile's list up to its deepest last contributor)
    from test_gpu_depth_limit import assert_prefix_property
    assert_prefix_property(full["region"], cut["region"], W, H)


def test_a_region_too_crowded_for_one_workgroup_falls_back_to_the_lsd_path(hip):
    """40 000 small Gaussians inside one 64 x 64-pixel region: more than the 16 384 entries a region's workgroup sorts in
    LDS.  The forward notices (status word 3), remembers the size and renders through the LSD path: same lists."""
    P, W, H = 40000, 256, 192
    g = torch.Generator().manual_seed(1)
    sc = synthetic.trained_like(P, seed=2, sh_degree=0)
    sc["means3D"] = (sc["means3D"] * 0.02).contiguous()          # everything projects into the centre of the image
    sc["scales"] = (sc["scales"] * 0.02).contiguous()
    cam = synthetic.orbit_cameras(W, H)[2]
    bg = torch.zeros(3)
    hip.binning = "lsd"
    a = forward_state(hip, sc, cam, DEV, bg, False)
    hip.binning = "region"
    hip._capacity_hint = 0
    b = forward_state(hip, sc, cam, DEV, bg, False)
    assert (P, W, H, False) in hip._region_off
    assert a["num_rendered"] == b["num_rendered"]
    assert np.array_equal(canonical_lists(a)[1], canonical_lists(b)[1]) and torch.equal(a["color"], b["color"])
    # sizes that fit keep using regions
    sc2 = synthetic.trained_like(2000, seed=2, sh_degree=0)
    c = forward_state(hip, sc2, cam, DEV, bg, False)
    assert (2000, W, H, False) not in hip._region_off and c["num_rendered"] > 0


def test_fused_train_step_on_region_lists_is_the_lsd_run(hip):
    """The deferred, depth-limited fused step (what bench.py times) on region-binned lists against the same trainer on
    the LSD path: the run is the same run (bar of test_gpu_depth_limit.test_training_with_limits_is_the_same_run)."""
    from test_gpu_depth_limit import make
    hip.binning = "lsd"
    a = make(hip)
    a.depth_limit = "deferred"
    la = [a.step(k) for k in range(12)]
    a.sync()
    hip.binning = "region"
    hip._cam_cache.clear()
    hip._capacity_hint = hip._capacity_hint_limited = 0
    b = make(hip)
    b.depth_limit = "deferred"
    used0, failed0 = hip.depth_limit_stats["used"], hip.depth_limit_stats["failed"]
    lb = [b.step(k) for k in range(12)]
    b.sync()
    la, lb = [float(x) for x in la], [float(x) for x in lb]
    used, failed = hip.depth_limit_stats["used"] - used0, hip.depth_limit_stats["failed"] - failed0
    print("limited views", used, "fall-backs", failed, "\n", la, "\n", lb)
    assert used >= 8 and failed <= 3
    assert a.model.optimizer.t == b.model.optimizer.t == 12
    assert max(abs(x - y) for x, y in zip(la, lb)) <= 1e-3 * max(la)
    d = (a.model.flat - b.model.flat).double()
    assert float(d.pow(2).mean().sqrt()) <= 1e-4 * float(a.model.flat.double().pow(2).mean().sqrt())
    assert torch.equal(a.model.denom, b.model.denom)
