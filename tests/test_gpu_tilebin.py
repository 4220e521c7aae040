"""Two-level binning of GsView.tile_cull = 0 / 1 (csrc/gs_tilebin.hip: region entries with 16-bit tile masks, partitioned by
region, expanded into the tile lists) - replaces duplicateWithKeys + cub::DeviceRadixSort::SortPairs + identifyTileRanges
(rasterizer_impl.cu:70-138, 280-321).

  1. tile_cull = 0: num_rendered, point_list, ranges and the rebuilt 64-bit keys are the ORACLE's bit for bit (the
     reference's lists), at shapes that cut regions and tiles on both axes, with splats that cover the whole image, with
     one Gaussian, and with fewer Gaussians than the one-workgroup depth sort takes;
  2. the row-wise entry enumeration (the capacity guard, GsView.debug bit 1 forces it on every third Gaussian) builds the
     very same lists in both modes;
  3. the lists hold what the geometry phase counted: the header's instance total equals num_rendered;
  4. a capacity that is too small is detected and the view rendered again.
(tests/test_gpu_fullsize.py holds the same comparison at BASELINE C2 / C3 / C4; tests/test_gpu_tilecull.py pins the culled
lists against the oracle's; tests/test_gpu_regionbin.py compares region binning with this path list by list.)"""
import numpy as np
import pytest
import torch

from gsplat_amd import synthetic
from helpers import canonical_lists
from test_gpu_raster_parity import forward_state

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


@pytest.fixture(autouse=True)
def lsd_lists(hip):
    old = (hip.tile_cull, hip.binning, hip.depth_limit_on, hip._capacity_hint, hip._capacity_hint_limited,
           hip.force_rowwise_entries)
    hip.binning, hip.depth_limit_on = "lsd", False
    hip._cam_cache.clear()
    yield
    (hip.tile_cull, hip.binning, hip.depth_limit_on, hip._capacity_hint, hip._capacity_hint_limited,
     hip.force_rowwise_entries) = old
    hip._cam_cache.clear()


def scene(kind, P, deg, seed=0):
    if kind == "init":
        return synthetic.init_like(P, seed=seed, sh_degree=deg)
    if kind == "huge":   # every splat covers most of the image: hundreds of regions per Gaussian
        sc = synthetic.trained_like(P, seed=seed, sh_degree=deg)
        sc["scales"] = sc["scales"] * 40.0
        return sc
    return synthetic.trained_like(P, seed=seed, sh_degree=deg)


SHAPES = [
    ("trained", 40000, 800, 800, 3),
    ("trained", 30000, 1920, 1080, 2),  # 120 x 68 tiles: the last region row holds a single tile row... of half tiles
    ("trained", 5000, 250, 130, 1),     # partial tiles and partial regions on both axes
    ("init", 10000, 400, 400, 0),       # isotropic, many regions per Gaussian
    ("huge", 300, 333, 211, 0),
    ("trained", 4, 96, 64, 0),
    ("trained", 700, 48, 40, 1),        # one region row; fewer Gaussians than the one-workgroup sort holds
    ("trained", 20000, 2100, 80, 1),    # 132 x 5 tiles: more than 256 regions would need two partition passes... (33 x 2 here)
    ("trained", 20000, 4100, 1100, 1),  # 257 x 69 tiles = 65 x 18 = 1170 regions: two partition passes
    ("huge", 200, 3840, 2160, 0),       # splats that cover a 4K image: up to 2 040 entries per Gaussian, 256 of them per workgroup
]


@pytest.mark.parametrize("kind,P,W,H,deg", SHAPES)
def test_reference_lists_are_the_oracles_bit_for_bit(hip, oracle, kind, P, W, H, deg):
    sc = scene(kind, P, deg)
    cam = synthetic.orbit_cameras(W, H)[2]
    bg = torch.tensor([0.1, 0.2, 0.3])
    hip.tile_cull = False
    h = forward_state(hip, sc, cam, DEV, bg, False)
    o = forward_state(oracle.backend, sc, cam, torch.device("cpu"), bg, False)
    assert h["num_rendered"] == o["num_rendered"]
    for k in ("radii", "tiles_touched", "point_offsets", "ranges", "point_list", "keys_sorted"):
        assert torch.equal(h[k], o[k]), k
    hip.force_rowwise_entries = True
    h2 = forward_state(hip, sc, cam, DEV, bg, False)
    for k in ("ranges", "point_list", "keys_sorted", "n_contrib", "color"):
        assert torch.equal(h[k], h2[k]), ("row-wise entries", k)


@pytest.mark.parametrize("kind,P,W,H,deg", SHAPES[:6] + SHAPES[-1:])
def test_culled_lists_do_not_depend_on_the_entry_enumeration(hip, kind, P, W, H, deg):
    sc = scene(kind, P, deg)
    cam = synthetic.orbit_cameras(W, H)[4]
    bg = torch.zeros(3)
    hip.tile_cull = True
    a = forward_state(hip, sc, cam, DEV, bg, True)
    hip.force_rowwise_entries = True
    b = forward_state(hip, sc, cam, DEV, bg, True)
    assert a["num_rendered"] == b["num_rendered"]
    # what the lists hold is what the geometry phase counted, Gaussian by Gaussian
    counts = torch.bincount(a["point_list"].long(), minlength=a["tiles_touched"].numel())
    assert torch.equal(counts.int(), a["tiles_touched"].int())
    for k in ("ranges", "point_list", "keys_sorted", "n_contrib", "color", "final_T"):
        assert torch.equal(a[k], b[k]), k
    # tile after tile, every list sorted by (depth bits, index): the exported keys are sorted as 64-bit numbers
    keys = a["keys_sorted"]
    assert bool((keys[1:] >= keys[:-1]).all())
    ca, ka = canonical_lists(a)
    assert int(ca.sum()) == a["num_rendered"] and np.array_equal(np.sort(ka), np.unique(ka))  # no duplicate pairs


def test_too_small_a_capacity_is_detected_and_the_view_rendered_again(hip):
    sc = synthetic.trained_like(20000, seed=3)
    cam = synthetic.orbit_cameras(640, 360)[5]
    bg = torch.zeros(3)
    for cull in (False, True):
        hip.tile_cull = cull
        hip._capacity_hint = 0
        ref = forward_state(hip, sc, cam, DEV, bg, False)
        R = ref["num_rendered"]
        for hint in (1, R // 3, R - 1, R, 10 * R):
            hip._capacity_hint = hint
            got = forward_state(hip, sc, cam, DEV, bg, False)
            assert got["num_rendered"] == R, hint
            for k in ("ranges", "point_list", "color", "n_contrib"):
                assert torch.equal(got[k], ref[k]), (cull, hint, k)
