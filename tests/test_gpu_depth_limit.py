"""Depth-limited emission (GsScratch.tile_depth_limit, csrc/gs_tilecull.h): on a camera's second visit the (tile,
Gaussian) pairs that lie behind the depth at which the tile's blend stopped last time are not emitted.  The reference
emits them all (rasterizer_impl.cu:70-111) and never reads them (forward.cu:326-328 ends the tile when every pixel has
T < 1e-4).  Checked here:
  1. the cut lists are per tile a PREFIX-preserving subsequence of the full lists and every pixel output of the forward
     (colour, inverse depth, final T, last contributor) is bit-identical; the gradients differ by float-atomic order only;
  2. limits that are too tight for the current parameters are detected by the forward itself and the view is rendered
     again without them (same bits as a render that never had limits);
  3. a short training run with and without limits is the same run.
"""
import numpy as np
import pytest
import torch

import diff_gaussian_rasterization as dgr
from gsplat_amd import synthetic
from helpers import run_scene
from test_gpu_raster_parity import forward_state, last_contributor_id
from test_gpu_tilecull import pair_keys

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


@pytest.fixture(autouse=True)
def limits_on(hip):
    old = (hip.tile_cull, hip.depth_limit_on)
    hip.tile_cull = hip.depth_limit_on = True
    hip._cam_cache.clear()
    yield
    hip.tile_cull, hip.depth_limit_on = old
    hip._cam_cache.clear()


@pytest.fixture(params=[False, True], ids=["one_launch_step", "two_phase_step"])
def step_form(request, hip):
    """The fused step as one launch after the blend, or in its two-phase form (gs_step_uninstanced beside the blend + phase 2:
    what runs from 100 k Gaussians on - forced here at test size)."""
    if request.param:
        hip.TWO_PHASE, hip.TWO_PHASE_MIN_P = True, 0
    else:
        hip.TWO_PHASE = False
    n0 = hip.two_phase_launches
    yield request.param
    assert (hip.two_phase_launches > n0) == request.param
    for k in ("TWO_PHASE", "TWO_PHASE_MIN_P"):
        if k in hip.__dict__:
            delattr(hip, k)


def device_camera(cam):
    """(one device tensor per camera, like a trainer; the per-camera state is found by key or by the matrix's contents)"""
    return cam._replace(world_view_transform=cam.world_view_transform.to(DEV),
                        full_proj_transform=cam.full_proj_transform.to(DEV), camera_center=cam.camera_center.to(DEV))


def make(hip, P=30000, W=480, H=320, seed=3):
    """A fused single-GPU trainer whose ground truth is a render of a slightly different scene (the situation of a run
    that is converging: parameters drift slowly).  tests/test_gpu_fused_step.py's random-noise targets make opacities
    swing so fast that limits fail naturally every few visits - fine for results (the checks below would still hold),
    useless for tests that need to know WHICH step falls back."""
    import lgdwt_loss
    from gsplat_amd.trainer import GaussianModelLite, Trainer, camera_to, render
    from simple_knn._C import distCUDA2
    sc = synthetic.trained_like(P, seed=seed, sh_degree=3, knn=lambda x: distCUDA2(x.to(DEV)).cpu())
    cams = [camera_to(c, DEV) for c in synthetic.orbit_cameras(W, H)[:4]]
    g = torch.Generator().manual_seed(5)
    target = dict(sc, shs=sc["shs"] + 0.02 * torch.randn(sc["shs"].shape, generator=g))
    bg = torch.zeros(3, device=DEV)
    tm = GaussianModelLite(target, DEV, api=hip.api)
    with torch.no_grad():
        gts = [render(c, tm, dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, bg)["render"].clone() for c in cams]
    model = GaussianModelLite(sc, DEV, api=hip.api)
    crit = lgdwt_loss.criterion(dwt_enable=True, patch_dwt_enable=True)
    tr = Trainer(model, cams, gts, crit, dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, bg, optimizer_step=True)
    tr.FUSED_STEP = True
    return tr


def assert_prefix_property(full, cut, W, H):
    """per tile: cut list = full list minus entries, and everything up to the tile's deepest last contributor is kept"""
    fk, ck = pair_keys(full), pair_keys(cut)
    kept = np.isin(fk, ck)
    assert np.array_equal(fk[kept], ck), "cut list is not a subsequence of the full list"
    gx, gy = (W + 15) // 16, (H + 15) // 16
    n = full["n_contrib"].reshape(H, W).long()
    pad = torch.zeros((gy * 16, gx * 16), dtype=torch.long)
    pad[:H, :W] = n
    per_tile = pad.reshape(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(gy * gx, 256).max(dim=1).values.numpy()
    from helpers import canonical_lists
    counts = canonical_lists(full)[0]
    first = np.cumsum(counts) - counts
    for t in np.nonzero(per_tile)[0]:
        lo = first[t]
        assert kept[lo:lo + per_tile[t]].all(), "tile %d lost an entry its blend visits" % t


@pytest.mark.parametrize("P,W,H,deg,aa", [(60000, 800, 800, 3, False), (30000, 1920, 1080, 2, True), (10000, 400, 400, 0, False)])
def test_second_visit_blends_the_same_pixels_from_shorter_lists(hip, P, W, H, deg, aa):
    sc = synthetic.trained_like(P, seed=0, sh_degree=deg)
    cam = device_camera(synthetic.orbit_cameras(W, H)[3])
    bg = torch.tensor([0.2, 0.1, 0.3])
    used0 = hip.depth_limit_stats["used"]
    a = forward_state(hip, sc, cam, DEV, bg, aa)   # first visit: full (culled) lists, measures the stop depths
    assert hip.depth_limit_stats["used"] == used0
    b = forward_state(hip, sc, cam, DEV, bg, aa)   # second visit: cut lists
    assert hip.depth_limit_stats["used"] == used0 + 1 and hip.last_status()[2] == 0
    print("instances", a["num_rendered"], "->", b["num_rendered"])
    # (how much is cut depends on how much of the image saturates: little of 30 k Gaussians spread over 1080p does)
    assert b["num_rendered"] < (0.9 if P >= 60000 or W <= 400 else 1.0) * a["num_rendered"]
    for k in ("color", "invdepth", "final_T", "radii", "n_contrib"):
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(last_contributor_id(a, W, H), last_contributor_id(b, W, H))
    assert_prefix_property(a, b, W, H)
    # visibility is the reference's: a Gaussian all of whose pairs were cut still reports its radius
    assert int(((b["tiles_touched"] == 0) & (b["radii"] > 0)).sum()) > 0
    # third visit: the stop depths exported from cut lists are the same numbers
    c = forward_state(hip, sc, cam, DEV, bg, aa)
    assert c["num_rendered"] == b["num_rendered"] and torch.equal(c["color"], a["color"])
    # a limited forward that runs out of binning capacity is repeated with the same bounds (its geometry phase has counted
    # with them): the overflowed attempt must not have overwritten them with what it did not measure
    hip._capacity_hint_limited = 4096
    d = forward_state(hip, sc, cam, DEV, bg, aa)
    assert d["num_rendered"] == b["num_rendered"]
    for k in ("color", "invdepth", "final_T", "n_contrib"):
        assert torch.equal(d[k], a[k]), k
    # gradients: same pairs in the same per-tile order, different float-atomic order across tiles
    g = torch.Generator().manual_seed(5)
    dL = torch.randn((3, H, W), generator=g)
    hip.depth_limit_on = False
    ga = run_scene(dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, sc, cam, DEV, bg=bg, antialiasing=aa, dL_dcolor=dL)
    hip.depth_limit_on = True
    used1 = hip.depth_limit_stats["used"]
    gb = run_scene(dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, sc, cam, DEV, bg=bg, antialiasing=aa, dL_dcolor=dL)
    assert hip.depth_limit_stats["used"] == used1 + 1
    assert torch.equal(ga["color"], gb["color"])
    for k in ga["grads"]:
        x, y = ga["grads"][k].double(), gb["grads"][k].double()
        assert float((x - y).abs().max()) <= 5e-4 * max(1e-12, float(x.abs().max())), k  # atomic-order noise only


def test_stale_limits_are_detected_and_the_view_is_rendered_again(hip):
    """Between two visits the opacities drop to a third: tiles saturate much deeper (or not at all) than the limits
    allow.  The forward flags it, the glue renders again with full lists: same bits as a backend that never had limits.
    A small move of the parameters (a training step's worth) stays inside the margin and keeps the short lists."""
    P, W, H = 40000, 640, 480
    sc = synthetic.trained_like(P, seed=4, sh_degree=1)
    cam = device_camera(synthetic.orbit_cameras(W, H)[5])
    bg = torch.zeros(3)
    forward_state(hip, sc, cam, DEV, bg, False)
    faint = dict(sc, opacities=sc["opacities"] * 0.3)
    failed0 = hip.depth_limit_stats["failed"]
    b = forward_state(hip, faint, cam, DEV, bg, False)
    assert hip.depth_limit_stats["failed"] == failed0 + 1
    hip.depth_limit_on = False
    ref = forward_state(hip, faint, cam, DEV, bg, False)
    hip.depth_limit_on = True
    for k in ("color", "invdepth", "final_T", "radii", "n_contrib"):
        assert torch.equal(b[k], ref[k]), k
    assert b["num_rendered"] == ref["num_rendered"] and np.array_equal(pair_keys(b), pair_keys(ref))
    # the fallback re-measured the stop depths - and this camera's bounds now carry more slack (RasterBackend.SLACK: a camera
    # whose limits failed exports them x 1.25, x 1.6, ... until they have held for a while): the next visit of the faint scene
    # is limited again (on this scene the slack leaves nothing to cut), and exact
    ent = hip.camera_entry(W, H, viewmatrix=cam.world_view_transform)
    assert ent["slack_level"] == 1 and float(ent["slack"]) == hip.SLACK[1]
    used0 = hip.depth_limit_stats["used"]
    c = forward_state(hip, faint, cam, DEV, bg, False)
    assert hip.depth_limit_stats["failed"] == failed0 + 1 and hip.depth_limit_stats["used"] == used0 + 1
    assert c["num_rendered"] <= ref["num_rendered"]
    assert torch.equal(c["color"], ref["color"]) and torch.equal(c["n_contrib"], ref["n_contrib"])
    # ... and after SLACK_RELAX_AFTER visits that held, the slack is taken back one level
    for _ in range(hip.SLACK_RELAX_AFTER):
        c = forward_state(hip, faint, cam, DEV, bg, False)
    assert ent["slack_level"] == 0 and float(ent["slack"]) == 1.0 and hip.depth_limit_stats["failed"] == failed0 + 1
    assert torch.equal(c["color"], ref["color"])
    # a training step's worth of motion of the opaque scene, seen from another camera
    cam = device_camera(synthetic.orbit_cameras(W, H)[9])
    forward_state(hip, sc, cam, DEV, bg, False)
    g = torch.Generator().manual_seed(1)
    moved = dict(sc, means3D=sc["means3D"] + 2e-3 * torch.randn((P, 3), generator=g),
                 opacities=(sc["opacities"] * (1 + 0.02 * torch.randn((P, 1), generator=g))).clamp(0, 1))
    d = forward_state(hip, moved, cam, DEV, bg, False)
    hip.depth_limit_on = False
    ref2 = forward_state(hip, moved, cam, DEV, bg, False)
    hip.depth_limit_on = True
    assert hip.depth_limit_stats["failed"] == failed0 + 1, "a small parameter move should stay inside the margin"
    assert d["num_rendered"] < 0.9 * ref2["num_rendered"]
    for k in ("color", "invdepth", "final_T", "radii", "n_contrib"):
        assert torch.equal(d[k], ref2[k]), k


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_any_parameter_motion_between_visits_gives_the_unlimited_image(hip, seed):
    """Whatever happens to the model between two visits of a camera - small steps, large steps, opacity resets, Gaussians
    moved across the scene, a capacity hint that is far too small - the limited forward either passes its own check or is
    repeated, and the caller gets the bits of the un-limited forward."""
    P, W, H = 20000, 640, 480
    g = torch.Generator().manual_seed(100 + seed)
    sc = synthetic.trained_like(P, seed=seed, sh_degree=1)
    cam = device_camera(synthetic.orbit_cameras(W, H)[2 + 5 * seed])
    bg = torch.tensor([0.1, 0.0, 0.2])
    forward_state(hip, sc, cam, DEV, bg, False)
    for step, (dpos, dop, dscale) in enumerate([(1e-3, 0.01, 0.01), (1e-2, 0.1, 0.05), (0.1, 0.5, 0.3), (1e-3, 0.0, 0.0),
                                                (0.5, 0.9, 0.5), (1e-4, 0.02, 0.0)]):
        sc = dict(sc, means3D=sc["means3D"] + dpos * torch.randn((P, 3), generator=g),
                  opacities=(sc["opacities"] * (1 + dop * (2 * torch.rand((P, 1), generator=g) - 1))).clamp(1e-3, 0.999),
                  scales=sc["scales"] * torch.exp(dscale * torch.randn((P, 3), generator=g)))
        if step == 3:
            sc["opacities"] = sc["opacities"].clamp(max=0.01)  # an opacity reset
        if step == 4:
            hip._capacity_hint_limited = 4096
        got = forward_state(hip, sc, cam, DEV, bg, False)
        hip.depth_limit_on = False
        ref = forward_state(hip, sc, cam, DEV, bg, False)
        hip.depth_limit_on = True
        for k in ("color", "invdepth", "final_T", "radii", "n_contrib"):
            assert torch.equal(got[k], ref[k]), (step, k)
        assert torch.equal(last_contributor_id(got, W, H), last_contributor_id(ref, W, H)), step


def test_foreign_limits_cannot_change_a_result(hip):
    """Limits of ANOTHER camera planted under this camera's key (what an address re-used by the allocator would do), and
    absurd ones (everything cut): the image is still the un-limited image."""
    P, W, H = 20000, 400, 400
    sc = synthetic.trained_like(P, seed=6, sh_degree=0)
    cams = [device_camera(c) for c in synthetic.orbit_cameras(W, H)[:9:8]]
    bg = torch.zeros(3)
    hip.depth_limit_on = False
    ref = forward_state(hip, sc, cams[1], DEV, bg, False)
    hip.depth_limit_on = True
    forward_state(hip, sc, cams[0], DEV, bg, False)
    forward_state(hip, sc, cams[1], DEV, bg, False)
    e0 = hip.camera_entry(W, H, viewmatrix=cams[0].world_view_transform)
    e1 = hip.camera_entry(W, H, viewmatrix=cams[1].world_view_transform)
    assert e0 is not None and e1 is not None and e0 is not e1
    for planted in (e0["limit"].clone(), torch.full_like(e1["limit"], 1e-3)):
        e1["limit"].copy_(planted)
        e1["limit_ok"] = True
        out = forward_state(hip, sc, cams[1], DEV, bg, False)
        for k in ("color", "invdepth", "final_T", "radii"):
            assert torch.equal(out[k], ref[k]), k
        assert torch.equal(last_contributor_id(out, W, H), last_contributor_id(ref, W, H))


def test_per_camera_state_survives_recreated_camera_tensors(hip):
    """A caller that builds fresh camera tensors for every render (as the reference's Camera objects on another device
    would be) still gets its hints and limits: cameras are told apart by an explicit GaussianRasterizer.camera_key or,
    without one, by the CONTENTS of the view matrix - never by an address the allocator may hand to the next camera."""
    from helpers import settings_for
    P, W, H = 20000, 400, 400
    sc = synthetic.trained_like(P, seed=6, sh_degree=0)
    cams = synthetic.orbit_cameras(W, H)[:9:8]
    bg = torch.zeros(3)
    dev_scene = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in sc.items()}

    def render_fresh(cam, key=None):
        c = cam._replace(world_view_transform=cam.world_view_transform.to(DEV).clone(),
                         full_proj_transform=cam.full_proj_transform.to(DEV).clone(), camera_center=cam.camera_center.to(DEV).clone())
        rast = dgr.GaussianRasterizer(settings_for(dgr.GaussianRasterizationSettings, c, bg, 0, DEV))
        if key is not None:
            rast.camera_key = key
        with torch.no_grad():
            color, _, _ = rast(means3D=dev_scene["means3D"], means2D=torch.zeros_like(dev_scene["means3D"]),
                               opacities=dev_scene["opacities"], shs=dev_scene["shs"], scales=dev_scene["scales"],
                               rotations=dev_scene["rotations"])
        torch.cuda.synchronize()
        return color.clone()

    for keyed in (False, True):
        hip._cam_cache.clear()
        used0, st0 = hip.depth_limit_stats["used"], dict(hip.camera_cache_stats)
        imgs = {0: [], 1: []}
        for visit in range(3):
            for ci in (0, 1):
                imgs[ci].append(render_fresh(cams[ci], ("view", ci) if keyed else None))
        st = {k: hip.camera_cache_stats[k] - st0[k] for k in st0}
        print("keyed" if keyed else "content hash", st, "limited", hip.depth_limit_stats["used"] - used0)
        assert st["misses"] == 2 and st["hits"] == 4            # two cameras, each found again on visits 2 and 3
        assert hip.depth_limit_stats["used"] - used0 == 4        # ... and rendered with its limits then
        assert st["hashed"] == (0 if keyed else 6)               # a key saves the 64-byte read-back per fresh tensor
        for ci in (0, 1):
            assert torch.equal(imgs[ci][0], imgs[ci][1]) and torch.equal(imgs[ci][0], imgs[ci][2])
    # one tensor per camera (a trainer): hashed once per camera, not per visit
    hip._cam_cache.clear()
    kept = [device_camera(c) for c in cams]
    h0 = hip.camera_cache_stats["hashed"]
    for visit in range(3):
        for c in kept:
            forward_state(hip, sc, c, DEV, bg, False)
    assert hip.camera_cache_stats["hashed"] - h0 == 2


@pytest.mark.parametrize("mode", ["checked_in_the_forward", "deferred"])
def test_training_with_limits_is_the_same_run(hip, mode, step_form):
    a, b = make(hip), make(hip)
    hip.depth_limit_on = False
    la = [float(a.step(k)) for k in range(16)]
    hip.depth_limit_on = mode != "deferred"
    if mode == "deferred":
        b.depth_limit = "deferred"   # the trainer asks per step and collects the verdict one step later
    hip._cam_cache.clear()
    used0, failed0 = hip.depth_limit_stats["used"], hip.depth_limit_stats["failed"]
    lb = [b.step(k) for k in range(16)]
    b.sync()
    lb = [float(x) for x in lb]
    used, failed = hip.depth_limit_stats["used"] - used0, hip.depth_limit_stats["failed"] - failed0
    print("limited views", used, "fallbacks", failed, "\n", la, "\n", lb)
    assert used == 12 and failed <= 3  # 4 cameras: every visit after the first is limited; natural fall-backs are rare
    # same tolerance as two eager runs of one path (float-atomic order -> Adam sign flips, see test_gpu_fused_step.py)
    assert max(abs(x - y) for x, y in zip(la, lb)) <= 1e-3 * max(la)
    d = (a.model.flat - b.model.flat).double()
    assert float(d.pow(2).mean().sqrt()) <= 1e-4 * float(a.model.flat.double().pow(2).mean().sqrt())
    assert torch.equal(a.model.denom, b.model.denom)


def test_deferred_verdict_redoes_a_step_whose_limits_failed(hip, step_form):
    """Deferred mode: camera 1's limits are sabotaged between two visits.  The step that used them is a no-op on the
    device; one step later the trainer learns it, puts counters and running mean back, repeats the step with full
    lists and overwrites the loss it had handed out.  The run is the un-limited run."""
    a, b = make(hip), make(hip)
    hip.depth_limit_on = False
    b.depth_limit = "deferred"
    la = [float(a.step(k)) for k in range(12)]
    lb = [b.step(k) for k in range(5)]                    # cameras 0 1 2 3 0
    b.sync()
    hip.camera_entry(480, 320, camera_key=("trainer", b.uid, 1))["limit"].fill_(1e-3)  # everything of camera 1 is cut on its next visit
    failed0 = hip.depth_limit_stats["failed"]
    before = b.model.flat.detach().clone()
    lb.append(b.step(5))                                  # camera 1 with useless limits: nothing may change
    torch.cuda.synchronize()
    assert torch.equal(before, b.model.flat.detach()) and b.model.optimizer.t == 6
    garbage = float(lb[5])
    lb += [b.step(k) for k in range(6, 12)]               # step 6 first settles step 5 (redo), then runs
    b.sync()
    assert hip.depth_limit_stats["failed"] >= failed0 + 1 and b.model.optimizer.t == 12
    lb = [float(x) for x in lb]
    print("loss of the invalid image %.6f -> after the redo %.6f (un-limited run %.6f)" % (garbage, lb[5], la[5]))
    assert max(abs(x - y) for x, y in zip(la, lb)) <= 1e-3 * max(la)
    d = (a.model.flat - b.model.flat).double()
    assert float(d.pow(2).mean().sqrt()) <= 1e-4 * float(a.model.flat.double().pow(2).mean().sqrt())
    assert torch.equal(a.model.denom, b.model.denom)


def test_graphed_step_with_stale_limits_falls_back(hip, step_form):
    from gsplat_amd.trainer import GraphedStep
    make(hip).step(0)  # (a first eager view sizes the capacity hint GraphedStep captures with)
    torch.cuda.synchronize()
    a, b = make(hip), make(hip)
    b.depth_limit = "deferred"
    gs = GraphedStep(b)
    lb = [float(gs.step(k)) for k in range(9)]          # every camera's capture (0: 4 steps; 1 2 3: 2 each), then 0 1 2 3 0
    assert gs.eager_steps == 0 and gs.captures == 4
    assert all(gs.camera_entry(c)["limit_ok"] for c in range(4))
    gs.camera_entry(1)["limit"].fill_(1e-3)              # camera 1's limits now cut everything
    gs.step(9)                                           # -> flagged; the host learns it when it settles the replay
    gs.sync()                                            #    (before the next step, or here): undone, stepped eagerly
    lb.append(float(gs.s_loss))
    assert gs.eager_steps == 1 and gs.captures == 4      # (camera 1's graph stays: the eager step measured new limits)
    lb += [float(gs.step(k)) for k in range(10, 14)]     # replays; camera 1 (step 13) on its re-learnt limits
    gs.sync()
    assert gs.eager_steps == 1 and gs.captures == 4 and gs.camera_entry(1)["limit_ok"]
    cams = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3] + [k % 4 for k in range(4, 14)]
    hip.depth_limit_on = False
    la = [float(a._step_camera(c, True, ())) for c in cams]
    hip.depth_limit_on = True
    assert a.model.optimizer.t == b.model.optimizer.t
    d = (a.model.flat - b.model.flat).double()
    assert float(d.pow(2).mean().sqrt()) <= 1e-4 * float(a.model.flat.double().pow(2).mean().sqrt())
    assert abs(la[-1] - lb[-1]) <= 1e-3 * la[-1]


def test_status_block_from_the_forwards_last_kernel_is_the_copied_one(hip):
    """GsScratch.status_host (the forward's last kernel writes the status words into the pinned block) against
    gs_forward_status (a copy command behind the forward): the deferred, depth-limited run - verdicts, fall-backs, every
    parameter bit - is the same run."""
    runs = []
    old = hip.STATUS_IN_RENDER
    try:
        for flag in (True, False):
            hip.STATUS_IN_RENDER = flag
            hip._cam_cache.clear()
            t = make(hip)
            t.depth_limit = "deferred"
            used0, failed0 = hip.depth_limit_stats["used"], hip.depth_limit_stats["failed"]
            losses = [t.step(k) for k in range(14)]
            t.sync()
            runs.append(([float(x) for x in losses], t.model.flat.detach().clone(), t.model.denom.clone(),
                         hip.depth_limit_stats["used"] - used0, hip.depth_limit_stats["failed"] - failed0))
    finally:
        hip.STATUS_IN_RENDER = old
    a, b = runs
    assert a[3] == b[3] >= 8   # (fall-backs may differ by one: the first run also sizes the binning capacity of limited views)
    assert a[0] == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
