"""Multispectral (RGB + NIR) train step (BASELINE config 5, counterpart of mult-dwtgs/train_nir.py): plumbing on the
CPU oracle with the reference's two rasterizer passes; on the GPU the fused 4-channel pass must give the same step."""
import pytest
import torch

from gsplat_amd import synthetic
from gsplat_amd.losses import LossOps
from gsplat_amd.trainer import GaussianModelLite, NirCriterion, TrainOptions, TrainerNIR, camera_to


def make(api, Settings, device, two_pass, P=400, W=96, H=64, seed=0):
    sc = synthetic.trained_like(P, seed=seed, scale_mult=1.5)
    cams = [camera_to(c, device) for c in synthetic.orbit_cameras(W, H)[:3]]
    g = torch.Generator().manual_seed(7)
    gts = [torch.rand((3, H, W), generator=g).to(device) for _ in cams]
    nirs = [torch.rand((1, H, W), generator=g).to(device) for _ in cams]
    model = GaussianModelLite(sc, device, api=api, with_nir=True)
    crit = NirCriterion(LossOps(api))
    return TrainerNIR(model, cams, gts, nirs, crit, Settings, torch.zeros(3, device=device), two_pass_rasterizer=two_pass)


def test_nir_step_on_the_oracle_with_two_passes(oracle):
    tr = make(oracle.api, oracle.Settings, torch.device("cpu"), oracle.Rasterizer)
    m = tr.model
    assert m.width == 60 and m.flat.numel() == 400 * 60 and m.params["nir_albedo"].shape == (400, 1)
    assert torch.equal(m.params["nir_albedo"].detach()[:, 0], m.params["features"].detach()[:, 0, 0])
    before = m.flat.clone()
    l0 = float(tr.step(0))
    gv = m.grad_views()
    assert float(gv["nir_albedo"].abs().sum()) > 0 and m.nir_gain.grad is not None and float(m.nir_gain.grad.abs()) > 0
    assert not torch.equal(before[-400:], m.flat[-400:]) and float(m.nir_gain) != 1.0
    losses = [l0] + [float(tr.step(k)) for k in range(3, 19, 3)]  # camera 0 every third step
    assert losses[-1] < losses[0]
    # densification carries the 60th column and its moments along
    m.xyz_gradient_accum += 1e-3
    nc, ns, npr = m.densify_and_prune(2e-4, 0.005, 4.4, None, None, generator=torch.Generator().manual_seed(1))
    assert nc + ns > 0 and m.flat.numel() == m.P * 60 and m.params["nir_albedo"].shape == (m.P, 1)
    opt = TrainOptions(iterations=10, densify_from_iter=100, cameras_extent=4.4)
    gain0 = float(m.nir_gain)
    out = tr.train_iteration(1, opt)
    assert out["P"] == m.P and torch.isfinite(out["loss"])
    assert float(m.nir_gain) != gain0  # the schedule loop steps the global gain with the main optimizer
    # ... but not on an iteration that densifies (the reference's optimizer.step() then changes nothing)
    m.xyz_gradient_accum += 1e-3
    gain1 = float(m.nir_gain)
    out = tr.train_iteration(200, TrainOptions(iterations=1000, densify_from_iter=100, cameras_extent=4.4))
    assert out["densified"] is not None and float(m.nir_gain) == gain1
    # checkpoints carry the gain and its Adam moments
    state = m.capture()
    tr.train_iteration(201, TrainOptions(iterations=1000, densify_from_iter=100, cameras_extent=4.4))
    assert float(m.nir_gain) != gain1
    m.restore(state)
    assert float(m.nir_gain) == gain1
    assert float(m.nir_gain_optimizer.state_dict()["state"][0]["exp_avg"]) == float(state["nir_gain_optimizer"]["state"][0]["exp_avg"])
    # a checkpoint written before the gain's step count moved into seg_steps restores too (the count is migrated)
    steps = int(state["seg_steps"]["nir_gain"])
    old_state = dict(state, seg_steps={k: v for k, v in state["seg_steps"].items() if k != "nir_gain"})
    m.restore(old_state)
    assert m.optimizer.seg_steps["nir_gain"] == int(state["nir_gain_optimizer"]["state"][0]["step"])
    assert abs(m.optimizer.seg_steps["nir_gain"] - steps) <= 1
    out = tr.train_iteration(202, TrainOptions(iterations=1000, densify_from_iter=1000, cameras_extent=4.4))
    assert torch.isfinite(out["loss"]) and float(m.nir_gain) != gain1


@pytest.mark.gpu
def test_fused_four_channel_step_equals_two_pass_step(hip):
    import diff_gaussian_rasterization as dgr
    dev = torch.device("cuda")
    a = make(hip.api, dgr.GaussianRasterizationSettings, dev, None, P=5000, W=320, H=240)
    b = make(hip.api, dgr.GaussianRasterizationSettings, dev, dgr.GaussianRasterizer, P=5000, W=320, H=240)
    a.optimizer_step = b.optimizer_step = False
    la, lb = a.step(1), b.step(1)
    assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(lb))
    ga, gb = a.model.flat_grad, b.model.flat_grad
    assert float((ga - gb).abs().max()) <= 1e-4 * float(gb.abs().max())
    assert abs(float(a.model.nir_gain.grad) - float(b.model.nir_gain.grad)) <= 1e-4 * abs(float(b.model.nir_gain.grad))
    assert torch.equal(a.model.denom, b.model.denom) and torch.equal(a.model.max_radii2D, b.model.max_radii2D)
    assert torch.allclose(a.model.xyz_gradient_accum, b.model.xyz_gradient_accum, rtol=1e-4, atol=1e-9)
    a.optimizer_step = True
    l = [float(a.step(k)) for k in range(0, 30, 3)]
    assert l[-1] < l[0]


def test_nir_criterion_matches_the_reference_loss_functions(oracle):
    """Golden from the reference's own l1_loss / ssim / combined_nir_loss (tests/golden/nir_loss.npz, generator
    make_golden.py): value and both image gradients."""
    import os

    import numpy as np
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "nir_loss.npz"))
    image = torch.from_numpy(z["image"]).requires_grad_(True)
    nir = torch.from_numpy(z["nir"]).requires_grad_(True)
    crit = NirCriterion(LossOps(oracle.api))
    total, parts = crit(image, torch.from_numpy(z["gt"]), nir, torch.from_numpy(z["nir_gt"]))
    total.backward()
    assert abs(float(parts["rgb"]) - float(z["rgb_loss"])) < 2e-6 and abs(float(parts["nir"]) - float(z["nir_loss"])) < 2e-6
    assert abs(float(total) - float(z["total"])) < 3e-6
    assert float((image.grad - torch.from_numpy(z["d_image"])).abs().max()) < 1e-7 + 2e-4 * float(np.abs(z["d_image"]).max())
    assert float((nir.grad - torch.from_numpy(z["d_nir"])).abs().max()) < 1e-7 + 2e-4 * float(np.abs(z["d_nir"]).max())


# ---- the fused multispectral step (round 4): gs_backward_step_x, fused criterion nodes, depth limits, two phases, graph
def _make_nir(hip, fused, P=30000, W=480, H=320, seed=3, dwt=False, two_pass=None):
    import diff_gaussian_rasterization as dgr
    import lgdwt_loss
    from simple_knn._C import distCUDA2
    dev = torch.device("cuda")
    sc = synthetic.trained_like(P, seed=seed, knn=lambda x: distCUDA2(x.to(dev)).cpu())
    cams = [camera_to(c, dev) for c in synthetic.orbit_cameras(W, H)[:4]]
    g = torch.Generator().manual_seed(5)
    gts = [torch.rand((3, H, W), generator=g).to(dev) for _ in cams]
    nirs = [torch.rand((1, H, W), generator=g).to(dev) for _ in cams]
    model = GaussianModelLite(sc, dev, api=hip.api, with_nir=True)
    rgb = lgdwt_loss.criterion(dwt_enable=True, patch_dwt_enable=True, fused=fused) if dwt else None
    masks = [rgb.elf_mask(x) for x in gts] if dwt else None
    crit = NirCriterion(LossOps(hip.api), rgb_criterion=rgb, fused=fused)
    return TrainerNIR(model, cams, gts, nirs, crit, dgr.GaussianRasterizationSettings, torch.zeros(3, device=dev),
                      two_pass_rasterizer=two_pass, masks=masks)


def _nir_state(tr):
    m, o = tr.model, tr.model.optimizer
    gs = m.nir_gain_optimizer.state[m.nir_gain]
    return dict(flat=m.flat.detach().clone(), exp_avg=o.exp_avg.clone(), exp_avg_sq=o.exp_avg_sq.clone(),
                gain=m.nir_gain.detach().clone().reshape(1), gain_m=gs["exp_avg"].clone().reshape(1),
                gain_v=gs["exp_avg_sq"].clone().reshape(1), accum=m.xyz_gradient_accum.clone(), denom=m.denom.clone(),
                max_radii=m.max_radii2D.clone())


@pytest.mark.gpu
@pytest.mark.parametrize("dwt", [False, True], ids=["train_nir_loss", "lgdwt_rgb_loss"])
def test_fused_multispectral_step_tracks_the_unfused_one(hip, dwt):
    """TrainerNIR on the fused machinery - raw rows, fused criterion nodes, gs_backward_step_x with the 60th row and the gain
    stepped in the per-Gaussian kernel - against the round-3 step (activated copies, torch criterion, gs_backward_x,
    collect_grads, Adam kernel, torch Adam for the gain): same trajectory to rounding (the bar of
    tests/test_gpu_fused_step.py::test_fused_train_step_tracks_the_unfused_one), statistics equal."""
    a, b = _make_nir(hip, False, dwt=dwt), _make_nir(hip, True, dwt=dwt)
    assert not a._fused_step_ok(a._backend(), True) and b._fused_step_ok(b._backend(), True)
    la, lb = [], []
    for k in range(8):
        la.append(float(a.step(k % 4)))
        lb.append(float(b.step(k % 4)))
    assert la[-1] < la[0] and lb[-1] < lb[0]
    assert max(abs(x - y) for x, y in zip(la, lb)) <= 1e-3 * max(la), (la, lb)
    sa, sb = _nir_state(a), _nir_state(b)
    d = (sa["flat"] - sb["flat"]).double()
    assert float(d.pow(2).mean().sqrt()) <= 1e-4 * float(sa["flat"].double().pow(2).mean().sqrt())
    P = a.model.P
    dn = (sa["flat"][-P:] - sb["flat"][-P:]).double()       # the 60th row on its own (it is 1/60 of the buffer)
    assert float(dn.pow(2).mean().sqrt()) <= 1e-4 * float(sa["flat"][-P:].double().pow(2).mean().sqrt())
    assert float((sa["flat"][-P:] - _make_nir(hip, False, dwt=dwt).model.flat[-P:]).abs().max()) > 0   # ... and it moved
    assert abs(float(sa["gain"]) - float(sb["gain"])) <= 1e-5 and float(sb["gain"]) != 1.0
    assert abs(float(sa["gain_m"]) - float(sb["gain_m"])) <= 1e-3 * abs(float(sa["gain_m"])) + 1e-9
    assert torch.equal(sa["denom"], sb["denom"]) and torch.equal(sa["max_radii"], sb["max_radii"])
    assert a.model.optimizer.seg_steps == b.model.optimizer.seg_steps and b.model.optimizer.seg_steps["nir_gain"] == 8


@pytest.mark.gpu
def test_fused_multispectral_step_forms_agree_bit_for_bit(hip):
    """With the blend sums pinned (rows_override) the forms of the fused multispectral step leave the very same bits: one
    launch / two phases (side stream), eager / replayed from a hipGraph - all on depth-limited lists (which Gaussians read
    their - random - row at all is a property of the lists: tiles_touched != 0)."""
    from gsplat_amd.trainer import GraphedStep
    old = (hip.binning, hip._capacity_hint, hip._capacity_hint_limited)
    hip.binning = "region"
    hip._cam_cache.clear()
    try:
        trs = [_make_nir(hip, True) for _ in range(4)]   # one launch | two phases | two phases again | hipGraph
        P = trs[0].model.P
        g = torch.Generator().manual_seed(13)
        rows = torch.zeros((P, 16), dtype=torch.float64)
        rows[:, :11] = (torch.randn((P, 11), generator=g) * 1e-3).double()   # incl. the 4th channel's slot (GR_EXTRA = 10)
        rows = rows.cuda()
        for t in trs:
            t.rows_override, t.depth_limit = rows, "deferred"
        graphed = GraphedStep(trs[3], warmup=1)
        n0 = hip.two_phase_launches
        try:
            for k in range(9):
                hip.TWO_PHASE = False
                trs[0].step(k)
                hip.TWO_PHASE, hip.TWO_PHASE_MIN_P = True, 0
                trs[1].step(k)
                trs[2].step(k)
            for t in trs[:3]:
                t.sync()
        finally:
            del hip.TWO_PHASE, hip.TWO_PHASE_MIN_P
        assert hip.two_phase_launches - n0 >= 9
        s0 = _nir_state(trs[0])
        for t in trs[1:3]:
            s = _nir_state(t)
            for key in s0:
                assert torch.equal(s0[key], s[key]), (key, float((s0[key] - s[key]).abs().max()))
        assert float(s0["gain"]) != 1.0 and trs[0].model.optimizer.seg_steps["nir_gain"] == 9
        # the graph form: its first call per camera takes warm-up steps and a capture on that camera - the same camera
        # sequence eagerly (tests/test_gpu_fused_step.py::_camera_sequence) must leave the same bits
        from test_gpu_fused_step import _camera_sequence
        ref = _make_nir(hip, True)
        ref.rows_override = rows
        ref.depth_limit = "deferred"
        for k in range(8):
            graphed.step(k)
        graphed.sync()
        for c in _camera_sequence(8, warmup=1):
            ref._step_camera(c, True, ())
        ref.sync()
        torch.cuda.synchronize()
        assert graphed.captures == 4 and graphed.replays == 4 and graphed.eager_steps == 0
        sg, sr = _nir_state(trs[3]), _nir_state(ref)
        for key in sg:
            assert torch.equal(sg[key], sr[key]), (key, float((sg[key] - sr[key]).abs().max()))
        assert trs[3].model.optimizer.seg_steps == ref.model.optimizer.seg_steps
    finally:
        hip.binning, hip._capacity_hint, hip._capacity_hint_limited = old
        hip._cam_cache.clear()
