"""optimizer_type = "sparse_adam" (LGDWT-GS/arguments/__init__.py:101, train.py:68, 282-284):

    visible = radii > 0
    gaussians.optimizer.step(visible, radii.shape[0])

Only the Gaussians visible in the step's view are stepped; parameters and both moments of the others keep their bits.  (The
optimizer class itself, SparseGaussianAdam, lives in the 3dgs_accel branch of the rasterizer, which the reference does not
vendor - its pinned branch is dr_aa.)  Restated here as a MASKED torch.optim.Adam: torch's own step on every group, then the
rows of the invisible Gaussians - parameter, exp_avg, exp_avg_sq - put back.

CPU: gs_adam_step_masked (oracle build) and the train loop's use of it against that restatement.
GPU: the masked kernel against the oracle's; the FUSED step (gs_backward_step / gs_step_uninstanced with GsStepState.sparse) against
the un-fused tail, bit for bit; invisible rows untouched; the side stream's traffic shrinks."""
import pytest
import torch

from gsplat_amd import synthetic
from gsplat_amd.trainer import FIELDS, GaussianModelLite


def masked_torch_adam_step(b, visible):
    """one step of b's torch.optim.Adam (reference groups) applied to the visible rows only"""
    opt = b.optimizer
    tensors = [(b.params[n], None) for n in ("xyz", "opacity", "scaling", "rotation")] + [(opt.f_dc, None), (opt.f_rest, None)]
    before = []
    for p, _ in tensors:
        st = opt.opt.state.get(p, {})
        before.append((p.detach().clone(), None if "exp_avg" not in st else st["exp_avg"].clone(),
                       None if "exp_avg_sq" not in st else st["exp_avg_sq"].clone()))
    feat_before = b.params["features"].detach().clone()
    opt.step()
    keep = ~visible
    with torch.no_grad():
        for (p, _), (p0, m0, v0) in zip(tensors, before):
            st = opt.opt.state[p]
            p[keep] = p0[keep]
            st["exp_avg"][keep] = 0.0 if m0 is None else m0[keep]
            st["exp_avg_sq"][keep] = 0.0 if v0 is None else v0[keep]
        b.params["features"][keep] = feat_before[keep]


def flat_moments(b):
    """torch Adam's moments in the flat field-major layout of FlatAdam"""
    opt = b.optimizer
    P = b.P
    out = []
    for which in ("exp_avg", "exp_avg_sq"):
        def st(p):
            s = opt.opt.state.get(p)
            return torch.zeros_like(p) if not s else s[which]
        feat = torch.cat((st(opt.f_dc), st(opt.f_rest)), dim=1)
        parts = {"xyz": st(b.params["xyz"]), "features": feat, "opacity": st(b.params["opacity"]),
                 "scaling": st(b.params["scaling"]), "rotation": st(b.params["rotation"])}
        out.append(torch.cat([parts[n].reshape(P, w).reshape(-1) for n, w in FIELDS]))
    return out


def test_masked_flat_adam_equals_masked_torch_adam(oracle):
    P = 311
    sc = synthetic.trained_like(P, seed=4)
    a = GaussianModelLite(sc, torch.device("cpu"), api=oracle.api)
    b = GaussianModelLite(sc, torch.device("cpu"), api=None)
    a.optimizer.sparse = True
    g = torch.Generator().manual_seed(0)
    start = a.flat.clone()
    never = torch.zeros(P, dtype=torch.bool)
    never[::11] = True                                      # Gaussians no view ever sees
    for it in range(1, 7):
        grad = torch.randn(a.flat.numel(), generator=g) * torch.logspace(-6, 0, a.flat.numel())
        visible = (torch.rand(P, generator=g) > 0.4) & ~never
        a.flat_grad.copy_(grad)
        b.flat_grad.copy_(grad)
        assert a.update_learning_rate(it) == b.update_learning_rate(it)
        skip = ("opacity",) if it == 4 else ()               # (a group without a gradient: reset_opacity iterations)
        for n_, p_ in b.params.items():
            p_.grad = None if n_ in skip else b.grad_views()[n_]
        a.optimizer.step(skip, row_mask=visible.float())
        masked_torch_adam_step(b, visible)
    assert float((a.flat - b.flat).abs().max()) < 2e-6 and float((a.flat - start).abs().max()) > 1e-3
    m, v = flat_moments(b)
    assert float((a.optimizer.exp_avg - m).abs().max()) < 2e-6 * float(m.abs().max())
    # (the kernels form 1 - beta2 in float - 1.3e-5 below torch's double 0.001 - as tests/test_adam.py's unmasked comparison has it)
    assert float((a.optimizer.exp_avg_sq - v).abs().max()) < 5e-5 * float(v.abs().max())
    # rows never visible: not a bit has moved, moments still +0
    rows = a.optimizer.field_views(a.flat)
    rows0 = a.optimizer.field_views(start)
    for n, _ in FIELDS:
        assert torch.equal(rows[n][never], rows0[n][never]), n
        assert not a.optimizer.field_views(a.optimizer.exp_avg)[n][never].any()
    # the default optimizer does move them (momentum of a zero gradient is zero here, but the moments are touched: v stays 0) -
    # what differs is a row that WAS visible once: the default keeps stepping it on momentum, sparse_adam freezes it
    c = GaussianModelLite(sc, torch.device("cpu"), api=oracle.api)
    d = GaussianModelLite(sc, torch.device("cpu"), api=oracle.api)
    d.optimizer.sparse = True
    grad = torch.randn(c.flat.numel(), generator=g) * 1e-3
    for m_ in (c, d):
        m_.flat_grad.copy_(grad)
    c.optimizer.step()
    d.optimizer.step((), row_mask=torch.ones(P))
    assert torch.equal(c.flat, d.flat)                       # everything visible: the same step
    for m_ in (c, d):
        m_.flat_grad.zero_()
    c.optimizer.step()
    d.optimizer.step((), row_mask=torch.zeros(P))
    assert not torch.equal(c.flat, d.flat)
    assert torch.equal(d.optimizer.exp_avg, c.optimizer.exp_avg / 0.9) or float((d.optimizer.exp_avg * 0.9 - c.optimizer.exp_avg).abs().max()) < 1e-9


def test_train_loop_steps_only_the_visible_gaussians(oracle):
    from test_trainer_cpu import make_trainer
    a = make_trainer(oracle, P=400, W=96, H=64)
    b = make_trainer(oracle, P=400, W=96, H=64)
    b.__init__(b.model, b.cameras, b.gts, b.criterion, b.Rasterizer, b.Settings, b.bg, optimizer_type="sparse_adam")
    assert b.model.optimizer.sparse and not a.model.optimizer.sparse
    with torch.no_grad():   # a fifth of the Gaussians far above the scene: outside every frustum
        for t in (a, b):
            t.model.params["xyz"][::5, 2] += 50.0
    start = b.model.flat.clone()
    for k in range(6):
        la, lb = float(a.step(k)), float(b.step(k))
    hidden = torch.zeros(400, dtype=torch.bool)
    hidden[::5] = True
    rows, rows0 = b.model.optimizer.field_views(b.model.flat), b.model.optimizer.field_views(start)
    for n, _ in FIELDS:
        assert torch.equal(rows[n][hidden], rows0[n][hidden]), n
    assert not torch.equal(b.model.flat, start) and float(b.model.denom[hidden].max()) == 0
    # one step against the restatement: same gradients, masked torch Adam
    from gsplat_amd.trainer import render
    c = make_trainer(oracle, P=400, W=96, H=64)
    c.__init__(c.model, c.cameras, c.gts, c.criterion, c.Rasterizer, c.Settings, c.bg, optimizer_type="sparse_adam")
    t = GaussianModelLite({k: v for k, v in synthetic.trained_like(400, seed=0, scale_mult=1.5).items()}, torch.device("cpu"), api=None)
    with torch.no_grad():
        t.flat.copy_(c.model.flat)
    c.step(1)
    t.flat_grad.copy_(c.model.flat_grad)
    for n, p in t.params.items():
        p.grad = t.grad_views()[n]
    masked_torch_adam_step(t, c.last_radii > 0)
    assert float((c.model.flat - t.flat).abs().max()) < 2e-6


@pytest.mark.gpu
def test_hip_masked_adam_matches_the_oracle(hip, oracle):
    P = 5003
    sc = synthetic.trained_like(P, seed=4)
    a = GaussianModelLite(sc, torch.device("cuda"), api=hip.api)
    o = GaussianModelLite(sc, torch.device("cpu"), api=oracle.api)
    a.optimizer.sparse = o.optimizer.sparse = True
    g = torch.Generator().manual_seed(1)
    for it in range(1, 6):
        grad = torch.randn(o.flat.numel(), generator=g) * 1e-3
        vis = (torch.rand(P, generator=g) > 0.5).float()
        a.flat_grad.copy_(grad.cuda())
        o.flat_grad.copy_(grad)
        a.optimizer.step((), row_mask=vis.cuda())
        o.optimizer.step((), row_mask=vis)
    assert float((a.flat.cpu() - o.flat).abs().max()) < 1e-6
    assert float((a.optimizer.exp_avg.cpu() - o.optimizer.exp_avg).abs().max()) < 1e-8
    assert float((a.optimizer.exp_avg_sq.cpu() - o.optimizer.exp_avg_sq).abs().max()) < 1e-9
    assert torch.equal((a.optimizer.exp_avg == 0).cpu(), o.optimizer.exp_avg == 0)   # the same rows were never touched


@pytest.mark.gpu
@pytest.mark.parametrize("lists", ["culled", "limited", "reference"])
@pytest.mark.parametrize("two_phase", [False, True])
def test_fused_sparse_step_equals_the_unfused_tail_bit_for_bit(hip, lists, two_phase):
    """gs_backward_step (+ gs_step_uninstanced) with GsStepState.sparse against gs_backward + gs_activations_bwd +
    gs_densify_stats + gs_adam_step_masked on the same blend sums: the very same bits in parameters, moments, statistics;
    and the Gaussians outside the frustum (a third of them, lifted above the scene) keep theirs."""
    from test_gpu_fused_step import make, state
    a = make(hip, False, P=120000, W=480, H=320, lifted=0.33)
    b = make(hip, True, P=120000, W=480, H=320, lifted=0.33)
    for t in (a, b):
        t.__init__(t.model, t.cameras, t.gts, t.criterion, t.Rasterizer, t.Settings, t.bg, optimizer_step=True,
                   optimizer_type="sparse_adam")
    a.FUSED_STEP, b.FUSED_STEP = False, True
    P = a.model.P
    start = state(a)
    old_cull = hip.tile_cull
    if lists == "reference":
        hip.tile_cull = False
    if lists == "limited":
        b.depth_limit = "deferred"
    hip.TWO_PHASE, hip.TWO_PHASE_MIN_P = two_phase, 0
    n0 = hip.two_phase_launches
    try:
        for it in range(6):
            hip.keep_workspace = True
            try:
                a._step_camera(it % 4, True, ())
                torch.cuda.synchronize()
                rows = hip.last_workspace[: P * 128].view(torch.float64).clone()
            finally:
                hip.keep_workspace, hip.last_workspace = False, None
            if lists == "limited":
                # (depth-limited lists drop instances the un-limited run has: the pinned sums then hold entries for Gaussians
                #  the limited view lists no instance of - their rows must read as zero, as the limited blend would leave them)
                b.rows_override = None
                b._step_camera(it % 4, True, ())
                b.sync()
                continue
            b.rows_override = rows
            b._step_camera(it % 4, True, ())
            torch.cuda.synchronize()
            sa, sb = state(a), state(b)
            for k in sa:
                assert torch.equal(sa[k], sb[k]), (it, k, float((sa[k] - sb[k]).abs().max()))
    finally:
        del hip.TWO_PHASE, hip.TWO_PHASE_MIN_P
        hip.tile_cull = old_cull
    if two_phase:
        assert hip.two_phase_launches - n0 >= 6
    sb = state(b)
    lifted = slice(0, int(P * 0.33))
    W = sb["flat"].numel() // P
    off = 0
    for n, w in FIELDS:
        for key in ("flat", "exp_avg", "exp_avg_sq"):
            x, x0 = sb[key][off:off + P * w].view(P, w), start[key][off:off + P * w].view(P, w)
            assert torch.equal(x[lifted], x0[lifted]), (n, key)
        off += P * w
    assert W == 59 and not torch.equal(sb["flat"], start["flat"])
    if lists == "limited":   # (own blend sums on both sides: the trajectories agree to rounding)
        sa = state(a)
        d = (sa["flat"] - sb["flat"]).double()
        assert float(d.pow(2).mean().sqrt()) <= 1e-4 * float(sa["flat"].double().pow(2).mean().sqrt())
