"""Densification, opacity reset and the iteration schedule of the build-owned train loop on the CPU.

`densify_and_prune` works on the flat parameter / Adam buffers in one re-layout; it is pinned here against a
step-by-step restatement of the reference's sequence of tensor operations (LGDWT-GS/scene/gaussian_model.py:
331-467: densify_and_clone -> cat, densify_and_split -> cat -> prune_points, final prune_points), written with
separate per-group tensors and `torch.cat` / boolean indexing exactly in the reference's order.  The optimizer
skip semantics (a replaced parameter has no gradient: torch skips it and its step counter) are pinned against
torch.optim.Adam itself.
"""
import torch

from gsplat_amd import synthetic
from gsplat_amd.trainer import FIELDS, GaussianModelLite, TrainOptions, cameras_extent
from test_trainer_cpu import make_trainer


def fresh_model(oracle, P=400, seed=3):
    sc = synthetic.trained_like(P, seed=seed, scale_mult=1.5)
    m = GaussianModelLite(sc, torch.device("cpu"), api=oracle.api)
    g = torch.Generator().manual_seed(seed)
    for it in range(1, 4):  # non-trivial Adam moments
        m.flat_grad.copy_(torch.randn(m.flat.numel(), generator=g) * 1e-2)
        m.optimizer.step()
    m.xyz_gradient_accum = torch.rand((P, 1), generator=g) * 6e-4
    m.denom = torch.randint(0, 3, (P, 1), generator=g).float()  # zeros -> NaN grads -> 0 (gaussian_model.py:446)
    m.max_radii2D = torch.rand((P,), generator=g) * 50
    return m


def reference_sequence(fields, mom1, mom2, accum, denom, extent, max_grad, min_opacity, max_screen_size, noise,
                       percent_dense=0.01, N=2):
    """The reference's operations, one after the other, on dicts of per-field tensors."""
    f = {k: v.clone() for k, v in fields.items()}
    m1 = {k: v.clone() for k, v in mom1.items()}
    m2 = {k: v.clone() for k, v in mom2.items()}

    def get_scaling():
        return torch.exp(f["scaling"])

    def cat(new):  # cat_tensors_to_optimizer + densification_postfix
        for k in f:
            f[k] = torch.cat((f[k], new[k]), dim=0)
            m1[k] = torch.cat((m1[k], torch.zeros_like(new[k])), dim=0)
            m2[k] = torch.cat((m2[k], torch.zeros_like(new[k])), dim=0)

    def prune(mask):  # prune_points
        keep = ~mask
        for k in f:
            f[k], m1[k], m2[k] = f[k][keep], m1[k][keep], m2[k][keep]

    grads = accum / denom
    grads[grads.isnan()] = 0.0
    # densify_and_clone
    sel = torch.where(torch.norm(grads, dim=-1) >= max_grad, True, False)
    sel = torch.logical_and(sel, torch.max(get_scaling(), dim=1).values <= percent_dense * extent)
    cat({k: f[k][sel] for k in f})
    # densify_and_split
    n_init = f["xyz"].shape[0]
    padded = torch.zeros((n_init,))
    padded[:grads.shape[0]] = grads.squeeze()
    sel = torch.where(padded >= max_grad, True, False)
    sel = torch.logical_and(sel, torch.max(get_scaling(), dim=1).values > percent_dense * extent)
    stds = get_scaling()[sel].repeat(N, 1)
    samples = torch.randn((stds.shape[0], 3), generator=noise) * stds  # torch.normal(mean=0, std=stds)
    q = f["rotation"][sel]
    q = q / q.norm(dim=1, keepdim=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.zeros((q.shape[0], 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - r * z); R[:, 0, 2] = 2 * (x * z + r * y)
    R[:, 1, 0] = 2 * (x * y + r * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - r * x)
    R[:, 2, 0] = 2 * (x * z - r * y); R[:, 2, 1] = 2 * (y * z + r * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    rots = R.repeat(N, 1, 1)
    new = {k: f[k][sel].repeat(N, 1) for k in f}
    new["xyz"] = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + f["xyz"][sel].repeat(N, 1)
    new["scaling"] = torch.log(get_scaling()[sel].repeat(N, 1) / (0.8 * N))
    n_sel = int(sel.sum())
    cat(new)
    prune(torch.cat((sel, torch.zeros(N * n_sel, dtype=torch.bool))))
    max_radii2D = torch.zeros((f["xyz"].shape[0],))  # densification_postfix zeroes it (:383)
    # final prune
    pm = (torch.sigmoid(f["opacity"]) < min_opacity).squeeze()
    if max_screen_size:
        big_vs = max_radii2D > max_screen_size
        big_ws = get_scaling().max(dim=1).values > 0.1 * extent
        pm = torch.logical_or(torch.logical_or(pm, big_vs), big_ws)
    prune(pm)
    return f, m1, m2


def test_densify_and_prune_equals_the_reference_sequence(oracle):
    for max_screen in (None, 20):
        m = fresh_model(oracle)
        opt = m.optimizer
        P = m.P
        fields = {n: m.params[n].detach().reshape(P, w).clone() for n, w in FIELDS}
        mom1 = {n: v.clone() for n, v in opt.field_views(opt.exp_avg).items()}
        mom2 = {n: v.clone() for n, v in opt.field_views(opt.exp_avg_sq).items()}
        with torch.no_grad():
            m.params["opacity"][::9] = -6.0   # some nearly transparent Gaussians (pruned)
            fields["opacity"][::9] = -6.0
            m.params["scaling"][::31] = 0.2   # some huge ones (world-size prune when a screen size is given)
            fields["scaling"][::31] = 0.2
            m.params["scaling"][::5] = -4.6   # small ones (max scale 0.01 <= percent_dense * extent): cloned
            fields["scaling"][::5] = -4.6
        extent = 4.4
        gen = torch.Generator().manual_seed(77)
        noise = torch.Generator().manual_seed(77)
        rf, r1, r2 = reference_sequence(fields, mom1, mom2, m.xyz_gradient_accum.clone(), m.denom.clone(), extent,
                                        2e-4, 0.005, max_screen, noise)
        steps_before = dict(opt.seg_steps)
        nc, ns, npr = m.densify_and_prune(2e-4, 0.005, extent, max_screen, None, generator=gen)
        assert nc > 0 and ns > 0 and npr > 0, (nc, ns, npr)
        assert m.P == rf["xyz"].shape[0] and m.flat.numel() == m.P * 59
        v1, v2 = opt.field_views(opt.exp_avg), opt.field_views(opt.exp_avg_sq)
        for n, w in FIELDS:
            # (rows are copies except the split samples' centres and scales, which both sides COMPUTE - torch may vectorise
            #  exp / bmm differently for the two call shapes on some CPUs: a last-bit tolerance there, copies stay exact)
            got = m.params[n].detach().reshape(m.P, w)
            if n in ("xyz", "scaling"):
                assert torch.allclose(got, rf[n], rtol=2e-6, atol=1e-6), n
                assert int((got != rf[n]).any(dim=1).sum()) <= 2 * ns, n   # (only split samples may differ in a last bit)
            else:
                assert torch.equal(got, rf[n]), n
            assert torch.equal(v1[n], r1[n]) and torch.equal(v2[n], r2[n]), n
            assert m.params[n].grad is None
        assert opt.seg_steps == steps_before
        assert float(m.xyz_gradient_accum.abs().sum()) == 0 and m.denom.shape == (m.P, 1) and m.max_radii2D.shape == (m.P,)


def test_reset_opacity_and_skipped_group_match_torch_adam(oracle):
    """reset_opacity replaces the opacity tensor (no gradient that iteration): torch.optim.Adam skips the group and
    its step counter; the other groups step normally.  Compared with torch.optim.Adam run the same way."""
    P = 200
    m = fresh_model(oracle, P=P, seed=5)
    # torch twin: one Adam over six groups on copies, driven with identical gradients
    names = [("xyz", 3), ("f_dc", 3), ("f_rest", 45), ("opacity", 1), ("scaling", 3), ("rotation", 4)]
    lrs = {"xyz": 0.00016, "f_dc": 0.0025, "f_rest": 0.0025 / 20, "opacity": 0.025, "scaling": 0.005, "rotation": 0.001}

    def split(buf_views):
        f = buf_views["features"].reshape(P, 16, 3)
        return {"xyz": buf_views["xyz"], "f_dc": f[:, :1].reshape(P, 3), "f_rest": f[:, 1:].reshape(P, 45),
                "opacity": buf_views["opacity"], "scaling": buf_views["scaling"], "rotation": buf_views["rotation"]}

    m2 = GaussianModelLite(synthetic.trained_like(P, seed=5, scale_mult=1.5), torch.device("cpu"), api=oracle.api)
    tp = {k: torch.nn.Parameter(v.clone()) for k, v in split({n: m2.params[n].detach().reshape(P, w) for n, w in FIELDS}).items()}
    topt = torch.optim.Adam([{"params": [tp[k]], "lr": lrs[k], "name": k} for k, _ in names], lr=0.0, eps=1e-15)
    g = torch.Generator().manual_seed(5)
    grads = [torch.randn(m2.flat.numel(), generator=g) * 1e-2 for _ in range(7)]

    def torch_step(grad_flat, skip_opacity):
        chunks = torch.split(grad_flat, [P * w for _, w in FIELDS])
        gv = split({n: chunks[i].reshape(P, w) for i, (n, w) in enumerate(FIELDS)})
        for k in tp:
            tp[k].grad = None if (skip_opacity and k == "opacity") else gv[k].clone()
        topt.step()

    for it in range(3):
        m2.flat_grad.copy_(grads[it])
        m2.optimizer.step()
        torch_step(grads[it], False)
    # iteration 4: reset, then step without the opacity group
    m2.flat_grad.copy_(grads[3])
    m2.reset_opacity()
    with torch.no_grad():
        op = torch.sigmoid(tp["opacity"])
        new = torch.min(op, torch.ones_like(op) * 0.01)
        tp["opacity"].copy_(torch.log(new / (1 - new)))
        st = topt.state[tp["opacity"]]
        st["exp_avg"].zero_(); st["exp_avg_sq"].zero_()
    m2.optimizer.step(skip=("opacity",))
    torch_step(grads[3], True)
    assert m2.optimizer.seg_steps["opacity"] == 3 and m2.optimizer.seg_steps["xyz"] == 4
    for it in range(4, 7):
        m2.flat_grad.copy_(grads[it])
        m2.optimizer.step()
        torch_step(grads[it], False)
    got = split({n: m2.params[n].detach().reshape(P, w) for n, w in FIELDS})
    for k in tp:
        d = float((got[k] - tp[k].detach()).abs().max())
        assert d < 2e-6, (k, d)
    assert float(torch.sigmoid(m2.params["opacity"]).max()) < 0.02


def test_train_iteration_schedule(oracle):
    tr = make_trainer(oracle, P=300, W=96, H=80, dwt=False)
    centers = [c.camera_center for c in tr.cameras]
    opt = TrainOptions(iterations=40, densify_from_iter=3, densification_interval=4, opacity_reset_interval=10,
                       densify_until_iter=30, sh_increase_interval=5, cameras_extent=cameras_extent(centers),
                       densify_grad_threshold=1e-7, seed=3)
    tr.model.active_sh_degree = 0
    seen, log = [], []
    for it in range(1, 26):
        flat_before = tr.model.flat.clone()
        P_before = tr.model.P
        out = tr.train_iteration(it, opt)
        seen.append(out["camera"])
        log.append(out)
        if out["densified"] is not None:
            assert it > 3 and it % 4 == 0
            # parameters were only re-laid out, never stepped: every surviving value was present before
            assert tr.model.P != P_before or out["densified"] == (0, 0, 0)
            assert tr.model.optimizer.t == log[-2]["t"] if "t" in log[-2] else True
        elif out["reset"]:
            assert it % 10 == 0
        else:
            assert tr.model.P == P_before and not torch.equal(tr.model.flat, flat_before)
        out["t"] = tr.model.optimizer.t
    assert tr.model.active_sh_degree == 3  # ramp at 5, 10, 15 (capped at 3 afterwards)
    assert any(o["densified"] is not None and o["densified"][0] + o["densified"][1] > 0 for o in log)
    assert any(o["reset"] for o in log)
    # cameras are drawn without replacement: every block of 4 draws is a permutation of the 4 cameras
    for k in range(0, 24, 4):
        assert sorted(seen[k:k + 4]) == [0, 1, 2, 3]
    # densification iterations do not advance Adam (24 iterations, minus the densify iterations 4, 8, ..., 24)
    n_dens = sum(1 for o in log if o["densified"] is not None)
    assert tr.model.optimizer.t == 25 - n_dens
    assert tr.model.flat.numel() == tr.model.P * 59 and tr.model.xyz_gradient_accum.shape == (tr.model.P, 1)


def _row_set(m):
    """The model as a sorted list of rows (parameters | both moments): what must not depend on the row order."""
    opt = m.optimizer
    cols = [m.params[n].detach().reshape(m.P, w) for n, w in m.fields]
    cols += [v for v in opt.field_views(opt.exp_avg).values()] + [v for v in opt.field_views(opt.exp_avg_sq).values()]
    rows = torch.cat(cols, dim=1)
    key = torch.argsort(rows[:, 0], stable=True)
    for c in (1, 2):   # sort by xyz (distinct points)
        key = key[torch.argsort(rows[key, c], stable=True)]
    return rows[key[torch.argsort(rows[key, 0], stable=True)]]


def test_spatial_order_is_a_permutation_of_the_same_model(oracle):
    """GaussianModelLite(spatial_order=True): the same Gaussians, rows in Morton order of the centres - at construction and
    after a densification (parameters and moments travel with their Gaussian)."""
    sc = synthetic.trained_like(700, seed=5, scale_mult=1.5)
    perm = synthetic.morton_order(sc["means3D"])
    assert sorted(perm.tolist()) == list(range(700))
    so = synthetic.spatially_ordered(sc)
    assert torch.equal(so["means3D"], sc["means3D"][perm]) and torch.equal(so["shs"], sc["shs"][perm])
    # neighbours in memory are neighbours in space: the mean step between consecutive rows shrinks by a lot
    step = lambda x: float((x[1:] - x[:-1]).norm(dim=1).mean())  # noqa: E731
    assert step(so["means3D"]) < 0.4 * step(sc["means3D"])
    models = []
    for flag in (False, True):
        m = GaussianModelLite(sc, torch.device("cpu"), api=oracle.api, spatial_order=flag)
        g = torch.Generator().manual_seed(1)
        # the same per-GAUSSIAN gradient whatever the row: derive it from the parameters themselves
        for _ in range(3):
            m.flat_grad.copy_(torch.sin(m.flat.detach() * 37.0) * 1e-2)
            m.optimizer.step()
        m.xyz_gradient_accum = (m.params["xyz"].detach()[:, :1].abs() * 4e-4).clone()
        m.denom = torch.ones((m.P, 1))
        m.max_radii2D = torch.zeros((m.P,))
        models.append((m, g))
    assert torch.equal(_row_set(models[0][0]), _row_set(models[1][0]))
    for m, g in models:
        # no split samples (their noise is drawn per row): max_grad such that only clones and prunes happen
        m.percent_dense = 1e9
        m.densify_and_prune(2e-4, 0.005, 4.0, None, generator=g)
    a, b = models[0][0], models[1][0]
    assert a.P == b.P and a.P != 700
    assert torch.equal(_row_set(a), _row_set(b))
    assert torch.equal(b.params["xyz"].detach(), b.params["xyz"].detach()[synthetic.morton_order(b.params["xyz"])])


def test_dormant_flags_are_derived_from_the_moments(oracle):
    """FlatAdam.dormant_flags: 1 exactly for the 256-row blocks all of whose moments are +0; recomputed after anything but the
    fused step wrote the moments."""
    sc = synthetic.trained_like(1000, seed=2)
    m = GaussianModelLite(sc, torch.device("cpu"), api=oracle.api)
    opt = m.optimizer
    assert opt.dormant_flags().tolist() == [1, 1, 1, 1]
    m.flat_grad.zero_()
    m.grad_views()["opacity"][300] = 1e-3     # one Gaussian of block 1 gets a gradient
    opt.step()
    assert opt.dormant_flags().tolist() == [1, 0, 1, 1]
    opt.field_views(opt.exp_avg_sq)["rotation"][999, 3] = -0.0   # a bit pattern that is not +0
    opt.invalidate_dormant()
    assert opt.dormant_flags().tolist() == [1, 0, 1, 0]
    state = m.capture()
    m2 = GaussianModelLite(sc, torch.device("cpu"), api=oracle.api)
    assert m2.optimizer.dormant_flags().tolist() == [1, 1, 1, 1]
    m2.restore(state)
    assert m2.optimizer.dormant_flags().tolist() == [1, 0, 1, 0]
