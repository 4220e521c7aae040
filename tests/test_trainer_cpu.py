"""BASELINE config-1-style plumbing on the CPU: the build-owned step loop (render adaptor, LGDWT
criterion, backward, Adam) running on the CPU oracle behind the reference's rasterizer API."""
import torch

from gsplat_amd import synthetic
from gsplat_amd.losses import LGDWTCriterion, LossOps
from gsplat_amd.trainer import FLOATS_PER_GAUSSIAN, GaussianModelLite, Trainer, camera_to, render


def make_trainer(oracle, P=600, W=160, H=128, world_size=1, rank=0, seed=0, dwt=True):
    dev = torch.device("cpu")
    sc = synthetic.trained_like(P, seed=seed, scale_mult=1.5)
    cams = [camera_to(c, dev) for c in synthetic.orbit_cameras(W, H)[:4]]
    g = torch.Generator().manual_seed(5)
    gts = [torch.rand((3, H, W), generator=g) for _ in cams]
    model = GaussianModelLite(sc, dev, api=oracle.api)
    crit = LGDWTCriterion(LossOps(oracle.api), dwt_enable=dwt, patch_dwt_enable=dwt)
    return Trainer(model, cams, gts, crit, oracle.Rasterizer, oracle.Settings, torch.zeros(3), rank, world_size)


def test_step_updates_all_six_parameter_tensors_and_lowers_the_loss(oracle):
    tr = make_trainer(oracle)
    before = tr.model.flat.clone()
    l0 = float(tr.step(0))
    assert tr.model.flat_grad.abs().sum() > 0
    for name, p in tr.model.params.items():
        assert p.grad.data_ptr() >= tr.model.flat_grad.data_ptr()  # still a view of the flat buffer
        assert float((p.detach() - before[: 0].new_zeros(())).abs().sum()) >= 0
    assert not torch.equal(before, tr.model.flat)
    assert tr.model.flat.numel() == 600 * FLOATS_PER_GAUSSIAN
    losses = [l0] + [float(tr.step(k)) for k in range(4, 24, 4)]  # same camera (index 0) every 4th step
    assert losses[-1] < losses[0]
    assert float(tr.model.denom.max()) >= 1 and float(tr.model.max_radii2D.max()) > 0


def test_render_contract_matches_reference_renderer(oracle):
    tr = make_trainer(oracle, dwt=False)
    pkg = render(tr.cameras[0], tr.model, oracle.Rasterizer, oracle.Settings, torch.zeros(3))
    assert set(pkg) == {"render", "viewspace_points", "visibility_filter", "radii", "depth"}
    assert pkg["render"].shape == (3, 128, 160) and pkg["depth"].shape == (1, 128, 160)
    assert pkg["visibility_filter"].dim() == 2 and pkg["visibility_filter"].shape[1] == 1  # nonzero() indices
    assert float(pkg["render"].min()) >= 0 and float(pkg["render"].max()) <= 1
    pkg["render"].sum().backward()
    assert pkg["viewspace_points"].grad is not None and pkg["viewspace_points"].grad.shape == (600, 3)


def test_exposure_is_applied_and_optimised(oracle):
    """render(..., use_trained_exp=True) = LGDWT-GS/gaussian_renderer/__init__.py:112-115: image -> image . E[:3,:3] + E[:3,3]
    per camera, before the clamp; the [n_cams,3,4] parameter starts at the identity (so the first render is unchanged),
    has its own Adam at the scheduled rate and only the rendered camera's row moves (train.py:280-281)."""
    from gsplat_amd.trainer import render
    tr = make_trainer(oracle, P=300, W=96, H=64)
    m = tr.model
    base = render(tr.cameras[1], m, tr.Rasterizer, tr.Settings, tr.bg)["render"].detach()
    m.enable_exposure(len(tr.cameras))
    same = render(tr.cameras[1], m, tr.Rasterizer, tr.Settings, tr.bg, use_trained_exp=True, camera_index=1)["render"]
    assert torch.equal(same.detach(), base)
    with torch.no_grad():
        m.exposure[1, :3, :3] = torch.tensor([[0.5, 0.1, 0.0], [0.0, 0.8, 0.0], [0.2, 0.0, 1.1]])
        m.exposure[1, :3, 3] = torch.tensor([0.01, 0.02, 0.03])
    raw = render(tr.cameras[1], m, tr.Rasterizer, tr.Settings, tr.bg, clamp=False)["render"].detach()
    got = render(tr.cameras[1], m, tr.Rasterizer, tr.Settings, tr.bg, use_trained_exp=True, camera_index=1,
                 clamp=False)["render"].detach()
    want = torch.einsum("chw,cd->dhw", raw, m.exposure[1, :3, :3].detach()) + m.exposure[1, :3, 3].detach()[:, None, None]
    assert torch.allclose(got, want, atol=1e-6)
    m.update_learning_rate(1)
    assert abs(m.exposure_optimizer.param_groups[0]["lr"] - 0.01) < 1e-4
    before = m.exposure.detach().clone()
    tr.step(1)  # camera 1
    moved = (m.exposure.detach() - before).abs().amax(dim=(1, 2))
    assert float(moved[1]) > 0 and float(moved[0]) == 0 and float(moved[2]) == 0


def test_exposure_optimizer_follows_the_reference_schedule(oracle):
    """train.py:279-281: the exposure optimizer steps on EVERY iteration below opt.iterations - densification iterations
    included (the exposure tensor is not replaced there, unlike the Gaussian parameters whose step then changes nothing)
    - and its rate decays over training_args.iterations, not position_lr_max_steps (gaussian_model.py:208-211)."""
    from gsplat_amd.trainer import TrainOptions, expon_lr
    tr = make_trainer(oracle, P=300, W=96, H=64)
    m = tr.model
    m.enable_exposure(len(tr.cameras))
    opt = TrainOptions(iterations=40, position_lr_max_steps=1000, densify_from_iter=2, densification_interval=3,
                       densify_until_iter=30, opacity_reset_interval=1000, cameras_extent=4.0)
    moved_on_densify = []
    for it in range(1, 8):
        before = m.exposure.detach().clone()
        flat_before = m.flat.detach().clone()
        out = tr.train_iteration(it, opt)
        lr = m.exposure_optimizer.param_groups[0]["lr"]
        assert abs(lr - expon_lr(it, 0.01, 0.001, lr_delay_steps=0, lr_delay_mult=0.0, max_steps=opt.iterations)) < 1e-12
        ci = out["camera"]
        moved = float((m.exposure.detach() - before)[ci].abs().max())
        assert moved > 0, it                                   # the rendered camera's row moves every iteration
        if out["densified"] is not None:
            moved_on_densify.append(it)
    assert moved_on_densify, "the schedule above densifies at iterations 3 and 6"
    # at the last iteration nothing steps any more (iteration < opt.iterations is false)
    before = m.exposure.detach().clone()
    tr.train_iteration(opt.iterations, opt)
    assert torch.equal(m.exposure.detach(), before)
