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
