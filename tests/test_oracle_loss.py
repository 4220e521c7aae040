"""CPU: the loss oracle (through the same LossOps autograd glue the product uses) against
(1) golden values from the reference's importable python (L1 / SSIM value + gradient),
(2) tests/golden/lgdwt_loss.npz - what the reference's own LGDWT-GS/utils/loss_utils.py returns for get_dwt_subbands /
    compute_elf_map / compute_patch_dwt_loss (+ autograd) and the defaults of its arguments module (tests/lgdwt_fixture.py),
(3) the independent Haar step of torch_loss_reference.py at odd / tiny sizes, (4) analytic KATs."""
import os

import numpy as np
import pytest
import torch

import lgdwt_fixture
import torch_loss_reference as ref
from gsplat_amd.losses import LGDWTCriterion, LossOps

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ops(oracle):
    return LossOps(oracle.api)


def test_l1_and_ssim_vs_reference_python_golden(ops):
    z = np.load(os.path.join(G, "image_losses.npz"))
    for tag in ("a", "b", "c"):
        img1 = torch.tensor(z["img1_" + tag]).requires_grad_(True)
        img2 = torch.tensor(z["img2_" + tag])
        l = ops.l1_loss(img1, img2)
        l.backward()
        assert abs(float(l) - float(z["l1_" + tag])) < 1e-6
        assert np.abs(img1.grad.numpy() - z["dl1_" + tag]).max() < 1e-9
        x = torch.tensor(z["img1_" + tag]).requires_grad_(True)
        s = ops.ssim(x, img2)
        s.backward()
        assert abs(float(s) - float(z["ssim_" + tag])) < 2e-6, tag
        gref = z["dssim_" + tag]
        assert np.abs(x.grad.numpy() - gref).max() < 2e-5 * np.abs(gref).max(), tag
        s2 = ops.fused_ssim(torch.tensor(z["img1_" + tag])[None], img2[None])
        assert abs(float(s2) - float(z["ssim_" + tag])) < 2e-6


def test_haar_known_answers(ops):
    x = torch.full((1, 2, 6, 8), 3.0)
    ll, lh, hl, hh = ops.dwt_haar(x)
    assert torch.allclose(ll, torch.full_like(ll, 6.0), atol=1e-6) and float(lh.abs().max()) < 1e-6
    assert float(hl.abs().max()) < 1e-6 and float(hh.abs().max()) < 1e-6
    # ramp along W: only the "high along W" band (HL) responds
    r = torch.arange(8, dtype=torch.float32).reshape(1, 1, 1, 8).repeat(1, 1, 6, 1)
    ll, lh, hl, hh = ops.dwt_haar(r)
    assert float(lh.abs().max()) < 1e-6 and float(hh.abs().max()) < 1e-6
    assert torch.allclose(hl, torch.full_like(hl, -1.0), atol=1e-6)   # (a - b + c - d)/2 = -1
    # ramp along H: only LH
    ll, lh, hl, hh = ops.dwt_haar(r.transpose(2, 3).contiguous())
    assert float(hl.abs().max()) < 1e-6 and torch.allclose(lh, torch.full_like(lh, -1.0), atol=1e-6)
    # checkerboard: only HH
    yy, xx = torch.meshgrid(torch.arange(6), torch.arange(8), indexing="ij")
    cb = ((yy + xx) % 2).float().reshape(1, 1, 6, 8)
    ll, lh, hl, hh = ops.dwt_haar(cb)
    assert float(lh.abs().max()) < 1e-6 and float(hl.abs().max()) < 1e-6
    assert torch.allclose(hh.abs(), torch.ones_like(hh), atol=1e-6)
    # orthonormal: energy is preserved for even sizes
    g = torch.Generator().manual_seed(0)
    x = torch.randn((1, 3, 10, 12), generator=g)
    bands = ops.dwt_haar(x)
    assert abs(sum(float((b ** 2).sum()) for b in bands) - float((x ** 2).sum())) < 1e-3


@pytest.mark.parametrize("H,W", [(16, 24), (17, 23), (5, 5), (33, 64), (1, 7)])
def test_dwt_bands_and_adjoint_vs_torch_restatement(ops, H, W):
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.randn((2, 3, H, W), generator=g)
    xo = x.clone().requires_grad_(True)
    xr = x.clone().double().requires_grad_(True)
    bo = ops.get_dwt_subbands(xo)
    br = ref.haar_bands_2level(xr)
    ws = {k: torch.randn(br[k].shape, generator=g) for k in br}
    for k in br:
        assert bo[k].shape == br[k].shape, k
        assert float((bo[k].double() - br[k]).abs().max()) < 2e-6, k
    sum((bo[k] * ws[k]).sum() for k in bo).backward()
    sum((br[k] * ws[k].double()).sum() for k in br).backward()
    assert float((xo.grad.double() - xr.grad).abs().max()) < 1e-5


@pytest.mark.parametrize("tag", lgdwt_fixture.CASES)
def test_dwt_elf_patch_terms_vs_the_reference_module_fixture(ops, tag):
    """D1-D4 against the reference's own loss_utils.py run in the build container (128x128, 131x260, 256x384, 75x141)."""
    lgdwt_fixture.check_case(ops, torch.device("cpu"), lgdwt_fixture.load(), tag)


def test_criterion_defaults_and_running_mean_vs_the_reference_fixture(ops):
    z = lgdwt_fixture.load()
    lgdwt_fixture.check_criterion_defaults(LGDWTCriterion, ops, z)
    lgdwt_fixture.check_running_mean(LGDWTCriterion, ops, z, torch.device("cpu"))
    assert float(ops.compute_patch_dwt_loss(torch.rand(1, 3, 40, 200), torch.rand(1, 3, 40, 200), torch.rand(1, 1, 40, 200))) == 0.0


def test_unfused_band_path_equals_the_one_pass_loss(ops):
    """get_dwt_subbands + l1_loss per band (as train.py:132-164 literally does) == dwt_l1_loss."""
    g = torch.Generator().manual_seed(130 + 131)
    pred = torch.rand((3, 130, 131), generator=g)
    gt = (pred + 0.2 * torch.randn((3, 130, 131), generator=g)).clamp(0, 1)
    weights = (1.0, 1.0, 1.0, 0.3, 0.5, 0.25, 0.0, 2.0)
    lo, _ = ops.dwt_l1_loss(pred, gt, weights)
    pb, gb = ops.get_dwt_subbands(pred[None]), ops.get_dwt_subbands(gt[None])
    unfused = sum(w * ops.l1_loss(pb[k], gb[k]) for w, k in zip(weights, lgdwt_fixture.BANDS) if w)
    assert abs(float(unfused) - float(lo)) < 1e-6


def test_criterion_composition_matches_reference_formula(ops):
    g = torch.Generator().manual_seed(3)
    gt = torch.rand((3, 160, 256), generator=g)
    pred = (gt + 0.1 * torch.randn((3, 160, 256), generator=g)).clamp(0, 1)
    crit = LGDWTCriterion(ops)
    p = pred.clone().requires_grad_(True)
    loss, parts = crit(p, gt)
    # train.py:188-202 by hand
    base = 0.8 * ops.l1_loss(pred, gt) + 0.2 * (1.0 - ops.fused_ssim(pred[None], gt[None]))
    dwt, _ = ops.dwt_l1_loss(pred, gt, (1, 1, 1, 0, 0, 0, 0, 0))
    ratio = float(base) / (float(dwt) + 1e-8)
    m = 0.95 * 1.0 + 0.05 * ratio
    scale = max(0.1, min(10.0, m))
    elf = ops.compute_elf_map(gt[None])
    patch = ops.compute_patch_dwt_loss(pred[None], gt[None], elf, 128, 0.2, 1.0, 1.0)
    expect = float(base) + scale * float(dwt) + 0.1 * float(patch)
    assert abs(float(loss) - expect) < 1e-6
    loss.backward()
    assert p.grad is not None and float(p.grad.abs().max()) > 0
    # second call advances the running mean
    loss2, parts2 = crit(pred, gt)
    m2 = 0.95 * m + 0.05 * ratio
    assert abs(float(parts2["dwt_scale"]) - max(0.1, min(10.0, m2))) < 1e-6


def test_fused_criterion_equals_the_term_by_term_composition(ops):
    """FusedLGDWTLoss (one autograd node on the un-clamped render, device-side loss composition) against the
    modular path (clamp + l1_loss + fused_ssim + dwt_l1_loss + compute_patch_dwt_loss + torch arithmetic)."""
    g = torch.Generator().manual_seed(9)
    gt = torch.rand((3, 160, 256), generator=g)
    raw = gt + 0.25 * torch.randn((3, 160, 256), generator=g)        # leaves [0,1] in places: exercises the clamp
    ca, cb = LGDWTCriterion(ops, fused=True), LGDWTCriterion(ops, fused=False)
    for it in range(3):
        ra = raw.clone().requires_grad_(True)
        rb = raw.clone().requires_grad_(True)
        la, pa = ca.fused_call(ra, gt)
        lb, pb = cb(rb.clamp(0, 1), gt)
        (la * 1.5).backward()
        (lb * 1.5).backward()
        assert abs(float(la) - float(lb)) < 2e-6, it
        assert abs(float(pa["dwt_scale"]) - float(pb["dwt_scale"])) < 1e-6
        gref = rb.grad
        assert float((ra.grad - gref).abs().max()) < 1e-6 * max(1.0, float(gref.abs().max())) + 1e-9
        assert float(ra.grad[(raw < 0) | (raw > 1)].abs().max()) == 0.0
    # DWT / patch switched off, small image (no patches)
    ca, cb = LGDWTCriterion(ops, dwt_enable=False, patch_dwt_enable=False), LGDWTCriterion(ops, dwt_enable=False, patch_dwt_enable=False, fused=False)
    r = raw[:, :40, :56].clone().contiguous()
    la, _ = ca.fused_call(r, gt[:, :40, :56].contiguous())
    lb, _ = cb(r.clamp(0, 1), gt[:, :40, :56].contiguous())
    assert abs(float(la) - float(lb)) < 2e-6
