"""Checks a LossOps implementation (CPU oracle or HIP) against tests/golden/lgdwt_loss.npz: what the reference's OWN
LGDWT-GS/utils/loss_utils.py (get_dwt_subbands :106-153, compute_elf_map :336-366, compute_patch_dwt_loss :368-442,
l1_loss, ssim) returned in the build container, with the defaults of LGDWT-GS/arguments/__init__.py:103-122
(generator: tests/golden/make_golden.py::gen_lgdwt_loss; the reference's `pytorch_wavelets` import was served by the
independent Haar step of tests/torch_loss_reference.py, so the 2x2 butterfly itself is the one piece not pinned)."""
import os

import numpy as np
import torch

import torch_loss_reference as tlr

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lgdwt_loss.npz")
CASES = ("a", "b", "c", "d")
BANDS = ("LL1", "LH1", "HL1", "HH1", "LL2", "LH2", "HL2", "HH2")


def load():
    return np.load(PATH)


def _close(a, b, tol, what):
    a = torch.as_tensor(np.asarray(a)).double() if not torch.is_tensor(a) else a.detach().cpu().double()
    b = torch.as_tensor(np.asarray(b)).double()
    scale = max(1e-12, float(b.abs().max()))
    e = float((a - b).abs().max()) / scale
    assert e <= tol, "%s: rel err %.3e > %.1e" % (what, e, tol)
    return e


def check_case(ops, device, z, tag):
    """Every D1-D4 quantity of one fixture case.  Tolerances: fp32 rounding of sums (the reference sums with torch's
    pairwise reductions, the kernels in their own order)."""
    rs = int(z["row_stride_" + tag])
    gt = torch.tensor(z["gt_u8_" + tag].astype(np.float32) / np.float32(255.0)).to(device)
    pred = torch.tensor(z["pred_" + tag]).to(device)
    ps, pct, w_lh, w_hl = (float(v) for v in z["patch_args_" + tag])
    ps = int(ps)
    # D1: bands (layout [N,C,h,w], level-2 from LL1, odd sizes repeat the last sample) and their adjoint
    x = pred[None].clone().requires_grad_(True)
    bands = ops.get_dwt_subbands(x)
    assert tuple(bands.keys()) == BANDS
    for k in BANDS:
        ref = z["band_%s_%s" % (k, tag)]
        got = bands[k].detach()[0, :, ::rs]
        assert tuple(got.shape) == ref.shape, (k, got.shape, ref.shape)
        _close(got, ref, 2e-6, "band %s %s" % (k, tag))
    sum((bands[k] * tlr.cotangent(bands[k].shape, i).to(device)).sum() for i, k in enumerate(BANDS)).backward()
    _close(x.grad[0, :, ::rs], z["dbands_" + tag], 5e-6, "adjoint " + tag)
    # D2: per-band L1 means and the gradient of their weighted sum (train.py:132-164)
    allw = tuple(float(w) for w in z["dwt_weights_all"])
    x = pred.clone().requires_grad_(True)
    total, means = ops.dwt_l1_loss(x, gt, allw)
    ref_l1 = z["band_l1_" + tag]
    _close(means.detach().cpu()[:8], ref_l1, 5e-6, "band L1 means " + tag)
    assert abs(float(total) - float((np.array(allw) * ref_l1).sum())) < 5e-6 * max(1.0, float(total))
    total.backward()
    # The L1 gradient carries sign(band difference): where a difference sits within fp32 rounding of zero (a handful of
    # 2x2 / 4x4 blocks per image) the sign is decided by the rounding order of the Haar step - conv taps in the reference,
    # the butterfly here.  Those blocks are found from the bands themselves, counted, bounded and left out.
    H, W = pred.shape[-2:]
    near = torch.zeros((H, W), dtype=torch.bool)
    pbands, gbands = ops.get_dwt_subbands(pred[None]), ops.get_dwt_subbands(gt[None])
    for i, k in enumerate(BANDS):
        if allw[i] == 0.0:
            continue
        diff = (pbands[k] - gbands[k]).abs()
        tie = ((diff < 1e-6) & (diff > 0)).any(dim=1)[0].cpu()   # (exactly 0 - both images equal - is sign(0) = 0 for both)
        f = 2 if k.endswith("1") else 4
        up = tie.repeat_interleave(f, 0).repeat_interleave(f, 1)[:H, :W]
        near[: up.shape[0], : up.shape[1]] |= up
    assert int(near.sum()) <= max(64, H * W // 500), "too many near-zero band differences: %d" % int(near.sum())
    keep = (~near)[None, ::rs].to(x.grad.device)
    _close(x.grad[:, ::rs] * keep, torch.tensor(z["dl1_all_" + tag]) * keep.cpu(), 2e-5, "d(weighted band L1) " + tag)
    # the reference's default weights (arguments/__init__.py:105-114)
    dflt = tuple(float(w) for w in z["dwt_weights"])
    t2, _ = ops.dwt_l1_loss(pred, gt, dflt)
    assert abs(float(t2) - float((np.array(dflt) * ref_l1).sum())) < 5e-6
    # D3: ELF map (bilinear x2, align_corners=False, 1e-8)
    elf = ops.compute_elf_map(gt[None])
    _close(elf[0, :, ::rs], z["elf_" + tag], 5e-6, "elf " + tag)
    # D4: patch selection (unfold order, kthvalue index, >=) and the loss with its HH weight
    mask, _ = ops.patch_mask(elf, ps, pct)
    assert np.array_equal(mask.detach().cpu().numpy().astype(bool).reshape(-1), z["patch_mask_" + tag]), "patch mask " + tag
    x = pred[None].clone().requires_grad_(True)
    pl = ops.compute_patch_dwt_loss(x, gt[None], elf, ps, pct, w_lh, w_hl)
    assert abs(float(pl) - float(z["patch_loss_" + tag])) < 5e-6 * max(1.0, float(z["patch_loss_" + tag])), "patch loss " + tag
    pl.backward()
    _close(x.grad[0, :, ::rs], z["dpatch_" + tag], 2e-5, "d patch " + tag)
    # base loss terms from the same reference module
    lam = float(z["lambda_dssim"])
    x = pred.clone().requires_grad_(True)
    l1 = ops.l1_loss(x, gt)
    ss = ops.fused_ssim(x[None], gt[None])
    assert abs(float(l1) - float(z["l1_" + tag])) < 2e-6 and abs(float(ss) - float(z["ssim_" + tag])) < 5e-6
    ((1.0 - lam) * l1 + lam * (1.0 - ss)).backward()
    _close(x.grad[:, ::rs], z["dbase_" + tag], 3e-5, "d base " + tag)


def check_criterion_defaults(crit_cls, ops, z):
    """The criterion's defaults are the reference's (arguments/__init__.py:103-122, train.py:188-202)."""
    c = crit_cls(ops)
    assert c.lambda_dssim == float(z["lambda_dssim"])
    assert tuple(float(w) for w in c.dwt_weights) == tuple(float(w) for w in z["dwt_weights"])
    assert c.patch_dwt_weight == float(z["patch_dwt_weight"]) and c.patch_size == int(z["patch_size"])
    assert c.patch_percentile == float(z["patch_percentile"])
    assert c.patch_lh1_weight == float(z["patch_lh1_weight"]) and c.patch_hl1_weight == float(z["patch_hl1_weight"])
    assert bool(z["dwt_enable"]) and bool(z["patch_dwt_enable"])
    assert float(z["patch_loss_small"]) == 0.0   # loss_utils.py:386-387: images smaller than one patch


def check_running_mean(crit_cls, ops, z, device):
    """train.py:188-202 on the fixture's constants: m <- 0.95 m + 0.05 base / (dwt + 1e-8) from m0 = 1, scale = clamp(m, 0.1,
    10), loss = base + scale dwt + beta patch; three consecutive calls of the criterion."""
    a, b, eps, m, lo, hi = (float(v) for v in z["running_mean"])
    gt = torch.tensor(z["gt_u8_c"].astype(np.float32) / np.float32(255.0)).to(device)
    pred = torch.tensor(z["pred_c"]).to(device)
    lam, beta = float(z["lambda_dssim"]), float(z["patch_dwt_weight"])
    base = (1.0 - lam) * float(z["l1_c"]) + lam * (1.0 - float(z["ssim_c"]))
    dwt = float((z["dwt_weights"] * z["band_l1_c"]).sum())
    patch = float(z["patch_loss_c"])   # (case c was generated with the default patch arguments)
    # (the fused form clamps the image itself - the fixture's pred is deliberately un-clamped - and is held to the modular
    # form on clamped input by test_fused_criterion_equals_the_term_by_term_composition / its GPU twin)
    for fused in (False,):
        crit = crit_cls(ops, fused=fused)
        mm = m
        for it in range(3):
            mm = a * mm + b * base / (dwt + eps)
            expect = base + min(hi, max(lo, mm)) * dwt + beta * patch
            loss, parts = (crit.fused_call(pred, gt) if fused else crit(pred, gt))
            assert abs(float(loss) - expect) < 1e-5 * max(1.0, expect), (fused, it, float(loss), expect)
            assert abs(float(parts["dwt_scale"]) - min(hi, max(lo, mm))) < 1e-5
