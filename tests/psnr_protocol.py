"""The PSNR-parity protocol of SURVEY.md 8d through DENSIFICATION, with the discrete decisions of one run replayed in another.

A 3DGS training is a smooth optimisation interrupted by discrete decisions (LGDWT-GS/scene/gaussian_model.py:409-467, called from
train.py:262-274): clone where `xyz_gradient_accum / denom >= densify_grad_threshold` and the Gaussian is small, split where it is
large, prune where `sigmoid(opacity) < min_opacity`.  Two implementations whose statistics agree to the last few bits take
DIFFERENT decisions for a Gaussian that sits within rounding of a threshold, and from then on train different models.  To separate
that (threshold chaos: nothing an implementation can match, the reference's own float atomics make it differ from itself) from a
real difference in arithmetic, `run()` can RECORD the masks of every densification and REPLAY them in another run: with the same
discrete trajectory the two implementations must again agree to the protocol's 0.05 dB.

Used by tests/test_gpu_psnr_parity.py (asserted) and tests/tools/psnr_replay.py (the long report)."""
import math

import torch

from gsplat_amd import synthetic
from gsplat_amd.losses import LGDWTCriterion, LossOps
from gsplat_amd.trainer import GaussianModelLite, TrainOptions, Trainer, camera_to, cameras_extent, render

TRAIN_IDX, TEST_IDX = [0, 8, 16], [4, 13, 21]     # "3-view" sparse setting + 3 held-out views


def psnr(a, b):
    """LGDWT-GS/utils/image_utils.py:17-19"""
    return 20 * math.log10(1.0 / math.sqrt(float(((a - b) ** 2).mean())))


def options(iters, cams, device):
    """the reference's schedule on a compressed timeline: densify every 40 iterations from 60 on, opacity reset every 150"""
    return TrainOptions(iterations=iters + 1, densify_from_iter=60, densification_interval=40, opacity_reset_interval=150,
                        densify_until_iter=int(iters * 0.7),
                        cameras_extent=cameras_extent([cams[i].camera_center for i in TRAIN_IDX]), seed=0,
                        position_lr_max_steps=iters)


def run(device, Rasterizer, Settings, api, iters=300, P=10000, W=400, H=400, replay=None, every=50, tag="", log=None):
    """-> dict(rows=[{iteration, psnr_test, psnr_train, loss, gaussians}], decisions={iteration: record}, flat)
    replay: the `decisions` of another run - its masks are used at every densification (GaussianModelLite.densify_and_prune)."""
    target = synthetic.trained_like(P, seed=1, scale_mult=1.0)
    g = torch.Generator().manual_seed(2)
    start = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in target.items()}
    start["means3D"] = start["means3D"] + 0.02 * torch.randn(start["means3D"].shape, generator=g)
    start["shs"] = start["shs"] + 0.3 * torch.randn(start["shs"].shape, generator=g)
    start["opacities"] = (start["opacities"] * 0.6).clamp(0.02, 0.98)
    cams = [camera_to(c, device) for c in synthetic.orbit_cameras(W, H)]
    bg = torch.zeros(3, device=device)
    tm = GaussianModelLite(target, device, api=api)
    with torch.no_grad():
        gt = {i: render(cams[i], tm, Rasterizer, Settings, bg)["render"].clone() for i in TRAIN_IDX + TEST_IDX}
    model = GaussianModelLite(start, device, api=api)
    crit = LGDWTCriterion(LossOps(api), dwt_enable=True, patch_dwt_enable=True)
    tr = Trainer(model, [cams[i] for i in TRAIN_IDX], [gt[i] for i in TRAIN_IDX], crit, Rasterizer, Settings, bg)
    opt = options(iters, cams, device)
    decisions = {}

    def hook(iteration):
        d = decisions[iteration] = {}
        if replay is not None:
            d["replay"] = replay[iteration]
        return d
    tr.densify_decisions = hook

    def evaluate(it, loss):
        with torch.no_grad():
            te = [psnr(render(cams[i], model, Rasterizer, Settings, bg)["render"], gt[i]) for i in TEST_IDX]
            trn = [psnr(render(cams[i], model, Rasterizer, Settings, bg)["render"], gt[i]) for i in TRAIN_IDX]
        row = dict(iteration=it, psnr_test=sum(te) / 3, psnr_train=sum(trn) / 3, loss=loss, gaussians=model.P)
        if log:
            log("%s %s" % (tag, row))
        return row
    rows = [evaluate(0, None)]
    for it in range(1, iters + 1):
        out = tr.train_iteration(it, opt)
        if it % every == 0 or it == iters:
            tr.sync()
            rows.append(evaluate(it, float(out["loss"])))
    return dict(rows=rows, decisions=decisions, flat=model.flat.detach().cpu().clone())


def psnr_gap(a, b):
    """max |dPSNR| over the checkpoints: (held-out, train)"""
    return (max(abs(x["psnr_test"] - y["psnr_test"]) for x, y in zip(a["rows"], b["rows"])),
            max(abs(x["psnr_train"] - y["psnr_train"]) for x, y in zip(a["rows"], b["rows"])))


def straddlers(own, other, rel=1e-6):
    """Two records of the SAME densification on the same rows (the discrete trajectory was shared up to here): the Gaussians
    the two sides decide differently, and how far each side's statistic is from the threshold it was compared with.
    -> dict(count, clone_or_split, prune, within_rel, worst_rel)"""
    thr, bound = own["max_grad"], own["scale_bound"]
    d_grad = (own["clone"] != other["clone"]) | (own["split"] != other["split"])
    idx = d_grad.nonzero().squeeze(1)
    # a clone/split decision has two thresholds: |g| >= max_grad and max_scale <= percent_dense * extent
    rel_g = torch.minimum((own["g"][idx].abs() - thr).abs(), (other["g"][idx].abs() - thr).abs()) / thr
    rel_s = torch.minimum((own["max_scale"][idx] - bound).abs(), (other["max_scale"][idx] - bound).abs()) / bound
    rel_cs = torch.minimum(rel_g, rel_s)
    out = dict(clone_or_split=int(idx.numel()), clone_or_split_within_rel=int((rel_cs <= rel).sum()),
               clone_or_split_worst_rel=float(rel_cs.max()) if idx.numel() else 0.0)
    if "prune" in own and "prune" in other and own["prune"].shape == other["prune"].shape:
        pi = (own["prune"] != other["prune"]).nonzero().squeeze(1)
        out["prune"] = int(pi.numel())
        # (min_opacity test; the world-size test of :459-461 compares scales - covered by rel_s-like reasoning, reported raw)
        out["prune_opacity"] = [float(x) for x in own["opacity"][pi][:8]]
    out["count"] = out["clone_or_split"] + out.get("prune", 0)
    out["rel"] = rel
    return out


def first_divergence(a, b):
    """first densification iteration at which two FREE runs took different decisions (None: never) + the straddle report there"""
    for it in sorted(a["decisions"]):
        da, db = a["decisions"][it], b["decisions"].get(it)
        da, db = da.get("own", da), (db.get("own", db) if db is not None else None)
        if db is None or da["clone"].shape != db["clone"].shape:
            return it, None   # (already a different model: an earlier prune differed)
        if not (torch.equal(da["clone"], db["clone"]) and torch.equal(da["split"], db["split"]) and
                torch.equal(da["prune"], db["prune"])):
            return it, straddlers(da, db)
    return None, None
