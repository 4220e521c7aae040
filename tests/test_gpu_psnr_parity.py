"""PSNR parity protocol of SURVEY.md 8d at BASELINE config-1 size (10 k Gaussians, 400x400, 3 training views):
the same short training run - same seeds, same camera order, same Adam - once on the HIP backend and once on
the CPU oracle; held-out-view PSNR (20 log10(1/sqrt(mse)), LGDWT-GS/utils/image_utils.py:17-19) must agree
to < 0.05 dB and the parameters must stay close."""
import math

import pytest
import torch

import diff_gaussian_rasterization as dgr
from gsplat_amd import synthetic
from gsplat_amd.losses import LGDWTCriterion, LossOps
from gsplat_amd.trainer import GaussianModelLite, Trainer, camera_to, render

pytestmark = pytest.mark.gpu


def psnr(a, b):
    mse = ((a - b) ** 2).reshape(-1).mean()
    return 20 * math.log10(1.0 / math.sqrt(float(mse)))


def run(device, Rasterizer, Settings, ops, iters, P=10000, W=400, H=400, api=None):
    target = synthetic.trained_like(P, seed=1, scale_mult=1.0)
    g = torch.Generator().manual_seed(2)
    start = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in target.items()}
    start["means3D"] = start["means3D"] + 0.01 * torch.randn(start["means3D"].shape, generator=g)
    start["shs"] = start["shs"] + 0.2 * torch.randn(start["shs"].shape, generator=g)
    start["opacities"] = (start["opacities"] * 0.7).clamp(0.02, 0.98)
    cams_all = [camera_to(c, device) for c in synthetic.orbit_cameras(W, H)]
    train_idx, test_idx = [0, 8, 16], [4, 13, 21]          # "3-view" sparse setting + 3 held-out views
    bg = torch.zeros(3, device=device)
    tm = GaussianModelLite(target, device, api=api)
    with torch.no_grad():
        gt = {i: render(cams_all[i], tm, Rasterizer, Settings, bg)["render"].clone() for i in train_idx + test_idx}
    model = GaussianModelLite(start, device, api=api)
    crit = LGDWTCriterion(ops, dwt_enable=True, patch_dwt_enable=True)
    tr = Trainer(model, [cams_all[i] for i in train_idx], [gt[i] for i in train_idx], crit, Rasterizer, Settings, bg)

    def test_psnr():
        with torch.no_grad():
            return [psnr(render(cams_all[i], model, Rasterizer, Settings, bg)["render"], gt[i]) for i in test_idx]
    p0 = test_psnr()
    losses = [float(tr.step(k)) for k in range(iters)]
    return dict(p0=p0, p1=test_psnr(), losses=losses, flat=model.flat.detach().cpu().clone())


def test_training_psnr_parity_hip_vs_oracle(hip, oracle):
    iters = 60
    h = run(torch.device("cuda"), dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, LossOps(hip.api), iters, api=hip.api)
    o = run(torch.device("cpu"), oracle.Rasterizer, oracle.Settings, LossOps(oracle.api), iters, api=oracle.api)
    print("PSNR before", h["p0"], o["p0"])
    print("PSNR after ", h["p1"], o["p1"])
    print("loss first/last", h["losses"][0], h["losses"][-1], o["losses"][0], o["losses"][-1])
    for a, b in zip(h["p0"] + h["p1"], o["p0"] + o["p1"]):
        assert abs(a - b) < 0.05, (a, b)
    assert sum(h["losses"][-6:]) < sum(h["losses"][:6])       # the run did optimise (3-view training loss fell)
    assert abs(h["losses"][-1] - o["losses"][-1]) < 2e-4 * max(1.0, abs(o["losses"][-1]))
    # Adam with eps 1e-15 turns a sign flip of a ~0 gradient into a full +-lr step, so single parameters may drift
    # by a few lr; the population stays together
    d = (h["flat"] - o["flat"])
    assert float(d.abs().max()) < 0.1 and float(d.pow(2).mean().sqrt()) < 1e-3, (float(d.abs().max()), float(d.pow(2).mean().sqrt()))


def test_psnr_parity_through_densification_with_the_decisions_replayed(hip, oracle):
    """The 300-iteration densifying protocol (tests/psnr_protocol.py: densification every 40 iterations from 60, opacity reset at
    150; LGDWT-GS/train.py:262-274, scene/gaussian_model.py:409-467).  Run freely, HIP and the oracle part at the FIRST
    densification where one Gaussian's `xyz_gradient_accum / denom` lies on different sides of `densify_grad_threshold` in the
    two runs (round 5: iteration 160, ONE Gaussian of 11 843, its statistic 3.4e-4 relative from the threshold) and end 0.19 dB
    (held-out) / 0.44 dB (train) apart: they train different models from there on.  With HIP's clone / split / prune masks
    replayed in the oracle run - same discrete trajectory, every float still the oracle's own - the two agree to 0.001 /
    0.003 dB.  Asserted: the protocol's 0.05 dB on both PSNRs with the decisions replayed; and that what the oracle's own
    statistics would have decided differs from HIP's decisions for a handful of Gaussians only, each within a few per cent of
    a threshold (reported: how many, how close)."""
    import psnr_protocol as pp
    h = pp.run(torch.device("cuda"), dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, hip.api, 300, every=100, tag="hip", log=print)
    o = pp.run(torch.device("cpu"), oracle.Rasterizer, oracle.Settings, oracle.api, 300, replay=h["decisions"], every=100,
               tag="oracle<-hip", log=print)
    gap_test, gap_train = pp.psnr_gap(h, o)
    print("decisions replayed: max |dPSNR| held-out %.4f dB, train %.4f dB" % (gap_test, gap_train))
    assert [r["gaussians"] for r in h["rows"]] == [r["gaussians"] for r in o["rows"]]
    assert len(set(r["gaussians"] for r in h["rows"])) > 2            # the model did grow and get pruned
    assert gap_test < 0.05 and gap_train < 0.05, (gap_test, gap_train)
    report, first = {}, None
    for it, d in sorted(o["decisions"].items()):
        st = pp.straddlers(d["own"], d["replay"])
        report[it] = dict(gaussians=int(d["own"]["clone"].numel()), straddle=st)
        print("densification at %d: %d Gaussians, oracle's own decisions differ for %d (worst %.1e relative from its threshold, "
              "%d within 1e-6)" % (it, report[it]["gaussians"], st["clone_or_split"], st["clone_or_split_worst_rel"],
                                  st["clone_or_split_within_rel"]))
        if st["clone_or_split"] and first is None:
            first = it
        # a decision can only differ for a Gaussian whose statistic is (nearly) ON a threshold in both runs
        assert st["clone_or_split"] <= max(5, report[it]["gaussians"] // 1000), (it, st)
        assert st["clone_or_split_worst_rel"] <= 5e-2, (it, st)   # (measured: 3.4e-4 at the first, 6.8e-3 at the last densification)
    print("first densification at which the free runs would part:", first)
    try:
        import json, os
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        json.dump(dict(gap_test=gap_test, gap_train=gap_train, first_diverging_densification=first, per_densification=report,
                       hip=h["rows"], oracle_replayed=o["rows"]), open(os.path.join(out, "psnr_replay_test.json"), "w"), indent=1)
    except OSError:
        pass


def test_psnr_parity_through_densification_the_converse(hip, oracle):
    """The other direction of the replay: the ORACLE runs freely and records its clone / split / prune masks, the HIP run follows
    them - every float HIP's own.  Same bar: 0.05 dB on both PSNRs (measured 0.0006 held-out / 0.0053 train,
    profiles/r05_psnr_replay.json)."""
    import psnr_protocol as pp
    o = pp.run(torch.device("cpu"), oracle.Rasterizer, oracle.Settings, oracle.api, 300, every=100, tag="oracle", log=print)
    h = pp.run(torch.device("cuda"), dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, hip.api, 300, replay=o["decisions"],
               every=100, tag="hip<-oracle", log=print)
    gap_test, gap_train = pp.psnr_gap(o, h)
    print("oracle's decisions replayed on HIP: max |dPSNR| held-out %.4f dB, train %.4f dB" % (gap_test, gap_train))
    assert [r["gaussians"] for r in h["rows"]] == [r["gaussians"] for r in o["rows"]]
    assert gap_test < 0.05 and gap_train < 0.05, (gap_test, gap_train)


def test_schedule_with_densification_runs_on_the_gpu(hip):
    """train_iteration on the HIP backend through densify / prune / opacity reset: the flat buffers are re-laid out,
    the next iterations render and step the grown model, statistics restart at zero."""
    import diff_gaussian_rasterization as dgr
    import lgdwt_loss
    from gsplat_amd import synthetic
    from gsplat_amd.trainer import GaussianModelLite, TrainOptions, Trainer, camera_to, cameras_extent
    dev = torch.device("cuda")
    sc = synthetic.trained_like(20000, seed=2, scale_mult=1.0)
    cams = [camera_to(c, dev) for c in synthetic.orbit_cameras(320, 240)[:6]]
    g = torch.Generator().manual_seed(0)
    gts = [torch.rand((3, 240, 320), generator=g).to(dev) for _ in cams]
    model = GaussianModelLite(sc, dev, api=hip.api)
    crit = lgdwt_loss.criterion(dwt_enable=True, patch_dwt_enable=False)
    tr = Trainer(model, cams, gts, crit, dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, torch.zeros(3, device=dev))
    opt = TrainOptions(iterations=40, densify_from_iter=4, densification_interval=5, opacity_reset_interval=12,
                       densify_until_iter=30, cameras_extent=cameras_extent([c.camera_center for c in cams]), seed=4)
    sizes, losses = [], []
    for it in range(1, 28):
        out = tr.train_iteration(it, opt)
        sizes.append(out["P"])
        losses.append(float(out["loss"]))
    assert len(set(sizes)) > 1 and all(torch.isfinite(torch.tensor(losses)))
    assert model.flat.numel() == model.P * 59 and model.optimizer.exp_avg.numel() == model.flat.numel()
    # (exchange buffer = padded gradients + this step's [2, P] statistic increments + 4 floats for the validity flag)
    assert model.exchange.numel() == model.flat_padded.numel() + 2 * model.P + 4 and model.denom.shape == (model.P, 1)
    assert 0 <= model.flat_padded.numel() - model.P * 59 < 3360  # padded to whole optimizer shards
