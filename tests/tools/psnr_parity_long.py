"""Developer tool / report generator: the PSNR-parity protocol of SURVEY 8d run LONGER than the test does - the same
training run (BASELINE config-1 size: 10 k Gaussians, 400x400, 3 training views, full LGDWT loss, the reference's
schedule incl. densification and opacity reset on a compressed timeline) on the HIP backend and on the CPU oracle.
Writes gpurun_out/<tag>_psnr_parity[_densify].json (per-checkpoint held-out PSNR, training PSNR, loss, number of
Gaussians).  The HIP run is done TWICE: two runs of the very same HIP path differ too (float-atomic order of the blend
backward -> sign flips of near-zero gradients -> Adam, eps 1e-15, turns each into a +-lr step), and that HIP-vs-HIP
spread is the yardstick for the HIP-vs-oracle gap.
    python tests/tools/psnr_parity_long.py [iterations] [densify|plain] [tag]"""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sparse-view-3dgs-pack_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import diff_gaussian_rasterization as dgr  # noqa: E402
import oracle_lib  # noqa: E402
from gsplat_amd import hip_backend, synthetic  # noqa: E402
from gsplat_amd.losses import LGDWTCriterion, LossOps  # noqa: E402
from gsplat_amd.trainer import GaussianModelLite, TrainOptions, Trainer, camera_to, cameras_extent, render  # noqa: E402

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
DENSIFY = (len(sys.argv) > 2 and sys.argv[2] == "densify")


def psnr(a, b):
    return 20 * math.log10(1.0 / math.sqrt(float(((a - b) ** 2).mean())))


def run(device, Rasterizer, Settings, api, tag):
    P, W, H = 10000, 400, 400
    target = synthetic.trained_like(P, seed=1, scale_mult=1.0)
    g = torch.Generator().manual_seed(2)
    start = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in target.items()}
    start["means3D"] = start["means3D"] + 0.02 * torch.randn(start["means3D"].shape, generator=g)
    start["shs"] = start["shs"] + 0.3 * torch.randn(start["shs"].shape, generator=g)
    start["opacities"] = (start["opacities"] * 0.6).clamp(0.02, 0.98)
    cams = [camera_to(c, device) for c in synthetic.orbit_cameras(W, H)]
    train_idx, test_idx = [0, 8, 16], [4, 13, 21]
    bg = torch.zeros(3, device=device)
    tm = GaussianModelLite(target, device, api=api)
    with torch.no_grad():
        gt = {i: render(cams[i], tm, Rasterizer, Settings, bg)["render"].clone() for i in train_idx + test_idx}
    model = GaussianModelLite(start, device, api=api)
    crit = LGDWTCriterion(LossOps(api), dwt_enable=True, patch_dwt_enable=True)
    tr = Trainer(model, [cams[i] for i in train_idx], [gt[i] for i in train_idx], crit, Rasterizer, Settings, bg)
    if "limits" in tag:  # depth-limited lists with the deferred verdict (through densification and opacity resets)
        tr.depth_limit = "deferred"
    opt = TrainOptions(iterations=ITERS + 1, densify_from_iter=60 if DENSIFY else 10 ** 9, densification_interval=40,
                       opacity_reset_interval=150 if DENSIFY else 10 ** 9, densify_until_iter=int(ITERS * 0.7),
                       cameras_extent=cameras_extent([cams[i].camera_center for i in train_idx]), seed=0,
                       position_lr_max_steps=ITERS)

    def evaluate(it, loss):
        with torch.no_grad():
            te = [psnr(render(cams[i], model, Rasterizer, Settings, bg)["render"], gt[i]) for i in test_idx]
            trn = [psnr(render(cams[i], model, Rasterizer, Settings, bg)["render"], gt[i]) for i in train_idx]
        row = dict(iteration=it, psnr_test=sum(te) / 3, psnr_train=sum(trn) / 3, loss=loss, gaussians=model.P)
        print(tag, row, flush=True)
        return row
    rows = [evaluate(0, None)]
    t0 = time.perf_counter()
    for it in range(1, ITERS + 1):
        out = tr.train_iteration(it, opt)
        if it % 50 == 0 or it == ITERS:
            tr.sync()
            rows.append(evaluate(it, float(out["loss"])))
    return rows, time.perf_counter() - t0


hip, orc = hip_backend(), oracle_lib.get()
TAG = sys.argv[3] if len(sys.argv) > 3 else "r02"
h, th = run(torch.device("cuda"), dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, hip.api, "hip   ")
h2, _ = run(torch.device("cuda"), dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, hip.api, "hip #2")
dl0 = dict(hip.depth_limit_stats)
h3, _ = run(torch.device("cuda"), dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, hip.api, "hip limits")
dl = {k: hip.depth_limit_stats[k] - dl0[k] for k in dl0}
o, to = run(torch.device("cpu"), orc.Rasterizer, orc.Settings, orc.api, "oracle")
rep = dict(protocol="SURVEY 8d PSNR parity, config-1 size (10k Gaussians, 400x400, 3 train / 3 held-out views), "
                    "L1+SSIM+DWT2+patchDWT, Adam, reference schedule%s" % (" with densification every 40 it from 60, "
                                                                        "opacity reset every 150" if DENSIFY else ""),
           iterations=ITERS, hip=h, hip_second_run=h2, hip_depth_limited=h3, depth_limited_views=dl, oracle=o,
           seconds=dict(hip=th, oracle=to),
           hip_vs_hip_limited_max_abs_psnr_test_diff=max(abs(a["psnr_test"] - b["psnr_test"]) for a, b in zip(h, h3)),
           hip_vs_hip_limited_max_abs_psnr_train_diff=max(abs(a["psnr_train"] - b["psnr_train"]) for a, b in zip(h, h3)),
           hip_vs_hip_max_abs_psnr_test_diff=max(abs(a["psnr_test"] - b["psnr_test"]) for a, b in zip(h, h2)),
           hip_vs_hip_max_abs_psnr_train_diff=max(abs(a["psnr_train"] - b["psnr_train"]) for a, b in zip(h, h2)),
           max_abs_psnr_test_diff=max(abs(a["psnr_test"] - b["psnr_test"]) for a, b in zip(h, o)),
           max_abs_psnr_train_diff=max(abs(a["psnr_train"] - b["psnr_train"]) for a, b in zip(h, o)))
name = "%s_psnr_parity%s.json" % (TAG, "_densify" if DENSIFY else "")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rep, open(os.path.join(ROOT, "gpurun_out", name), "w"), indent=1)
print("HIP vs oracle: max |dPSNR| test %.4f dB, train %.4f dB; HIP vs HIP: test %.4f dB, train %.4f dB; HIP vs HIP with depth "
      "limits: test %.4f dB, train %.4f dB (%s); %s" % (
          rep["max_abs_psnr_test_diff"], rep["max_abs_psnr_train_diff"], rep["hip_vs_hip_max_abs_psnr_test_diff"],
          rep["hip_vs_hip_max_abs_psnr_train_diff"], rep["hip_vs_hip_limited_max_abs_psnr_test_diff"],
          rep["hip_vs_hip_limited_max_abs_psnr_train_diff"], dl, name))
