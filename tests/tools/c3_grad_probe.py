#!/usr/bin/env python3
"""Full-size (BASELINE configs[2]/[3]) HIP-vs-oracle probe: where does the fp32 gradient noise enter?

  python tests/tools/c3_grad_probe.py [P W H [out.json]]

For one view of the trained-like scene: forward state (bit-exact in reference-list mode), image, every gradient
tensor and the per-Gaussian sums of the blend backward (`rows`) of the HIP library against the double-accumulating
CPU oracle, for two image cotangents (white noise = worst-case cancellation; the LGDWT loss gradient = what training
feeds) and both list modes; plus HIP run-to-run differences (atomic order) and a description of the worst Gaussian.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "sparse-view-3dgs-pack_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402

import fullsize_parity as fp  # noqa: E402
import oracle_lib  # noqa: E402
from gsplat_amd import hip_backend, synthetic  # noqa: E402


def main():
    P, W, H = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (1_000_000, 1920, 1080)
    out_path = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "gpurun_out", "c3_grad_probe.json")
    cam_i = int(os.environ.get("PROBE_CAM", "3"))
    dev = torch.device("cuda")
    from simple_knn._C import distCUDA2
    import lgdwt_loss
    hip = hip_backend()
    orc = oracle_lib.get()
    knn = lambda x: distCUDA2(x.to(dev)).cpu()  # noqa: E731
    sc = synthetic.trained_like(P, seed=0, sh_degree=3, knn=knn)
    gt_sc = synthetic.trained_like(P, seed=1, sh_degree=3, knn=knn)
    cam = synthetic.orbit_cameras(W, H)[cam_i]
    bg = torch.zeros(3)
    cpu = torch.device("cpu")
    report = dict(P=P, W=W, H=H, camera=cam_i)

    t0 = time.time()
    ofw = fp.forward(orc.backend, sc, cam, cpu, bg)
    report["oracle_forward_s"] = time.time() - t0
    print("oracle forward %.1f s, R=%d" % (report["oracle_forward_s"], ofw["R"]), flush=True)

    # cotangents
    g = torch.Generator().manual_seed(3)
    cots = {"noise": torch.randn((3, H, W), generator=g)}
    hip.tile_cull = True
    gt = fp.forward(hip, gt_sc, cam, dev, bg)["color"]
    gt = (torch.round(gt.clamp(0, 1) * 255.0) / 255.0).contiguous()
    crit = lgdwt_loss.criterion(dwt_enable=True, patch_dwt_enable=True)
    mask = crit.elf_mask(gt)
    img = fp.forward(hip, sc, cam, dev, bg)["color"].clone().requires_grad_(True)
    loss, _ = crit(img.clamp(0, 1), gt, mask=mask)
    loss.backward()
    cots["loss"] = img.grad.detach().cpu()
    report["loss"] = float(loss.detach())

    for cull in (False, True):
        hip.tile_cull = cull
        hfw = fp.forward(hip, sc, cam, dev, bg)
        tag = "cull%d" % int(cull)
        rep = report[tag] = dict(R=hfw["R"])
        assert torch.equal(hfw["radii"].cpu(), ofw["radii"])
        dc = (hfw["color"].cpu() - ofw["color"]).abs()
        rep["color_max_abs_err"] = float(dc.max())
        rep["color_pixels_over_1e-4"] = int((dc.amax(dim=0) > 1e-4).sum())
        # flipped pixels: zero the cotangent there for both sides (see test_gpu_raster_parity.flip_mask)
        flip = dc.amax(dim=0) > 2e-5
        rep["flipped_pixels"] = int(flip.sum())
        for cname, cot in cots.items():
            cot = cot.clone()
            cot[:, flip] = 0
            t0 = time.time()
            og, orows = fp.backward(orc.backend, ofw, cot)
            t_or = time.time() - t0
            hg, hrows = fp.backward(hip, hfw, cot)
            hg2, hrows2 = fp.backward(hip, hfw, cot)
            r = rep[cname] = dict(oracle_backward_s=t_or)
            r["grads"] = fp.compare_grads(hg, og)
            r["rows"] = fp.compare_rows(hrows, orows)
            r["hip_run_to_run"] = fp.compare_grads(hg2, hg)
            r["hip_rows_run_to_run"] = fp.compare_rows(hrows2, hrows)
            print("== %s / %s (oracle bwd %.1f s)" % (tag, cname, t_or))
            for k, v in r["grads"].items():
                print("  grad %-14s max_rel %.3e rms_rel %.3e (ref max %.3e) | run-to-run max_rel %.3e" % (
                    k, v["max_rel"], v["rms_rel"], v["ref_max"], r["hip_run_to_run"][k]["max_rel"]))
            for k, v in r["rows"].items():
                print("  rows %-14s max_rel %.3e rms_rel %.3e (ref max %.3e) | run-to-run max_rel %.3e" % (
                    k, v["max_rel"], v["rms_rel"], v["ref_max"], r["hip_rows_run_to_run"][k]["max_rel"]))
            # the worst rotation-gradient Gaussian
            for name in ("rotations", "scales"):
                i = r["grads"][name]["argmax"] // hg[name].shape[1]
                st = hip.export_state(P, W, H, hfw["R"], hfw["geom"], hfw["binning"], hfw["img"])
                co = st["conic_opacity"][i].cpu().double()
                det = float(co[0] * co[2] - co[1] * co[1])
                a, c_, b = float(co[2] / det), float(co[0] / det), float(-co[1] / det)  # cov2D (+0.3)
                w = dict(index=int(i), radius=int(hfw["radii"][i]), tiles=int(st["tiles_touched"][i]),
                         conic=[float(x) for x in co], cov2D=[a, b, c_], depth=float(st["depths"][i]),
                         scales=[float(x) for x in sc["scales"][i]],
                         hip=[float(x) for x in hg[name][i]], oracle=[float(x) for x in og[name][i]],
                         rows_hip=[float(x) for x in hrows[i, :10]], rows_oracle=[float(x) for x in orows[i, :10]])
                r["worst_" + name] = w
                print("  worst %s: %s" % (name, json.dumps(w)))
            # per-Gaussian conditioning: how much a relative change of the conic sums is amplified into dL_dscales
            dcon = (hrows[:, 2:5].double() - orows[:, 2:5].double()).abs().amax(dim=1)
            ncon = orows[:, 2:5].double().abs().amax(dim=1).clamp_min(1e-30)
            rel_in = dcon / ncon
            dsc = (hg["scales"].double() - og["scales"].double()).abs().amax(dim=1)
            vis = ofw["radii"] > 0
            r["conic_rows_rel_err_per_gaussian"] = dict(
                median=float(rel_in[vis].median()), p99=float(rel_in[vis].quantile(0.99)), max=float(rel_in[vis].max()))
            r["scales_abs_err_top"] = [float(x) for x in dsc.topk(5).values]
            print("  per-Gaussian rel err of conic rows: median %.2e p99 %.2e max %.2e" % (
                r["conic_rows_rel_err_per_gaussian"]["median"], r["conic_rows_rel_err_per_gaussian"]["p99"],
                r["conic_rows_rel_err_per_gaussian"]["max"]), flush=True)
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(report, open(out_path, "w"), indent=1)
    print("wrote", out_path)


if __name__ == "__main__":
    main()
