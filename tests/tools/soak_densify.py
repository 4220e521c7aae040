"""Developer tool: the train loop WITH its densification schedule (gsplat_amd.trainer.Trainer.train_iteration) on a model whose
rows are kept in spatial order, eager two-phase fused step with deferred depth limits - what the 20-step bench cannot show: the
re-layouts (rows re-ordered after every densification, moments travelling with their Gaussians, dormant-block flags derived
again), opacity resets, an SH ramp; checks finiteness and that every dormant flag the kernels left standing is true of the moments.
   python tests/tools/soak_densify.py [iterations] [config]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sparse-view-3dgs-pack_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from gsplat_amd import hip_backend, synthetic  # noqa: E402
from gsplat_amd.trainer import TrainOptions, cameras_extent  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
cfg = sys.argv[2] if len(sys.argv) > 2 else "c2"
dev = torch.device("cuda", 0)
tr, scene, cams, gts = bench.build_workload(cfg, dev, 0, 1)
assert tr.model.spatial_order
tr.depth_limit = "deferred"
be = hip_backend()
opt = TrainOptions(iterations=n + 1, densify_from_iter=60, densification_interval=60, opacity_reset_interval=200,
                   densify_until_iter=n - 50, sh_increase_interval=100, densify_grad_threshold=0.00005,
                   cameras_extent=cameras_extent([c.camera_center.cpu() for c in cams]))
t0 = time.time()
P0 = tr.model.P
events = []
for it in range(1, n + 1):
    out = tr.train_iteration(it, opt)
    if out["densified"] is not None or out["reset"]:
        events.append((it, out["densified"], out["reset"], out["P"]))
        if out["densified"] is not None:
            x = tr.model.params["xyz"].detach()
            perm = synthetic.morton_order(x).to(dev)
            moved = int((perm != torch.arange(x.shape[0], device=dev)).sum())
            assert bool(torch.isfinite(x).all()), "non-finite centres after the densification of iteration %d" % it
            assert moved == 0, "rows are not in spatial order after the re-layout of iteration %d: %d of %d rows out of place (%r)" % (
                it, moved, x.shape[0], out["densified"])
    if it % 100 == 0:
        tr.sync()
        loss = out["loss"]
        assert torch.isfinite(loss).all() and bool(torch.isfinite(tr.model.flat).all()), it
        o = tr.model.optimizer
        kept = o.dormant_flags().clone()
        o.invalidate_dormant()
        derived = o.dormant_flags().clone()
        assert bool(((kept == 0) | (derived == 1)).all()), "a dormant flag is not true of the moments (iteration %d)" % it
        print("iteration %d: loss %.5f, P %d, dormant blocks %d kept / %d derived of %d" % (
            it, float(loss), tr.model.P, int(kept.sum()), int(derived.sum()), kept.numel()), flush=True)
tr.sync()
torch.cuda.synchronize()
print("%d iterations in %.1f s; P %d -> %d; densifications / resets: %r" % (n, time.time() - t0, P0, tr.model.P, events))
print("depth-limit stats", be.depth_limit_stats, "two-phase launches", be.two_phase_launches)
