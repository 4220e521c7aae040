"""Developer tool: a few thousand train steps of the benched configuration in each launch form (eager two-phase step with
deferred depth limits; one hipGraph per camera), watching for anything a 20-step bench cannot show: non-finite losses or
parameters, a counter that drifts, fall-back storms, host memory growth.   python tests/tools/soak.py [steps] [config]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sparse-view-3dgs-pack_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from gsplat_amd import hip_backend  # noqa: E402
from gsplat_amd.trainer import GraphedStep  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cfg = sys.argv[2] if len(sys.argv) > 2 else "c3"
target = sys.argv[3] if len(sys.argv) > 3 else "unrelated"   # unrelated (bench.py's: renders of another random scene) | near
dev = torch.device("cuda", 0)
gts = None
if target == "near":
    # a run that is converging: the ground truth is a render of the SAME scene with slightly different colours
    import diff_gaussian_rasterization as dgr
    from gsplat_amd import synthetic
    from gsplat_amd._lib import hip_api
    from gsplat_amd.trainer import GaussianModelLite, camera_to, render
    from simple_knn._C import distCUDA2
    P, W, H = bench.CONFIGS[cfg][:3]
    sc = synthetic.trained_like(P, seed=0, knn=lambda x: distCUDA2(x.to(dev)).cpu(), sh_degree=3)
    g = torch.Generator().manual_seed(5)
    tgt = dict(sc, shs=sc["shs"] + 0.02 * torch.randn(sc["shs"].shape, generator=g))
    tm = GaussianModelLite(tgt, dev, api=hip_api())
    cams_ = [camera_to(c, dev) for c in synthetic.orbit_cameras(W, H)]
    with torch.no_grad():
        gts = [render(c, tm, dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, torch.zeros(3, device=dev))["render"].clone()
               for c in cams_]
    del tm
tr, scene, cams, gts = bench.build_workload(cfg, dev, 0, 1, gts=gts)
tr.depth_limit = "deferred"
be = hip_backend()
k = 0
forms = ("graph", "eager") if (len(sys.argv) > 4 and sys.argv[4] == "graph_first") else ("eager", "graph")
for form in forms:
    gs = GraphedStep(tr) if form == "graph" else None
    s0 = dict(be.depth_limit_stats)
    t0 = time.time()
    losses = []
    for i in range(n):
        loss = gs.step(k) if gs is not None else tr.step(k)
        k += 1
        if i % 100 == 99:
            (gs or tr).sync()
            losses.append(float(loss))
            assert torch.isfinite(loss).all(), (form, i)
    (gs or tr).sync()
    torch.cuda.synchronize()
    dt = time.time() - t0
    flat = tr.model.flat.detach()
    assert bool(torch.isfinite(flat).all()), form
    s1 = be.depth_limit_stats
    per100 = None
    print("%s: %d steps, %.3f ms/step, loss %.5f -> %.5f, limited views %d, fall-backs %d, optimizer t %d, two-phase launches %d%s"
          % (form, n, dt / n * 1e3, losses[0], losses[-1], s1["used"] - s0["used"], s1["failed"] - s0["failed"],
             tr.model.optimizer.t, be.two_phase_launches,
             "" if gs is None else ", captures %d, replays %d, eager steps %d, failures %r" % (gs.captures, gs.replays, gs.eager_steps, gs.fail_kinds)))
print("max memory allocated %.2f GB" % (torch.cuda.max_memory_allocated() / 2 ** 30))
