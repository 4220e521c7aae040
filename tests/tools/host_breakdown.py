"""Developer tool: the host time of one eager train step (C3) by piece, with the GPU idle at the start of every step (so nothing
waits): the rasterizer forward glue, the criterion's forward, the criterion's backward (incl. the side launch), the rasterizer
backward glue, the rest (autograd engine, trainer bookkeeping).   python tests/tools/host_breakdown.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sparse-view-3dgs-pack_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from gsplat_amd import hip_backend  # noqa: E402
from gsplat_amd.losses import FusedLGDWTLoss  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
tr, scene, cams, gts = bench.build_workload("c3", dev, 0, 1)
tr.depth_limit = "deferred"
be = hip_backend()
acc = {}


def timed(obj, name, label):
    f = getattr(obj, name)

    def w(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t
    setattr(obj, name, w)


timed(be, "rasterize_gaussians", "raster forward glue")
timed(be, "rasterize_gaussians_backward", "raster backward glue")
timed(be, "launch_uninstanced_early", "side launch (inside the criterion's backward)")
fw, bw = FusedLGDWTLoss.forward, FusedLGDWTLoss.backward


def fwd(ctx, *a):
    t = time.perf_counter()
    try:
        return fw(ctx, *a)
    finally:
        acc["criterion forward"] = acc.get("criterion forward", 0.0) + time.perf_counter() - t


def bwd(ctx, *a):
    t = time.perf_counter()
    try:
        return bw(ctx, *a)
    finally:
        acc["criterion backward (incl. side launch)"] = acc.get("criterion backward (incl. side launch)", 0.0) + time.perf_counter() - t


FusedLGDWTLoss.forward = staticmethod(fwd)
FusedLGDWTLoss.backward = staticmethod(bwd)
k = 0
for _ in range(40):
    tr.step(k)
    k += 1
tr.sync()
import gc
gc.collect()
gc.disable()
acc.clear()
tot = 0.0
for _ in range(n):
    tr.sync()
    torch.cuda.synchronize()
    t = time.perf_counter()
    tr.step(k)
    k += 1
    tot += time.perf_counter() - t
print("host time per step (GPU idle at its start): %.3f ms" % (tot / n * 1e3))
for lab, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("   %-50s %.3f ms" % (lab, v / n * 1e3))
known = acc.get("raster forward glue", 0) + acc.get("criterion forward", 0) + acc.get("criterion backward (incl. side launch)", 0) + acc.get("raster backward glue", 0)
print("   %-50s %.3f ms" % ("the rest (autograd, render(), trainer bookkeeping)", (tot - known) / n * 1e3))
