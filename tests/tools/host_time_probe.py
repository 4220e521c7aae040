"""Developer tool: how long does the HOST need to enqueue one train step (C3, eager, deferred limits)?  If that is close to
the step time the GPU waits for Python now and then.   python tests/tools/host_time_probe.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sparse-view-3dgs-pack_amd"))
import torch  # noqa: E402

import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
tr, scene, cams, gts = bench.build_workload("c3", dev, 0, 1)
tr.depth_limit = "deferred"
k = 0
for _ in range(30):
    tr.step(k)
    k += 1
tr.sync()
torch.cuda.synchronize()
import gc
gc.collect()
gc.disable()
for _ in range(5):
    tr.step(k)
    k += 1
torch.cuda.synchronize()
t0 = time.perf_counter()
per = []
for _ in range(n):
    a = time.perf_counter()
    tr.step(k)
    k += 1
    per.append(time.perf_counter() - a)
t1 = time.perf_counter()
tr.sync()
torch.cuda.synchronize()
t2 = time.perf_counter()
per.sort()
print("host enqueue per step: mean %.3f ms, median %.3f, min %.3f, max %.3f" % ((t1 - t0) / n * 1e3, per[n // 2] * 1e3, per[0] * 1e3, per[-1] * 1e3))
print("wall per step incl. final drain: %.3f ms" % ((t2 - t0) / n * 1e3))
# the same with the GPU idle between steps: pure host cost of a step (nothing to wait for)
per2 = []
for _ in range(10):
    torch.cuda.synchronize()
    a = time.perf_counter()
    tr.step(k)
    k += 1
    per2.append(time.perf_counter() - a)
per2.sort()
print("host cost of a step with an idle GPU (includes the deferred verdict's wait for the previous forward: none): median %.3f ms, min %.3f" % (per2[5] * 1e3, per2[0] * 1e3))
