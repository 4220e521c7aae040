"""Developer tool: stage timers (HIP events) of the eager train step at a BASELINE size, nothing else.
   python3 tests/tools/step_probe.py [c3] [steps]
A/B of kernel builds on ONE box: copy a variant library over csrc/libgsplat_hip.so between two runs of this script (the
gpurun copy of the tree is scratch)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sparse-view-3dgs-pack_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from gsplat_amd._lib import hip_api  # noqa: E402
from gsplat_amd.capi import read_profile  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda", 0)
tr, scene, cams, gts = bench.build_workload(cfg, dev, 0, 1)
api = hip_api()
k = 0
for _ in range(len(cams) + 4):
    tr.step(k)
    k += 1
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(k)
    k += 1
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
api.call("profile_reset")
api.call("profile_enable", 1)
for _ in range(steps):
    tr.step(k)
    k += 1
torch.cuda.synchronize()
api.call("profile_enable", 0)
print("%s: %.4f ms / step; stages (ms / launch):" % (cfg, dt * 1e3),
      {n: round(v[0] / v[1], 4) for n, v in read_profile(api).items() if v[1]})
