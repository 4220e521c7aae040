"""Developer tool: the drop-in loop (gsplat_amd/dropin.py) at C3 size on its own - for rocprofv3 --kernel-trace --stats.
    python tests/tools/dropin_probe.py [torch|fused|fused_key] [iterations]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sparse-view-3dgs-pack_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from gsplat_amd.dropin import DropInLoop  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "torch"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
tr, scene, cams, gts = bench.build_workload("c3", dev, 0, 1)
del tr
loop = DropInLoop(scene, cams, gts, dev, dwt=True, patch=True, optimizer="torch" if mode == "torch" else "fused",
                  use_camera_key=mode == "fused_key")
for j in range(len(cams) + 2):
    loop.iteration(j % len(cams))
torch.cuda.synchronize()
from gsplat_amd import hip_backend  # noqa: E402
be = hip_backend()
d0 = dict(be.depth_limit_stats)
t0 = time.perf_counter()
for j in range(n):
    loop.iteration((j + 2) % len(cams))
torch.cuda.synchronize()
print("%s: %.3f ms/step" % (mode, (time.perf_counter() - t0) / n * 1e3), "limits used/failed",
      be.depth_limit_stats["used"] - d0["used"], be.depth_limit_stats["failed"] - d0["failed"], "binning", be.binning,
      "capacity hint", be._capacity_hint, be._capacity_hint_limited)
