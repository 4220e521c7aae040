"""Developer tool: where the HOST time of one eager train step goes (cProfile over 200 steps of the benched C3 configuration).
   python tests/tools/host_profile.py [steps] [lines]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sparse-view-3dgs-pack_amd"))
import torch  # noqa: E402

import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lines = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
tr, scene, cams, gts = bench.build_workload("c3", dev, 0, 1)
tr.depth_limit = "deferred"
k = 0
for _ in range(40):
    tr.step(k)
    k += 1
tr.sync()
torch.cuda.synchronize()
import gc
gc.collect()
gc.disable()
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    tr.step(k)
    k += 1
pr.disable()
tr.sync()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(lines)
