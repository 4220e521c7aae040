"""Developer tool: distCUDA2 timing."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sparse-view-3dgs-pack_amd"))
import torch  # noqa: E402

from simple_knn._C import distCUDA2  # noqa: E402
from sknn_fsgs import distCUDA2 as dist_idx  # noqa: E402

for P in (100_000, 1_000_000, 4_000_000):
    g = torch.Generator().manual_seed(0)
    pts = (torch.rand((P, 3), generator=g) * 2.6 - 1.3).cuda()
    for name, fn in (("distCUDA2", distCUDA2), ("with indices", dist_idx)):
        for _ in range(2):
            fn(pts)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            fn(pts)
        torch.cuda.synchronize()
        print("P=%8d  %-13s %.3f ms" % (P, name, (time.perf_counter() - t) * 200), flush=True)
