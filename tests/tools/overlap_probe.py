"""Developer probe: how much of the flat Adam update hides behind the next forward when it runs on a second stream."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sparse-view-3dgs-pack_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
import diff_gaussian_rasterization as dgr  # noqa: E402
from gsplat_amd.trainer import render  # noqa: E402

dev = torch.device("cuda")
tr, scene, cams, gts = bench.build_workload("c3", dev, 0, 1)
m = tr.model
for k in range(3):
    tr.step(k)
P = m.P
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def fwd():
    with torch.no_grad():
        render(cams[3], m, dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, tr.bg, filter_as_indices=False, fused=True)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def adam_features(stream=None):
    if stream is None:
        m.optimizer.step_range(3 * P, 51 * P)
    else:
        stream.wait_stream(main)
        with torch.cuda.stream(stream):
            m.optimizer.step_range(3 * P, 51 * P)


def seq():
    adam_features()
    fwd()


def par():
    adam_features(side)
    fwd()
    main.wait_stream(side)


print("forward alone        %.3f ms" % timeit(fwd))
print("adam(features) alone %.3f ms" % timeit(adam_features))
print("sequential           %.3f ms" % timeit(seq))
print("two streams          %.3f ms" % timeit(par))
