"""Developer tool: where an iteration of the drop-in loop (gsplat_amd/dropin.py, the render_raw + FusedAdam + camera_key +
lgdwt_loss.criterion() variant bench.py reports) spends its time at BASELINE C3.
   python3 tests/tools/dropin_probe.py [c3] [iters] [cprofile|trace]
   cd /tmp && rocprofv3 --kernel-trace --stats -d <out> -- python3 <repo>/tests/tools/dropin_probe.py c3 40
prints wall ms / iteration with and without the reference's `loss.item()` (train.py:224), the host-only enqueue time of an
iteration (no sync at all), and with `cprofile` the twenty heaviest host functions."""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sparse-view-3dgs-pack_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from gsplat_amd.dropin import DropInLoop  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
mode = sys.argv[3] if len(sys.argv) > 3 else ""
dev = torch.device("cuda", 0)
tr, scene, cams, gts = bench.build_workload(cfg, dev, 0, 1)
del tr
loop = DropInLoop(scene, cams, gts, dev, dwt=True, patch=True, optimizer="fused", use_camera_key=True, fused_criterion=True,
                  raw_render=True)
n = len(cams)
for j in range(n + 2):
    loop.iteration(j % n)
torch.cuda.synchronize()
gc.collect()
gc.disable()
prof = None
if mode == "cprofile":
    import cProfile
    prof = cProfile.Profile()
    prof.enable()
t0 = time.perf_counter()
for j in range(iters):
    loop.iteration((j + 2) % n)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
if prof is not None:
    import io
    import pstats
    prof.disable()
    buf = io.StringIO()
    pstats.Stats(prof, stream=buf).sort_stats("tottime").print_stats(25)
    print(buf.getvalue())
print("drop-in raw loop: %.3f ms / iteration (loss.item() every iteration)" % (dt * 1e3))

# A/B in ONE process (boxes differ by 5-10 %): where the loss.item() of train.py:224 sits, or no host read of the loss at all
def finish(order):
    def f(loss, vsp, vis, radii):
        pc = loop.pc
        loss.backward()
        with torch.no_grad():
            value = loss.item() if order == "after_backward" else 0.0
            vf = vis.squeeze(1)
            pc.max_radii2D[vf] = torch.max(pc.max_radii2D[vf], radii[vf].to(torch.float32))
            pc.add_densification_stats(vsp, vf)
            loop.optimizer.step()
            loop.optimizer.zero_grad(set_to_none=True)
            if order == "last":
                value = loss.item()
        return value
    return f


for rep in range(2):
    for order in ("after_backward", "last", "none"):
        loop._finish = finish(order)
        for j in range(4):
            loop.iteration(j % n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(iters):
            loop.iteration((j + 2) % n)
        t_host = (time.perf_counter() - t0) / iters
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / iters
        print("loss.item() %-15s %.3f ms / iteration (host side alone %.3f ms)" % (order, dt2 * 1e3, t_host * 1e3))
gc.enable()
