"""Developer tool: the drop-in loop (gsplat_amd/dropin.py) at C3 size on its own - for rocprofv3 --kernel-trace --stats.
    python tests/tools/dropin_probe.py [torch|fused|fused_key|fused_key_crit ...] (several modes run one after the other)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sparse-view-3dgs-pack_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from gsplat_amd import hip_backend  # noqa: E402
from gsplat_amd._lib import hip_api  # noqa: E402
from gsplat_amd.capi import read_profile  # noqa: E402
from gsplat_amd.dropin import DropInLoop  # noqa: E402

modes = sys.argv[1:] or ["torch"]
n = 20
dev = torch.device("cuda", 0)
tr, scene, cams, gts = bench.build_workload("c3", dev, 0, 1)
if os.environ.get("PROBE_TRAINER_FIRST"):
    tr.depth_limit = "deferred"
    for k in range(30):
        tr.step(k)
    tr.sync()
if os.environ.get("PROBE_GRAPH"):       # ... and replayed from hipGraphs, as bench.py's launch trial does
    from gsplat_amd.trainer import GraphedStep
    gs_ = GraphedStep(tr)
    for k in range(60):
        gs_.step(k)
    gs_.sync()
if os.environ.get("PROBE_REFLISTS"):    # ... and the reference-lists leg of bench.py (tile_cull off for a dozen steps, then back)
    be0 = hip_backend()
    old = be0.tile_cull
    be0.tile_cull = False
    for k in range(12):
        tr.step(100 + k)
    torch.cuda.synchronize()
    be0.tile_cull = old
    for k in range(2):
        tr.step(120 + k)
    tr.sync()
    print("after the reference-lists leg: capacity hints", be0._capacity_hint, be0._capacity_hint_limited, flush=True)
if not os.environ.get("PROBE_KEEP_TRAINER"):
    del tr
be, api = hip_backend(), hip_api()
for mode in modes:
    loop = DropInLoop(scene, cams, gts, dev, dwt=True, patch=True, optimizer="torch" if mode == "torch" else "fused",
                      use_camera_key="key" in mode, fused_criterion="crit" in mode)
    for j in range(len(cams) + 2):
        loop.iteration(j % len(cams))
    torch.cuda.synchronize()
    d0 = dict(be.depth_limit_stats)
    prof_ = None
    if os.environ.get("PROBE_CPROFILE"):
        import cProfile
        prof_ = cProfile.Profile()
        prof_.enable()
    t0 = time.perf_counter()
    for j in range(n):
        loop.iteration((j + 2) % len(cams))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    if prof_ is not None:
        import pstats
        prof_.disable()
        pstats.Stats(prof_).sort_stats("tottime").print_stats(12)
    api.call("profile_reset")
    api.call("profile_enable", 1)
    for j in range(6):
        loop.iteration((j + 2) % len(cams))
    torch.cuda.synchronize()
    api.call("profile_enable", 0)
    st = {k: round(v[0] / v[1], 4) for k, v in read_profile(api).items()}
    print("%s: %.3f ms/step" % (mode, ms), "limits used/failed", be.depth_limit_stats["used"] - d0["used"],
          be.depth_limit_stats["failed"] - d0["failed"], "capacity hints", be._capacity_hint, be._capacity_hint_limited)
    print("   stages", st, "sum %.3f" % sum(st.values()))
    del loop
