"""developer tool: what the backward blend's loop meets at the bench workload (gs_debug_blend_stats).
   python tests/tools/blend_stats.py [P W H]"""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sparse-view-3dgs-pack_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fullsize_parity as fp  # noqa: E402
from gsplat_amd import hip_backend, synthetic  # noqa: E402
from simple_knn._C import distCUDA2  # noqa: E402

P, W, H = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (1_000_000, 1920, 1080)
dev = torch.device("cuda")
hip = hip_backend()
res = {}
for kind in ("trained_like", "init_like", "ball_in_shell"):
    sc = getattr(synthetic, kind)(P, seed=0, sh_degree=3, knn=lambda x: distCUDA2(x.to(dev)).cpu())
    cam = synthetic.orbit_cameras(W, H)[3]
    fw = fp.forward(hip, sc, cam, dev, torch.zeros(3))
    out = torch.zeros(8, dtype=torch.int64, device=dev)
    s = hip._scratch(fw["geom"], fw["img"], fw["binning"], hip._capacity_for(fw["binning"], P, W, H, fw["R"]))
    hip.api.call("debug_blend_stats", C.byref(s), P, W, H, out.data_ptr(), None)
    torch.cuda.synchronize()
    v = [int(x) for x in out.cpu()]
    r = dict(num_rendered=fw["R"], visited=v[0], with_valid_pixel=v[1], live_quadrants=v[2], valid_pixels=v[3], tiles=v[4],
             list_entries=v[5], visited_per_tile=v[0] / max(1, v[4]), frac_entries_live=v[1] / max(1, v[0]),
             quadrants_per_visited_entry=v[2] / max(1, v[0]), quadrants_per_live_entry=v[2] / max(1, v[1]),
             pixels_per_live_quadrant=v[3] / max(1, v[2]))
    res[kind] = r
    print(kind, json.dumps(r))
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "blend_stats.json"), "w"), indent=1) if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None
