"""Developer tool (verdict r04 item 6): of the Gaussians whose SH row preprocess_fwd reads on a depth-limited, region-binned C3
view, how many end with ZERO list entries after region_bin's exact per-tile test?  (preprocess already skips the SH row of a
Gaussian that no REGION accepts.)   python tests/tools/sh_read_probe.py [c3]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sparse-view-3dgs-pack_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from gsplat_amd import hip_backend  # noqa: E402
from test_gpu_raster_parity import forward_state  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
dev = torch.device("cuda", 0)
tr, scene, cams, gts = bench.build_workload(cfg, dev, 0, 1)
be = hip_backend()
be.tile_cull, be.binning, be.depth_limit_on = True, "region", True
be._cam_cache.clear()
P = scene["means3D"].shape[0]
tot = dict(visible=0, region=0, listed=0, R=0)
for ci in (0, 5, 11, 17):
    cam = cams[ci]
    full = forward_state(be, scene, cam, dev, torch.zeros(3), False)      # first visit: measures the stop depths
    cut = forward_state(be, scene, cam, dev, torch.zeros(3), False)       # second visit: depth-limited lists
    vis = int((cut["radii"] > 0).sum())
    reg = int((cut["tiles_touched"] > 0).sum())
    listed = int(torch.unique(cut["point_list"]).numel())
    print("camera %2d: P %d | visible %d | accepted by a region (SH row read) %d | with list entries %d | R %d (full %d) | "
          "SH rows read for nothing: %.1f %%" % (ci, P, vis, reg, listed, cut["num_rendered"], full["num_rendered"],
                                                 100.0 * (reg - listed) / max(reg, 1)))
    for k, v in (("visible", vis), ("region", reg), ("listed", listed), ("R", cut["num_rendered"])):
        tot[k] += v
print("all: SH rows read %d, of them without list entries %d = %.1f %%" % (tot["region"], tot["region"] - tot["listed"],
                                                                           100.0 * (tot["region"] - tot["listed"]) / tot["region"]))
