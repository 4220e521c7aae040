"""How robust are per-tile depth limits (csrc/gs_tilecull.h) on the bench workload, and what do they save?

Runs the bench's eager train step (limits OFF) and, before every step, renders that step's camera once more to read the
depth at which each tile's blend stops now.  Against the stop depths of the camera's previous visit it evaluates, for a
few (relative, absolute) margins and with / without taking each tile's bound as the maximum over its 3 x 3 neighbourhood:
  fail  - would any tile with a finite bound now stop beyond it (the forward's verdict)?
  kept  - fraction of the instance list that survives the bound (per-pair test; the span rule keeps a few more).

    python tests/tools/depth_limit_probe.py [config] [steps]
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sparse-view-3dgs-pack_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from gsplat_amd import hip_backend  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 80
dev = torch.device("cuda")
tr, scene, cams, gts = bench.build_workload(cfg, dev, 0, 1)
be = hip_backend()
P, W, H = bench.CONFIGS[cfg][:3]
gx, gy = (W + 15) // 16, (H + 15) // 16
T = gx * gy
VARIANTS = [(rel, ab, dil) for dil in (0, 1) for (rel, ab) in ((1.05, 0.02), (1.10, 0.05), (1.20, 0.10), (1.40, 0.20))]


def measure(ci):
    """-> (stop[T], tile of every list entry, depth of every list entry) of camera ci at the current parameters"""
    m = tr.model
    cam = cams[ci]
    with torch.no_grad():
        args = (tr.bg, m.get_xyz.detach(), torch.empty(0, device=dev), m.get_opacity.detach(), m.get_scaling.detach(),
                m.get_rotation.detach(), 1.0, torch.empty(0, device=dev), cam.world_view_transform, cam.full_proj_transform,
                cam.tanfovx, cam.tanfovy, cam.image_height, cam.image_width, m.get_features.detach(), m.active_sh_degree,
                cam.camera_center, False, False, False)
        R, color, radii, geom, binning, img, invd = be.rasterize_gaussians(*args)
        s = be._scratch(geom, img, binning, be._capacity_for(binning, P, W, H, R))
        out = torch.empty((3 * T,), dtype=torch.float32, device=dev)
        be.api.call("export_tile_stop_depth", C.byref(s), W, H, out.data_ptr(), be._stream(dev))
        st = be.export_state(P, W, H, R, geom, binning, img)
        tile = (st["keys_sorted"] >> 32).long()
        depth = st["depths"][st["point_list"].long()]
        return out[:T].clone(), tile, depth


def dilate(stop):
    g = stop.reshape(1, 1, gy, gx)
    return torch.nn.functional.max_pool2d(g, 3, 1, 1).reshape(-1)


prev = {}
rows = []
for k in range(steps):
    ci = tr.camera_index(k)
    stop, tile, depth = measure(ci)
    if ci in prev:
        old = prev[ci]
        rec = dict(step=k, camera=ci, instances=int(tile.numel()), tiles_with_limit=int(torch.isfinite(old).sum()))
        for rel, ab, dil in VARIANTS:
            base = dilate(old) if dil else old
            bound = base * rel + ab
            failing = torch.isfinite(base) & ~(stop <= bound)
            kept = float((depth <= bound[tile]).float().mean())
            rec["rel%.2f_abs%.2f_dil%d" % (rel, ab, dil)] = dict(fail_tiles=int(failing.sum()), kept=round(kept, 4),
                                                                 limited_tiles=int(torch.isfinite(base).sum()))
        rows.append(rec)
    prev[ci] = stop
    tr.step(k)
torch.cuda.synchronize()
summary = {}
for rel, ab, dil in VARIANTS:
    key = "rel%.2f_abs%.2f_dil%d" % (rel, ab, dil)
    v = [r[key] for r in rows]
    summary[key] = dict(views=len(v), views_failing=sum(1 for x in v if x["fail_tiles"] > 0),
                        mean_kept=sum(x["kept"] for x in v) / max(1, len(v)),
                        mean_fail_tiles=sum(x["fail_tiles"] for x in v) / max(1, len(v)))
print(json.dumps(dict(config=cfg, steps=steps, cameras=len(cams), T=T, summary=summary), indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(dict(config=cfg, summary=summary, rows=rows), open(os.path.join(ROOT, "gpurun_out", "depth_limit_probe_%s.json" % cfg), "w"))
