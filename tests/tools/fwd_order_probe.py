#!/usr/bin/env python3
"""Does a longest-tile-first launch order help the FORWARD blend?  Times render_fwd (library HIP-event profiler) at C3
for: no hint (image order), the previous view's order of the SAME camera, and the order of a DIFFERENT camera."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "sparse-view-3dgs-pack_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import fullsize_parity as fp  # noqa: E402
from gsplat_amd import hip_backend, synthetic  # noqa: E402
from gsplat_amd.capi import read_profile  # noqa: E402
from simple_knn._C import distCUDA2  # noqa: E402

P, W, H = 1_000_000, 1920, 1080
dev = torch.device("cuda")
hip = hip_backend()
api = hip.api
sc = synthetic.trained_like(P, seed=0, sh_degree=3, knn=lambda x: distCUDA2(x.to(dev)).cpu())
cams = synthetic.orbit_cameras(W, H)
bg = torch.zeros(3)


def timed(cam_seq, hint_on, n=10):
    hip.order_hint_on = hint_on
    hip._cam_cache.clear()
    for c in cam_seq[:2]:
        fp.forward(hip, sc, cams[c], dev, bg)
    torch.cuda.synchronize()
    api.call("profile_reset")
    api.call("profile_only", -1)
    api.call("profile_enable", 1)
    for i in range(n):
        fp.forward(hip, sc, cams[cam_seq[(2 + i) % len(cam_seq)]], dev, bg)
    torch.cuda.synchronize()
    api.call("profile_enable", 0)
    pr = read_profile(api)
    return pr["render_fwd"][0] / pr["render_fwd"][1]


for name, seq, hint in (("no hint, camera 3", [3], False), ("hint = same camera (3)", [3], True),
                        ("no hint, cameras 3,4 alternating", [3, 4], False),
                        ("hint = the other camera (3,4 alternating)", [3, 4], True),
                        ("no hint, cameras 3,12 alternating", [3, 12], False),
                        ("hint = the other camera (3,12 alternating)", [3, 12], True)):
    print("%-45s render_fwd %.4f ms" % (name, timed(seq, hint)), flush=True)
