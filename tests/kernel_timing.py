"""Developer tool (not a test): times the forward / backward of the rasterizer alone on the C3 scene with the
library's HIP-event stage timers.  usage: python tests/kernel_timing.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-view-3dgs-pack_amd"))
import torch  # noqa: E402

import diff_gaussian_rasterization as dgr  # noqa: E402
from gsplat_amd import synthetic  # noqa: E402
from gsplat_amd._lib import hip_api  # noqa: E402
from gsplat_amd.capi import read_profile  # noqa: E402
from gsplat_amd.trainer import GaussianModelLite, camera_to, render  # noqa: E402
from simple_knn._C import distCUDA2  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
P, W, H = int(os.environ.get("GS_P", 1000000)), 1920, 1080
dev = torch.device("cuda")
sc = synthetic.trained_like(P, seed=0, knn=lambda x: distCUDA2(x.to(dev)).cpu())
model = GaussianModelLite(sc, dev, api=hip_api())
cams = [camera_to(c, dev) for c in synthetic.orbit_cameras(W, H)]
api = hip_api()
g = torch.Generator().manual_seed(0)
dL = torch.randn((3, H, W), generator=g).to(dev)
warm = 26 if os.environ.get("GS_KT_KEYED") else 3   # (keyed: every camera visited once, later visits are depth-limited)
for it in range(reps + warm):
    if it == warm:
        torch.cuda.synchronize()
        api.call("profile_reset")
        api.call("profile_enable", 1)
    model.zero_grad()
    pkg = render(cams[it % 24], model, dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, torch.zeros(3, device=dev),
                 filter_as_indices=False, camera_key=("kt", it % 24) if os.environ.get("GS_KT_KEYED") else None)
    (pkg["render"] * dL).sum().backward()
torch.cuda.synchronize()
api.call("profile_enable", 0)
for k, (ms, n) in read_profile(api).items():
    print("%-16s %8.4f ms x %d" % (k, ms / n, n))
