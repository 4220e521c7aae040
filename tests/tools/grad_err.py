"""Developer tool: per-tensor gradient error of the HIP backward vs the oracle on one config-sized scene."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sparse-view-3dgs-pack_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import diff_gaussian_rasterization as dgr  # noqa: E402
import oracle_lib  # noqa: E402
from gsplat_amd import synthetic  # noqa: E402
from helpers import run_scene  # noqa: E402

kind, P, W, H, deg = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
orc = oracle_lib.get()
sc = (synthetic.init_like if kind == "init" else synthetic.trained_like)(P, seed=0, sh_degree=deg)
cam = synthetic.orbit_cameras(W, H)[3]
g = torch.Generator().manual_seed(5)
dL = torch.randn((3, H, W), generator=g)
if len(sys.argv) > 6 and sys.argv[6] == "mask":  # what the parity tests do: drop pixels whose threshold decision flipped
    from gsplat_amd import hip_backend
    from test_gpu_raster_parity import flip_mask, forward_state
    hb = hip_backend()
    h0 = forward_state(hb, sc, cam, torch.device("cuda"), torch.zeros(3), False)
    o0 = forward_state(orc.backend, sc, cam, torch.device("cpu"), torch.zeros(3), False)
    keep = ~flip_mask(h0, o0) if h0["num_rendered"] == o0["num_rendered"] else None
    if keep is None:
        hb.tile_cull = False
        h0 = forward_state(hb, sc, cam, torch.device("cuda"), torch.zeros(3), False)
        keep = ~flip_mask(h0, o0)
        hb.tile_cull = True
    print("flipped pixels:", int((~keep).sum()))
    dL = dL * keep.float()
ho = run_scene(dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, sc, cam, torch.device("cuda"), dL_dcolor=dL)
oo = run_scene(orc.Rasterizer, orc.Settings, sc, cam, torch.device("cpu"), dL_dcolor=dL)
for k in oo["grads"]:
    a, b = ho["grads"][k].cpu().double(), oo["grads"][k].double()
    scale = max(float(b.abs().max()), 1e-12)
    rms = float((a - b).pow(2).mean().sqrt() / max(float(b.pow(2).mean().sqrt()), 1e-20))
    print("%-10s max-rel %.3e  rms-rel %.3e  scale %.2e" % (k, float((a - b).abs().max()) / scale, rms, scale))
