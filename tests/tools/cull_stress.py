"""Developer tool: randomised stress of the exact-safe tile cull (csrc/gs_tilecull.h) - many cameras / scenes / opacity and
scale regimes, each checked with tests/test_gpu_tilecull.check_lists (kept pairs = subsequence of the reference list,
every dropped pair dead on all 256 pixels)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sparse-view-3dgs-pack_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle_lib  # noqa: E402
from gsplat_amd import hip_backend, synthetic  # noqa: E402
from test_gpu_raster_parity import forward_state  # noqa: E402
from test_gpu_tilecull import check_lists  # noqa: E402

hip, orc = hip_backend(), oracle_lib.get()
hip.tile_cull = True
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
worst = 0.0
for it in range(n):
    P = int(rng.choice([500, 3000, 12000]))
    W, H = int(rng.choice([97, 256, 640])), int(rng.choice([64, 200, 360]))
    kind = rng.choice(["init", "trained"])
    sc = (synthetic.init_like if kind == "init" else synthetic.trained_like)(P, seed=int(rng.randint(1000)), sh_degree=0)
    regime = rng.choice(["plain", "needles", "lowop", "huge", "tiny"])
    g = torch.Generator().manual_seed(int(rng.randint(1 << 30)))
    if regime == "needles":
        sc["scales"] = sc["scales"] * torch.tensor([8.0, 0.05, 0.05])
    elif regime == "lowop":
        sc["opacities"] = torch.rand((P, 1), generator=g) * 0.02
    elif regime == "huge":
        sc["scales"] = sc["scales"] * 6.0
    elif regime == "tiny":
        sc["scales"] = sc["scales"] * 0.1
    r = float(rng.uniform(0.3, 5.0))  # from inside the cloud to far away
    th, ph = rng.uniform(0, 2 * np.pi), rng.uniform(-1.2, 1.2)
    eye = (r * np.cos(th) * np.cos(ph), r * np.sin(th) * np.cos(ph), r * np.sin(ph))
    cam = synthetic.look_at_camera(eye, W, H, FoVx=float(rng.uniform(0.4, 1.6)))
    aa = bool(rng.randint(2))
    bg = torch.zeros(3)
    h = forward_state(hip, sc, cam, torch.device("cuda"), bg, aa)
    o = forward_state(orc.backend, sc, cam, torch.device("cpu"), bg, aa)
    info = check_lists(h, o, W, H, "stress%d" % it)
    assert torch.equal(h["radii"], o["radii"])
    worst = max(worst, info["closest_alpha_x255"])
    print(it, kind, regime, P, "%dx%d" % (W, H), "aa" if aa else "  ", "r=%.2f" % r, "kept %d / %d" % (info["kept"], info["reference"]),
          "closest %.4f" % info["closest_alpha_x255"], flush=True)
print("ALL OK, closest dropped alpha*255 =", worst)
