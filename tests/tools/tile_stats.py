"""Developer tool: distribution of per-tile list lengths / last contributors at the bench workload (load balance of
the one-wave-per-tile blend kernels)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sparse-view-3dgs-pack_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gsplat_amd import hip_backend, synthetic  # noqa: E402
from simple_knn._C import distCUDA2  # noqa: E402
from test_gpu_raster_parity import forward_state  # noqa: E402

P, W, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda")
sc = synthetic.trained_like(P, seed=0, sh_degree=3, knn=lambda x: distCUDA2(x.to(dev)).cpu())
cam = synthetic.orbit_cameras(W, H)[3]
st = forward_state(hip_backend(), sc, cam, dev, torch.zeros(3), False)
r = st["ranges"].reshape(-1, 2).long()
n = (r[:, 1] - r[:, 0]).numpy()
gx = (W + 15) // 16
nc = st["n_contrib"].reshape(H, W).long()
pad = torch.zeros(((H + 15) // 16 * 16, gx * 16), dtype=torch.long)
pad[:H, :W] = nc
lmax = pad.reshape(-1, 16, gx, 16).permute(0, 2, 1, 3).reshape(-1, 256).max(dim=1).values.numpy()
print("tiles %d  R %d" % (len(n), n.sum()))
for name, a in (("list length", n), ("entries visited by bwd (max last contributor)", lmax)):
    print("%-48s mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %d  sum/4096 slots %.0f" %
          (name, a.mean(), np.percentile(a, 50), np.percentile(a, 90), np.percentile(a, 99), a.max(), a.sum() / 4096))
