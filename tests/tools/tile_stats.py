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

# Launch order of the one-wave-per-tile blend kernels: a wave runs at the pace of its dependent chain, so a tile costs
# ~ its visited entries and a kernel ends with its last wave.  List-scheduling model: S wave slots, tiles handed out in
# the given order to the first free slot; efficiency = mean slot load / makespan.
import heapq  # noqa: E402


def makespan(order, work, slots):
    h = [0.0] * slots
    heapq.heapify(h)
    for t in order:
        heapq.heappush(h, heapq.heappop(h) + float(work[t]))
    return max(h)


T = len(lmax)
per = (T + 7) // 8
image_order = [t for t in ((b & 7) * per + (b >> 3) for b in range(per * 8)) if t < T]
lpt = np.argsort(-lmax, kind="stable")
print("corr(list length, visited) = %.3f" % np.corrcoef(n, lmax)[0, 1])
for S, what in ((4096, "backward, 4 waves/SIMD"), (7168, "forward, 7 waves/SIMD"), (8192, "8 waves/SIMD")):
    avg = lmax.sum() / S
    a, b_ = makespan(image_order, lmax, S), makespan(lpt, lmax, S)
    print("S = %d (%s): mean slot load %.0f | image order: makespan %.0f (%.0f %% busy) | longest first: %.0f (%.0f %% busy)" % (
        S, what, avg, a, 100 * avg / a, b_, 100 * avg / b_))

# What a two-chunk emission would process: chunk A = the nearest f*P Gaussians of the depth order,
# chunk B = the rest, only for tiles some pixel of which is not saturated when A's part of the list ends.
pl = st["point_list"].long()
depth = st["depths"].float()
vis = st["radii"] > 0
key = torch.where(vis, depth, torch.full_like(depth, float("inf")))
order = torch.argsort(key, stable=True)
rank = torch.empty_like(order)
rank[order] = torch.arange(P)
inst_rank = rank[pl]                                   # depth rank of every instance, in list order
tile_of = torch.repeat_interleave(torch.arange(len(n)), torch.from_numpy(n))
T = st["final_T"].reshape(H, W)
padT = torch.zeros(((H + 15) // 16 * 16, gx * 16))
padT[:H, :W] = T
# a tile is "open-ended" if a pixel never saturated (it consumes its whole list).  final_T is the transmittance BEFORE
# the entry that would have pushed it below 1e-4 (alpha <= 0.99), so a stopped pixel has final_T < 1e-2
tile_maxT = padT.reshape(-1, 16, gx, 16).permute(0, 2, 1, 3).reshape(-1, 256).max(dim=1).values
R = int(n.sum())
for f in (0.05, 0.1, 0.2, 0.3, 0.5):
    K = int(f * P)
    in_a = inst_rank < K
    a_len = torch.zeros(len(n), dtype=torch.long).index_add_(0, tile_of, in_a.long())
    complete = (torch.from_numpy(lmax) <= a_len) & (tile_maxT < 1e-2)
    complete |= a_len == torch.from_numpy(n)             # nothing left for B anyway
    r_a = int(in_a.sum())
    r_b = int((torch.from_numpy(n) - a_len)[~complete].sum())
    print("f=%.2f  R_A %.2f M (%.0f %%)  tiles needing B %d of %d  R_B %.2f M (%.0f %%)  A+B %.0f %% of R" %
          (f, r_a / 1e6, 100 * r_a / R, int((~complete).sum()), len(n), r_b / 1e6, 100 * r_b / R, 100 * (r_a + r_b) / R))
print("tiles with an unsaturated pixel: %d of %d" % (int((tile_maxT >= 1e-2).sum()), len(n)))

# load balance of the one-wave-per-tile kernels over the 8 XCDs (work ~ entries visited per tile)
gy = (H + 15) // 16
work = torch.from_numpy(lmax).double().reshape(gy, gx)
ntile = gx * gy
per = (ntile + 7) // 8
flat = work.reshape(-1)
band = [float(flat[x * per:(x + 1) * per].sum()) for x in range(8)]
rows = [float(work[x::8].sum()) for x in range(8)]
cols = [float(flat[x::8].sum()) for x in range(8)]
for name, v in (("contiguous bands (current)", band), ("tile rows interleaved", rows), ("tiles interleaved", cols)):
    print("XCD work, %-28s max/mean %.3f   %s" % (name, max(v) / (sum(v) / 8), " ".join("%.0f" % (x / 1e3) for x in v)))
