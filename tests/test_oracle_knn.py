"""Oracle distCUDA2 vs brute force + known answers (CPU)."""
import numpy as np
import torch

from gsplat_amd import synthetic
from gsplat_amd.knn import dist2


def brute(xyz):
    d = torch.cdist(xyz.double(), xyz.double()) ** 2
    d.fill_diagonal_(float("inf"))
    return d.topk(3, dim=1, largest=False).values.mean(dim=1)


def test_unit_lattice_known_answer(oracle):
    g = torch.arange(6, dtype=torch.float32)
    pts = torch.stack(torch.meshgrid(g, g, g, indexing="ij"), dim=-1).reshape(-1, 3)
    out = dist2(oracle.api, pts)
    assert torch.all(out == 1.0)  # three axis neighbours at distance 1 everywhere (corners included)


def test_random_cloud_matches_brute_force(oracle):
    for P, seed in ((5, 0), (1000, 1), (5000, 2)):
        rng = np.random.RandomState(seed)
        pts = torch.from_numpy((rng.random_sample((P, 3)) * 2.6 - 1.3).astype(np.float32))
        out = dist2(oracle.api, pts)
        ref = brute(pts)
        assert torch.allclose(out.double(), ref, rtol=2e-6, atol=1e-12)


def test_duplicates_and_tiny_inputs(oracle):
    pts = torch.tensor([[0.5, 0.5, 0.5]] * 4 + [[1.0, 0.0, 0.0]])
    out = dist2(oracle.api, pts)
    assert torch.all(out[:4] == 0.0)  # coincident points count with distance 0
    # fewer than 4 points: missing neighbours stay at FLT_MAX in the reference -> huge mean
    out2 = dist2(oracle.api, torch.tensor([[0.0, 0, 0], [1.0, 0, 0]]))
    assert torch.all(out2 > 1e30)
    assert dist2(oracle.api, torch.zeros((0, 3))).numel() == 0


def test_fsgs_neighbour_indices_vs_brute_force(oracle):
    """FSGS's distCUDA2 also returns the three neighbours (nearest first): checked against an exhaustive float64 search
    on points without distance ties."""
    import numpy as np
    import torch
    from gsplat_amd.knn import dist2, dist2_with_indices
    rng = np.random.RandomState(12)
    for P in (4, 50, 3000):
        pts = torch.from_numpy(rng.uniform(-1.3, 1.3, (P, 3)).astype(np.float32))
        d, idx = dist2_with_indices(oracle.api, pts)
        assert idx.dtype == torch.int32 and idx.shape == (P, 3)
        assert torch.equal(d, dist2(oracle.api, pts))  # same distances as the index-free entry point, bit for bit
        D = torch.cdist(pts.double(), pts.double()) ** 2
        D.fill_diagonal_(float("inf"))
        want = torch.argsort(D, dim=1)[:, :3]
        assert torch.equal(idx.long(), want)
        assert torch.allclose(d.double(), torch.gather(D, 1, want).mean(dim=1), rtol=1e-5)
