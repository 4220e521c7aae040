"""Independent torch restatement of the Haar analysis step the LGDWT-GS losses stand on (test-side, float64 capable).

The reference takes its DWT from the third-party `pytorch_wavelets` (un-vendored, un-versioned, absent here).  This file
restates that package's PUBLISHED algorithm for DWTForward(J, 'db1', 'symmetric') with F.conv2d: afb1d = grouped stride-2
correlation along W, then along H, filters h0=[c,c], h1=[c,-c], one 'symmetric' pad sample on the right / bottom for odd
sizes, band order LH / HL / HH.  It is used two ways:
  * `DWTForward` below is registered as `pytorch_wavelets` in sys.modules by tests/golden/make_golden.py so that the
    reference's own LGDWT-GS/utils/loss_utils.py imports and RUNS in the build container; what it returns for
    get_dwt_subbands / compute_elf_map / compute_patch_dwt_loss (+ autograd) is committed as tests/golden/lgdwt_loss.npz;
  * `haar_bands_2level` is the independent check of the oracle's / HIP kernels' bands at odd and tiny sizes.
Nothing of the reference's loss code is restated here: D2-D4 are pinned by that fixture (everything except the 2x2 Haar
signs, which no consumer can see: L1 of differences, abs, squares)."""
import torch
import torch.nn.functional as F

C = 0.7071067811865476


def afb1d(x, dim):
    """x [N,C,H,W] -> (lo, hi) along `dim` (2 = H, 3 = W)."""
    Cn = x.shape[1]
    N = x.shape[dim]
    if N % 2 == 1:
        pad = (0, 1, 0, 0) if dim == 3 else (0, 0, 0, 1)
        x = F.pad(x, pad, mode="replicate")  # one symmetric sample == repeat of the last one
    h0 = torch.tensor([C, C], dtype=x.dtype)
    h1 = torch.tensor([C, -C], dtype=x.dtype)
    shape = [1, 1, 1, 1]
    shape[dim] = 2
    h = torch.cat([h0.reshape(shape), h1.reshape(shape)] * Cn, dim=0)
    stride = (2, 1) if dim == 2 else (1, 2)
    y = F.conv2d(x, h, stride=stride, groups=Cn)
    return y[:, 0::2], y[:, 1::2]


def dwt1(x):
    lo, hi = afb1d(x, 3)
    ll, lh = afb1d(lo, 2)
    hl, hh = afb1d(hi, 2)
    return ll, lh, hl, hh


def haar_bands_2level(x):
    """{LL1 .. HH2}: one Haar level of x, one more of its LL band."""
    LL1, LH1, HL1, HH1 = dwt1(x)
    LL2, LH2, HL2, HH2 = dwt1(LL1)
    return {"LL1": LL1, "LH1": LH1, "HL1": HL1, "HH1": HH1, "LL2": LL2, "LH2": LH2, "HL2": HL2, "HH2": HH2}


def cotangent(shape, salt=0):
    """A fixed, seed-free cotangent tensor in [-1, 1] (shared by the fixture generator and the tests, so that it need
    not be stored): a sign-alternating pattern of the flat index."""
    n = 1
    for d in shape:
        n *= int(d)
    i = torch.arange(n, dtype=torch.int64)
    v = ((i * 7 + (i // 13) * 5 + salt * 3) % 11 - 5).to(torch.float32) / 5.0
    return v.reshape(tuple(shape))


class DWTForward(torch.nn.Module):
    """Stand-in for pytorch_wavelets.DWTForward(J, mode='symmetric', wave='db1'): forward(x[N,C,H,W]) ->
    (Yl, [Yh_1 .. Yh_J]) with Yh_j [N,C,3,h_j,w_j] = (LH, HL, HH) of level j, finest first."""

    def __init__(self, J=1, mode="symmetric", wave="db1"):
        super().__init__()
        if mode != "symmetric" or wave not in ("db1", "haar"):
            raise NotImplementedError("only the Haar / symmetric transform the LGDWT-GS losses use")
        self.J = int(J)

    def forward(self, x):
        yh = []
        ll = x
        for _ in range(self.J):
            ll, lh, hl, hh = dwt1(ll)
            yh.append(torch.stack((lh, hl, hh), dim=2))
        return ll, yh
