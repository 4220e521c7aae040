"""Independent torch restatement of the LGDWT-GS loss terms (test-side reference, float64 capable).

DWT: the published algorithm of pytorch_wavelets' DWTForward(J=1,'db1','symmetric') written with
F.conv2d exactly as that package does it (afb1d: grouped stride-2 correlation along W, then along H,
filters h0=[c,c], h1=[c,-c], one 'symmetric' pad sample on the right/bottom for odd sizes), and the loss
functions of LGDWT-GS/utils/loss_utils.py:106-153,336-442 / train.py:132-164 re-typed on top of it."""
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


def get_dwt_subbands(x):
    LL1, LH1, HL1, HH1 = dwt1(x)
    LL2, LH2, HL2, HH2 = dwt1(LL1)
    return {"LL1": LL1, "LH1": LH1, "HL1": HL1, "HH1": HH1, "LL2": LL2, "LH2": LH2, "HL2": HL2, "HH2": HH2}


def l1_loss(a, b):
    return torch.abs(a - b).mean()


def dwt_loss(pred, gt, weights):
    pb, gb = get_dwt_subbands(pred), get_dwt_subbands(gt)
    total = 0.0
    for w, k in zip(weights, ("LL1", "LH1", "HL1", "HH1", "LL2", "LH2", "HL2", "HH2")):
        if w != 0.0:
            total = total + w * l1_loss(pb[k], gb[k])
    return total


def compute_elf_map(image):
    bands = get_dwt_subbands(image)

    def l1(x):
        return torch.sum(torch.abs(x), dim=1, keepdim=True)
    LL, LH, HL, HH = l1(bands["LL1"]), l1(bands["LH1"]), l1(bands["HL1"]), l1(bands["HH1"])
    HF = LH + HL + HH
    elf_low = LL / (LL + HF + 1e-8)
    H, W = image.shape[-2:]
    return F.interpolate(elf_low, size=(H, W), mode="bilinear", align_corners=False), elf_low


def compute_patch_dwt_loss(pred, gt, elf_map, patch_size=128, percentile=0.2, lh1_weight=1.0, hl1_weight=1.0):
    N, Cn, H, W = pred.shape
    if H < patch_size or W < patch_size:
        return torch.tensor(0.0)
    pred_patches = F.unfold(pred, kernel_size=patch_size, stride=patch_size)
    gt_patches = F.unfold(gt, kernel_size=patch_size, stride=patch_size)
    elf_patches = F.unfold(elf_map, kernel_size=patch_size, stride=patch_size)
    L = pred_patches.shape[2]
    patch_elf_means = elf_patches.mean(dim=1)
    all_elf_means = patch_elf_means.view(-1)
    k = int(all_elf_means.numel() * (1.0 - percentile))
    k = max(1, k)
    k = min(k, all_elf_means.numel())
    threshold, _ = torch.kthvalue(all_elf_means, k)
    mask = patch_elf_means >= threshold
    pred_patches = pred_patches.view(N, Cn, patch_size, patch_size, L).permute(0, 4, 1, 2, 3)
    gt_patches = gt_patches.view(N, Cn, patch_size, patch_size, L).permute(0, 4, 1, 2, 3)
    pred_sel, gt_sel = pred_patches[mask], gt_patches[mask]
    pb, gb = get_dwt_subbands(pred_sel), get_dwt_subbands(gt_sel)
    loss_LH, loss_HL, loss_HH = l1_loss(pb["LH1"], gb["LH1"]), l1_loss(pb["HL1"], gb["HL1"]), l1_loss(pb["HH1"], gb["HH1"])
    return (lh1_weight * loss_LH) + (hl1_weight * loss_HL) + (0.5 * (lh1_weight + hl1_weight) * loss_HH), mask
