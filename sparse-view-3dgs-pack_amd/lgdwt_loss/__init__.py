"""LGDWT-GS loss functions on the MI355X kernels - same names and signatures as the reference's
LGDWT-GS/utils/loss_utils.py (l1_loss, ssim, get_dwt_subbands, compute_elf_map,
compute_patch_dwt_loss) plus the fused forms used by the step loop.  No CPU fallback."""
from gsplat_amd._lib import hip_api
from gsplat_amd.losses import BANDS, LGDWTCriterion, LossOps  # noqa: F401

_ops = None


def ops():
    global _ops
    if _ops is None:
        _ops = LossOps(hip_api())
    return _ops


def l1_loss(network_output, gt):
    return ops().l1_loss(network_output, gt)


def ssim(img1, img2, window_size=11, size_average=True):
    return ops().ssim(img1, img2, window_size, size_average)


def get_dwt_subbands(x):
    return ops().get_dwt_subbands(x)


def dwt_l1_loss(pred, gt, weights):
    return ops().dwt_l1_loss(pred, gt, weights)


def compute_elf_map(image):
    return ops().compute_elf_map(image)


def compute_patch_dwt_loss(pred, gt, elf_map, patch_size=128, percentile=0.2, lh1_weight=1.0, hl1_weight=1.0):
    return ops().compute_patch_dwt_loss(pred, gt, elf_map, patch_size, percentile, lh1_weight, hl1_weight)


def criterion(**kw):
    return LGDWTCriterion(ops(), **kw)
