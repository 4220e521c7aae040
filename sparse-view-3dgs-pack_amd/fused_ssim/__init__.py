"""Drop-in for the reference's optional `fused_ssim` package (fused-ssim/fused_ssim/__init__.py:34-41,
imported at LGDWT-GS/train.py:36-40)."""
import lgdwt_loss

allowed_padding = ["same", "valid"]


def fused_ssim(img1, img2, padding="same", train=True):
    assert padding in allowed_padding
    return lgdwt_loss.ops().fused_ssim(img1, img2, padding, train)
