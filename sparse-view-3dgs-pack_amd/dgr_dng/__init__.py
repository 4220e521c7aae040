"""Drop-in for `dgr_dng`, DNGaussian's rasterizer package (DNGaussian/gaussian_renderer/__init__.py:14;
DNGaussian/submodules/diff-gaussian-rasterization/dgr_dng/__init__.py).  Its CUDA sources are byte-identical to the
FSGS fork's (depth + alpha outputs and their gradients); the Python wrapper has no `confidence` field and does not
rescale gradients.  Served by the same entry points as `dgr_fsgs` (gs_forward_render_fsgs / gs_backward_fsgs)."""
from typing import NamedTuple

import torch
import torch.nn as nn

import dgr_fsgs
from dgr_fsgs import _C  # noqa: F401


class _RasterizeGaussians(dgr_fsgs._RasterizeGaussians):
    @classmethod
    def backward(cls, ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        rs = ctx.raster_settings
        ctx.raster_settings = _WithConfidence(rs)
        try:
            return super().backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha)
        finally:
            ctx.raster_settings = rs


class _WithConfidence:
    """View of the 12-field settings with the neutral confidence the shared backward multiplies by."""

    def __init__(self, rs):
        self._rs = rs
        self.confidence = torch.ones((1, 1), device=rs.bg.device)

    def __getattr__(self, name):
        return getattr(self._rs, name)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class GaussianRasterizer(dgr_fsgs.GaussianRasterizer):
    _fn = _RasterizeGaussians
