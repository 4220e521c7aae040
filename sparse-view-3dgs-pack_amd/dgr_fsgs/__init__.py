"""Drop-in for `dgr_fsgs`, the rasterizer generation FSGS trains with (FSGS/gaussian_renderer/__init__.py:14;
FSGS/submodules/diff-gaussian-rasterization-confidence/dgr_fsgs/__init__.py), backed by the same HIP core:

  GaussianRasterizationSettings   NamedTuple, 13 fields ending in `confidence`   dgr_fsgs/__init__.py:169-182
  GaussianRasterizer.forward      -> (color, radii, depth, alpha)                 dgr_fsgs/__init__.py:184-229
  _RasterizeGaussians.backward    every parameter gradient except means2D is scaled by `confidence`  :147-157

Differences to the `dr_aa` generation served by `diff_gaussian_rasterization`: no anti-aliasing flag, no
inverse-depth output; `depth` = sum depth alpha T (view-space z) and `alpha` = sum alpha T are differentiable
outputs; the backward receives their image gradients (gs_forward_render_fsgs / gs_backward_fsgs).
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class _RasterizeGaussians(torch.autograd.Function):
    _impl = _C  # swapped by the test-suite to run the same plumbing on another implementation of the C ABI

    @classmethod
    def forward(cls, ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        rs = raster_settings
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
                rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
        num_rendered, color, depth, alpha, radii, geomBuffer, binningBuffer, imgBuffer = \
            cls._impl.rasterize_gaussians(*args)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        # (the reference does not keep `opacities`: its backward never reads them; this ABI validates the pointer)
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer, alpha, opacities)
        ctx.set_materialize_grads(False)
        return color, radii, depth, alpha

    @classmethod
    def backward(cls, ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer,
         alpha, opacities) = ctx.saved_tensors
        if grad_color is None:
            grad_color = torch.zeros((3, rs.image_height, rs.image_width), dtype=torch.float32, device=means3D.device)
        args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_color, grad_depth, grad_alpha, sh,
                rs.sh_degree, rs.campos, geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, alpha, rs.debug)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = cls._impl.rasterize_gaussians_backward(*args, opacities=opacities)
        c = rs.confidence

        def scaled(g, w):  # gradients of absent inputs (None) stay None
            return None if g is None or g.numel() == 0 else g * w
        return (scaled(grad_means3D, c), grad_means2D, scaled(grad_sh, c[..., None]), scaled(grad_colors_precomp, c),
                scaled(grad_opacities, c), scaled(grad_scales, c), scaled(grad_rotations, c),
                scaled(grad_cov3Ds_precomp, c), None)


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
    confidence: torch.Tensor


class GaussianRasterizer(nn.Module):
    _fn = _RasterizeGaussians

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return self._fn._impl.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        e = torch.Tensor([])
        return self._fn.apply(means3D, means2D, e if shs is None else shs, e if colors_precomp is None else colors_precomp,
                              opacities, e if scales is None else scales, e if rotations is None else rotations,
                              e if cov3D_precomp is None else cov3D_precomp, self.raster_settings)
