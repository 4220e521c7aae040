"""Fused RGB + NIR rasterization (BASELINE config C5).

The reference renders the near-infrared image with a SECOND rasterizer pass per view
(mult-dwtgs/gaussian_renderer/__init__.py:151-258, `render_nir`): `colors_precomp` = the NIR albedo
replicated to three identical channels, `shs=None`, same geometry, and keeps channel 0.  Preprocess, scan,
duplicate, sort, tile ranges and every alpha evaluation of that pass repeat the RGB pass bit for bit.
Here the albedo rides along as a 4th blended channel of ONE pass (`gs_forward_render_x` /
`gs_backward_x`): nir[H,W] = sum_i nir_i alpha_i T_i + T_final bg[0].  Parity target = the reference's two
3-channel passes (tests/test_gpu_nir.py): images equal, geometry gradients equal the SUM of both passes'.

`GaussianRasterizerX` mirrors `GaussianRasterizer` with one more keyword (`extra`) and one more output;
`nir_colors()` restates the reference's albedo/gain shape handling.
"""
import torch
import torch.nn as nn

import diff_gaussian_rasterization as dgr


def nir_colors(nir_albedo, nir_gain=None):
    """[P] albedo the NIR pass blends: albedo * clamp(gain, 0.1, 10), first channel of whatever shape the
    model stores ((N,), (N,1), (N,1,1), (N,1,3), (N,3)) - render_nir:166-194 (the reference replicates it to
    three identical channels and keeps channel 0 of the image, so only channel 0 matters)."""
    nir = nir_albedo if nir_gain is None else nir_albedo * torch.clamp(nir_gain, 0.1, 10.0)
    if nir.dim() == 3:
        nir = nir.reshape(nir.shape[0], -1)
    if nir.dim() == 2:
        nir = nir[:, 0]
    return nir


class _RasterizeGaussiansX(torch.autograd.Function):
    _impl = dgr._C

    @classmethod
    def forward(cls, ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, extra,
                extra_gain, raster_settings):
        rs = raster_settings
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
                rs.sh_degree, rs.campos, rs.prefiltered, rs.antialiasing, rs.debug)
        kw = {} if extra_gain is None else {"extra_gain": extra_gain}
        num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, invdepths, extra_img = \
            cls._impl.rasterize_gaussians(*args, extra=extra, **kw)
        ctx.extra_gain = extra_gain
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities,
                              geomBuffer, binningBuffer, imgBuffer, extra)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)  # an unused output's gradient arrives as None (see diff_gaussian_rasterization)
        return color, radii, invdepths, extra_img

    @classmethod
    def backward(cls, ctx, grad_out_color, _, grad_out_depth, grad_out_extra):
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities, geomBuffer,
         binningBuffer, imgBuffer, extra) = ctx.saved_tensors
        if grad_out_color is None:
            grad_out_color = torch.zeros((3, rs.image_height, rs.image_width), dtype=torch.float32, device=means3D.device)
        args = (rs.bg, means3D, radii, colors_precomp, opacities, scales, rotations, rs.scale_modifier,
                cov3Ds_precomp, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color,
                grad_out_depth, sh, rs.sh_degree, rs.campos, geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer,
                rs.antialiasing, rs.debug)
        kw = {} if ctx.extra_gain is None else {"extra_gain": ctx.extra_gain}
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh,
         grad_scales, grad_rotations, grad_extra) = cls._impl.rasterize_gaussians_backward(
            *args, extra=extra, dL_dout_extra=grad_out_extra, **kw)
        # (the fused train step - RasterBackend.fused_step - has applied every gradient itself and returns none)
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales,
                grad_rotations, grad_cov3Ds_precomp, None if grad_extra is None else grad_extra.reshape(extra.shape), None,
                None)


class GaussianRasterizerX(nn.Module):
    """`GaussianRasterizer` + `extra` [P]: returns (color, radii, invdepth, extra_img[1,H,W])."""
    _fn = _RasterizeGaussiansX
    camera_key = None   # as GaussianRasterizer.camera_key: the caller's name for this camera (per-camera tile order / depth limits)
    camera_limits = True

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, extra, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, extra_gain=None):
        """extra_gain (with the backend's raw activations only): `extra` is then the RAW per-Gaussian row and the kernels
        blend sigmoid(extra) * clamp(extra_gain, 0.1, 10) - the fused multispectral train step's form."""
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        if extra is None or extra.numel() != means3D.shape[0]:
            raise Exception('extra must hold one value per Gaussian')
        e = torch.Tensor([])
        if self.camera_key is not None:
            backend = getattr(self._fn._impl, "backend", None)
            if backend is not None:
                backend.camera_key = self.camera_key   # one-shot: consumed by the forward below
                backend.camera_key_limits = bool(self.camera_limits)
        return self._fn.apply(means3D, means2D, e if shs is None else shs, e if colors_precomp is None else colors_precomp,
                              opacities, e if scales is None else scales, e if rotations is None else rotations,
                              e if cov3D_precomp is None else cov3D_precomp, extra.reshape(-1), extra_gain,
                              self.raster_settings)
