"""`render()` of LGDWT-GS/gaussian_renderer/__init__.py:18-128 for the reference's OWN GaussianModel - same signature, same
returned dict - that hands the library the model's RAW rows instead of activated copies.

The reference's render() evaluates `pc.get_scaling / get_rotation / get_opacity / get_features` (exp, F.normalize, sigmoid,
torch.cat: six torch kernels forward, a dozen backward, 0.5 ms of a 2.5 ms iteration at BASELINE C3) and gives the results to
the rasterizer.  Here the six leaf tensors of the model (`_xyz, _features_dc, _features_rest, _opacity, _scaling, _rotation`,
scene/gaussian_model.py:40-60) go through ONE autograd node: the forward kernels activate the rows as they load them
(GsGaussians.raw_activations), the backward is gs_backward_step in its gradients-out form, which writes the gradients with
respect to the RAW rows - the activation backward folded into the per-Gaussian kernel.  The SH coefficients are read where the
model keeps them - `_features_dc` [P,1,3] and `_features_rest` [P,15,3], GsGaussians.shs_rest - and their gradients are written
as two contiguous tensors (GsStepState.grad_out_rest): no torch.cat of 192 B per Gaussian forward (0.14 ms at C3), no copy of
a strided gradient slice by autograd's AccumulateGrad backward (0.09 ms).  What a maintainer changes:

    from gsplat_amd.render_raw import render          # instead of: from gaussian_renderer import render

Everything else of train.py stays: `viewspace_point_tensor.grad` is filled so that
`gaussians.add_densification_stats(viewspace_point_tensor, visibility_filter)` (train.py:268, gaussian_model.py:471-473)
computes the reference's statistic - the node returns (|dL/dmean2D|, 0, 0) per Gaussian, whose norm over the first two
columns IS |dL/dmean2D| (sqrt(x * x) = |x| exactly in binary floating point).

`separate_sh` (gaussian_renderer/__init__.py:82-100: `dc = pc.get_features_dc, shs = pc.get_features_rest` handed to the accelerated
rasterizer as two tensors; train.py passes SPARSE_ADAM_AVAILABLE for it) is accepted with either value: the two tensors are what
the kernels read here anyway.  Not served (the reference's python fall-backs; the plain render() of the drop-in packages covers
them): compute_cov3D_python, convert_SHs_python, override_color."""
import math

import torch

_EMPTY = None


def _empty():
    global _EMPTY
    if _EMPTY is None:
        _EMPTY = torch.Tensor([])
    return _EMPTY


class _RawRender(torch.autograd.Function):
    """(xyz, f_dc, f_rest, opacity, scaling, rotation: the model's leaves; means2D: the gradient carrier) -> (color, radii, invdepth)"""

    @staticmethod
    def forward(ctx, xyz, f_dc, f_rest, opacity, scaling, rotation, means2D, rs, backend, camera_key):
        if f_dc.shape[1:] != (1, 3) or f_rest.shape[1:] != (15, 3):
            raise NotImplementedError("render_raw serves the model of max_sh_degree = 3: _features_dc [P,1,3], _features_rest [P,15,3]")
        f_dc, f_rest = f_dc.contiguous(), f_rest.contiguous()     # (they are: gaussian_model.py:127-130 makes them so)
        if camera_key is not None:
            backend.camera_key = camera_key
            backend.camera_key_limits = True
        backend.raw_activations = True
        backend.sh_rest = f_rest
        e = _empty()
        out = backend.rasterize_gaussians(rs.bg, xyz, e, opacity, scaling, rotation, rs.scale_modifier, e, rs.viewmatrix,
                                          rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, f_dc,
                                          rs.sh_degree, rs.campos, rs.prefiltered, rs.antialiasing, rs.debug)
        num_rendered, color, radii, geom, binning, img, invdepth = out
        ctx.rs, ctx.backend, ctx.num_rendered = rs, backend, num_rendered
        ctx.save_for_backward(xyz, f_dc, f_rest, opacity, scaling, rotation, radii, geom, binning, img)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)
        return color, radii, invdepth

    @staticmethod
    def backward(ctx, g_color, _g_radii, g_depth):
        from .capi import GsStepState
        xyz, f_dc, f_rest, opacity, scaling, rotation, radii, geom, binning, img = ctx.saved_tensors
        rs, backend = ctx.rs, ctx.backend
        dev, P = xyz.device, int(xyz.shape[0])
        f32 = dict(dtype=torch.float32, device=dev)
        if P == 0:   # an empty model (rasterize_points.cu:88): nothing was rendered, every gradient is empty
            z = lambda *shape: torch.zeros(shape, **f32)
            return z(0, 3), z(0, 1, 3), z(0, 15, 3), z(0, 1), z(0, 3), z(0, 4), z(0, 3), None, None, None
        if g_color is None:
            g_color = torch.zeros((3, rs.image_height, rs.image_width), **f32)
        gx, gdc, grest = torch.empty((P, 3), **f32), torch.empty((P, 1, 3), **f32), torch.empty((P, 15, 3), **f32)
        gop, gsc, grot = torch.empty((P, 1), **f32), torch.empty((P, 3), **f32), torch.empty((P, 4), **f32)
        vsp = torch.zeros((P, 3), **f32)            # column 0 <- |dL/dmean2D| (the statistic's increment), columns 1, 2 stay 0
        stats = torch.zeros((3, P), **f32)          # rows: |dL/dmean2D|, visible ? 1 : 0, max_radii2D scratch
        st = GsStepState()
        st.xyz, st.features, st.opacity = xyz.data_ptr(), f_dc.data_ptr(), opacity.data_ptr()
        st.scaling, st.rotation = scaling.data_ptr(), rotation.data_ptr()
        for k, t in enumerate((gx, gdc, gop, gsc, grot)):
            st.grad_out[k] = t.data_ptr()
            st.step[k] = 1
        st.grad_out_rest = grest.data_ptr()
        st.beta1, st.beta2, st.eps = 0.9, 0.999, 1e-15
        st.xyz_gradient_accum, st.denom, st.max_radii2D = stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr()
        backend._raw_backward = True
        backend.sh_rest = f_rest
        backend.fused_step = st
        e = _empty()
        backend.rasterize_gaussians_backward(rs.bg, xyz, radii, e, opacity, scaling, rotation, rs.scale_modifier, e, rs.viewmatrix,
                                             rs.projmatrix, rs.tanfovx, rs.tanfovy, g_color, g_depth, f_dc, rs.sh_degree,
                                             rs.campos, geom, ctx.num_rendered, binning, img, rs.antialiasing, rs.debug)
        vsp[:, 0] = stats[0]
        return gx, gdc, grest, gop, gsc, grot, vsp, None, None, None


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, separate_sh=False, override_color=None,
           use_trained_exp=False, camera_key=None):
    """gaussian_renderer/__init__.py:18-128.  camera_key (optional, not in the reference): a stable identity of the camera - the
    backend then keeps its tile order and verified depth limits between visits."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, _RasterizeGaussians
    if override_color is not None or getattr(pipe, "compute_cov3D_python", False) or getattr(pipe, "convert_SHs_python", False):
        raise NotImplementedError("render_raw serves the rasterizer's own SH / covariance path only "
                                  "(use the drop-in GaussianRasterizer for the python fall-backs)")
    backend = getattr(_RasterizeGaussians._impl, "backend", None)
    if backend is None:
        raise RuntimeError("the raw-row render needs the HIP backend")
    xyz = pc._xyz
    screenspace_points = torch.empty_like(xyz).requires_grad_(True)   # never read: it only carries the gradient
    rs = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center,
        prefiltered=False, debug=bool(getattr(pipe, "debug", False)), antialiasing=bool(getattr(pipe, "antialiasing", False)))
    rendered_image, radii, depth_image = _RawRender.apply(xyz, pc._features_dc, pc._features_rest, pc._opacity, pc._scaling,
                                                          pc._rotation, screenspace_points, rs, backend, camera_key)
    if use_trained_exp:   # gaussian_renderer/__init__.py:112-115
        exposure = pc.get_exposure_from_name(viewpoint_camera.image_name)
        rendered_image = torch.matmul(rendered_image.permute(1, 2, 0), exposure[:3, :3]).permute(2, 0, 1) + exposure[:3, 3, None, None]
    # one key more than the reference's dict: the render BEFORE the clamp of gaussian_renderer/__init__.py:119, for
    # lgdwt_loss.criterion().fused_call(), which applies - and differentiates - the clamp inside its one node
    return {"render": rendered_image.clamp(0, 1), "render_unclamped": rendered_image, "viewspace_points": screenspace_points,
            "visibility_filter": (radii > 0).nonzero(), "radii": radii, "depth": depth_image}
