"""Stand-in for the compiled module `dgr_fsgs._C` (-confidence fork, ext.cpp): same positional arguments and
return tuples as its `rasterize_gaussians` (19 args -> num_rendered, color, depth, alpha, radii, 3 buffers) and
`rasterize_gaussians_backward` (24 args), served by libgsplat_hip.so through gs_*_fsgs."""
from gsplat_amd import hip_backend


def adapt_forward(backend, args):
    (bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
     tanfovx, tanfovy, image_height, image_width, sh, degree, campos, prefiltered, debug) = args
    num_rendered, color, radii, geom, binning, img, depth, alpha = backend.rasterize_gaussians(
        bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
        tanfovx, tanfovy, image_height, image_width, sh, degree, campos, prefiltered, False, debug, fsgs=True)
    return num_rendered, color, depth, alpha, radii, geom, binning, img


def adapt_backward(backend, args, opacities):
    (bg, means3D, radii, colors_precomp, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
     tanfovx, tanfovy, grad_color, grad_depth, grad_alpha, sh, degree, campos, geomBuffer, R, binningBuffer, imgBuffer,
     alpha, debug) = args
    return backend.rasterize_gaussians_backward(
        bg, means3D, radii, colors_precomp, opacities, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
        projmatrix, tanfovx, tanfovy, grad_color, grad_depth, sh, degree, campos, geomBuffer, R, binningBuffer,
        imgBuffer, False, debug, dL_dout_extra=grad_alpha, fsgs=True)


def rasterize_gaussians(*args):
    return adapt_forward(hip_backend(), args)


def rasterize_gaussians_backward(*args, opacities=None):
    return adapt_backward(hip_backend(), args, opacities)


def mark_visible(*args):
    return hip_backend().mark_visible(*args)
