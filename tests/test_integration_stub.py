"""INTEGRATION.md section 2 prints the ctypes binding a maintainer of the reference would add.  The block is executed here as it
stands in the document (only the library's path is filled in): its struct mirrors must have the layout the library was compiled
with (its own gs_struct_bytes assertions run at import), and on a GPU its `rasterize_gaussians` must return what the drop-in
package returns."""
import math
import os
import re
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_stub():
    from gsplat_amd._lib import LIB_PATH
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. The binding"):]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    assert 'C.CDLL("libgsplat_hip.so")' in code
    code = code.replace('C.CDLL("libgsplat_hip.so")', "C.CDLL(%r)" % LIB_PATH)
    mod = types.ModuleType("integration_stub")
    exec(compile(code, "INTEGRATION.md#2", "exec"), mod.__dict__)
    return mod


def test_the_documented_binding_has_the_library_s_struct_layout():
    import ctypes
    from gsplat_amd import capi
    stub = load_stub()       # (its own layout assertions have run)
    for name in ("GsView", "GsGaussians", "GsScratch"):
        a, b = getattr(stub, name), getattr(capi, name)
        assert [f[0] for f in a._fields_] == [f[0] for f in b._fields_], name
        assert ctypes.sizeof(a) == ctypes.sizeof(b)


@pytest.mark.gpu
def test_the_documented_binding_renders_what_the_package_renders(hip):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from gsplat_amd import synthetic
    from gsplat_amd.trainer import camera_to
    stub = load_stub()
    dev = torch.device("cuda")
    P, W, H = 5000, 200, 136
    sc = synthetic.trained_like(P, seed=5)
    cam = camera_to(synthetic.orbit_cameras(W, H)[3], dev)
    bg = torch.tensor([0.3, 0.2, 0.1], device=dev)
    means, sh, op, scl, rot = (sc[k].to(dev).contiguous() for k in ("means3D", "shs", "opacities", "scales", "rotations"))
    e = torch.Tensor([])
    tfx, tfy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    R, color, radii, geom, binning, img, invdepth = stub.rasterize_gaussians(
        bg, means, e, op, scl, rot, 1.0, e, cam.world_view_transform, cam.full_proj_transform, tfx, tfy, H, W, sh, 3,
        cam.camera_center, False, False, False)
    torch.cuda.synchronize()
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=tfx, tanfovy=tfy, bg=bg, scale_modifier=1.0,
                                       viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=3,
                                       campos=cam.camera_center, prefiltered=False, debug=False, antialiasing=False)
    with torch.no_grad():
        want_color, want_radii, want_depth = GaussianRasterizer(raster_settings=rs)(
            means3D=means, means2D=torch.zeros_like(means), shs=sh, colors_precomp=None, opacities=op, scales=scl,
            rotations=rot, cov3D_precomp=None)
    assert R > 0 and torch.equal(radii, want_radii)
    assert torch.equal(color, want_color) and torch.equal(invdepth, want_depth)
