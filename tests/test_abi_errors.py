"""Error conventions of the C ABI (include/gsplat.h: 0 ok, < 0 caller error, never a throw across the boundary): the
argument checks return before anything is launched, so they run on the CPU checker here and on the HIP library in the
GPU suite (same table)."""
import ctypes as C

import pytest
import torch

from gsplat_amd.capi import GsAdamSeg, GsGaussians, GsGrads, GsScratch, GsView

E_NULL, E_SHAPE, E_SCRATCH, E_UNSUPPORTED = -1, -2, -3, -5


def setup(api, device, P=8, W=32, H=32):
    t = dict(bg=torch.zeros(3), view=torch.eye(4), proj=torch.eye(4), campos=torch.zeros(3), means=torch.rand(P, 3),
             op=torch.rand(P), sh=torch.rand(P, 16, 3), scales=torch.rand(P, 3), rot=torch.rand(P, 4))
    t = {k: v.to(device).contiguous() for k, v in t.items()}
    v = GsView(H, W, 0.5, 0.5, 1.0, 3, 0, 0, 0, 0, t["bg"].data_ptr(), t["view"].data_ptr(), t["proj"].data_ptr(),
               t["campos"].data_ptr())
    g = GsGaussians(P, 16, t["means"].data_ptr(), t["sh"].data_ptr(), None, t["op"].data_ptr(), t["scales"].data_ptr(),
                    t["rot"].data_ptr(), None, None)
    sizes = (C.c_size_t * 3)()
    ws = C.c_size_t(0)
    assert api.raw("scratch_bytes")(P, W, H, 64, sizes, C.byref(ws)) == 0
    bufs = [torch.zeros(int(n), dtype=torch.uint8, device=device) for n in sizes]
    s = GsScratch(bufs[0].data_ptr(), bufs[0].numel(), bufs[1].data_ptr(), bufs[1].numel(), bufs[2].data_ptr(),
                  bufs[2].numel(), 64)
    return t, v, g, s, bufs, int(ws.value)


def run_table(api, device):
    fg, fr, bw = api.raw("forward_geometry"), api.raw("forward_render"), api.raw("backward")
    t, v, g, s, bufs, wsb = setup(api, device)
    radii = torch.zeros(8, dtype=torch.int32, device=device)
    color = torch.zeros(3, 32, 32, device=device)
    st = None
    # scratch_bytes
    sizes = (C.c_size_t * 3)()
    assert api.raw("scratch_bytes")(8, 32, 32, 0, None, None) == E_NULL
    assert api.raw("scratch_bytes")(-1, 32, 32, 0, sizes, None) == E_SHAPE
    assert api.raw("scratch_bytes")(8, 0, 32, 0, sizes, None) == E_SHAPE
    # NULL structs / mandatory pointers
    assert fg(None, C.byref(g), C.byref(s), radii.data_ptr(), None, st) == E_NULL
    assert fg(C.byref(v), None, C.byref(s), radii.data_ptr(), None, st) == E_NULL
    assert fg(C.byref(v), C.byref(g), None, radii.data_ptr(), None, st) == E_NULL
    assert fg(C.byref(v), C.byref(g), C.byref(s), None, None, st) == E_NULL
    g2 = GsGaussians.from_buffer_copy(g)
    g2.means3D = None
    assert fg(C.byref(v), C.byref(g2), C.byref(s), radii.data_ptr(), None, st) == E_NULL
    # exactly one of {shs, colors_precomp}, {scales+rotations, cov3D_precomp} (dgr_3dgs/__init__.py:178-182)
    g3 = GsGaussians.from_buffer_copy(g)
    g3.colors_precomp = t["means"].data_ptr()
    assert fg(C.byref(v), C.byref(g3), C.byref(s), radii.data_ptr(), None, st) == E_SHAPE
    g4 = GsGaussians.from_buffer_copy(g)
    g4.shs = None
    assert fg(C.byref(v), C.byref(g4), C.byref(s), radii.data_ptr(), None, st) == E_SHAPE
    g5 = GsGaussians.from_buffer_copy(g)
    g5.rotations = None
    assert fg(C.byref(v), C.byref(g5), C.byref(s), radii.data_ptr(), None, st) == E_SHAPE
    g6 = GsGaussians.from_buffer_copy(g)
    g6.cov3D_precomp = t["means"].data_ptr()
    assert fg(C.byref(v), C.byref(g6), C.byref(s), radii.data_ptr(), None, st) == E_SHAPE
    g7 = GsGaussians.from_buffer_copy(g)
    g7.M = 4  # fewer coefficients than the active degree needs
    assert fg(C.byref(v), C.byref(g7), C.byref(s), radii.data_ptr(), None, st) == E_SHAPE
    g8 = GsGaussians.from_buffer_copy(g)
    g8.P = -3
    assert fg(C.byref(v), C.byref(g8), C.byref(s), radii.data_ptr(), None, st) == E_SHAPE
    v2 = GsView.from_buffer_copy(v)
    v2.image_width = 0
    assert fg(C.byref(v2), C.byref(g), C.byref(s), radii.data_ptr(), None, st) == E_SHAPE
    # scratch too small
    s2 = GsScratch.from_buffer_copy(s)
    s2.geom_bytes = 16
    assert fg(C.byref(v), C.byref(g), C.byref(s2), radii.data_ptr(), None, st) == E_SCRATCH
    s3 = GsScratch.from_buffer_copy(s)
    s3.img_bytes = 16
    assert fr(C.byref(v), C.byref(g), C.byref(s3), color.data_ptr(), None, st) == E_SCRATCH
    assert fr(C.byref(v), C.byref(g), C.byref(s), None, None, st) == E_NULL
    # backward
    grads = GsGrads()
    ws = torch.zeros(wsb, dtype=torch.uint8, device=device)
    assert bw(C.byref(v), C.byref(g), radii.data_ptr(), C.byref(s), 0, None, None, C.byref(grads), ws.data_ptr(), wsb, st) == E_NULL
    assert bw(C.byref(v), C.byref(g), radii.data_ptr(), C.byref(s), 0, color.data_ptr(), None, None, ws.data_ptr(), wsb, st) == E_NULL
    # Adam
    seg = (GsAdamSeg * 1)()
    buf = torch.zeros(16, device=device)
    assert api.raw("adam_step")(None, buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), 16, seg, 1, 0.9, 0.999, 1e-15, 1, st) == E_NULL
    assert api.raw("adam_step")(buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), 16, seg, 9, 0.9, 0.999, 1e-15, 1, st) == E_SHAPE
    assert api.raw("adam_step")(buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), 16, seg, 1, 0.9, 0.999, 1e-15, 0, st) == E_SHAPE
    # kNN
    assert api.raw("knn_mean_dist2")(None, 8, buf.data_ptr(), buf.data_ptr(), 64, st) == E_NULL
    assert api.raw("knn_mean_dist2")(t["means"].data_ptr(), -1, buf.data_ptr(), buf.data_ptr(), 64, st) == E_SHAPE
    assert api.raw("knn_mean_dist2")(t["means"].data_ptr(), 0, buf.data_ptr(), buf.data_ptr(), 64, st) == 0
    # fused-criterion entry points
    img = torch.rand(3, 32, 32, device=device)
    nine = torch.zeros(16, device=device)
    from gsplat_amd.capi import GsLgdwtParams
    lp = GsLgdwtParams()
    lp.n_pix, lp.n_band1, lp.n_band2 = 3072.0, 768.0, 192.0
    f = api.raw
    assert f("l1_dwt2_fwd")(img.data_ptr(), img.data_ptr(), 3, 32, 32, None, nine.data_ptr(), st) == E_NULL
    assert f("l1_dwt2_fwd")(img.data_ptr(), img.data_ptr(), 3, 0, 32, nine.data_ptr(), nine[1:].data_ptr(), st) == E_SHAPE
    assert f("l1_dwt2_bwd")(img.data_ptr(), img.data_ptr(), 3, 32, 32, None, nine.data_ptr(), img.data_ptr(), 0, st) == E_NULL
    assert f("l1_dwt2_bwd")(img.data_ptr(), img.data_ptr(), 0, 32, 32, nine.data_ptr(), nine.data_ptr(), img.data_ptr(), 0, st) == E_SHAPE
    assert f("ssim_partials_count")(1, 3, 32, 32) == 3 and f("ssim_partials_count")(1, 3, 33, 65) == 3 * 2 * 3
    assert f("ssim_partials_count")(0, 3, 32, 32) == 0
    assert f("ssim_fwd_partials")(img.data_ptr(), img.data_ptr(), 1, 3, 32, 32, 1e-4, 9e-4, None, None, None, None, st) == E_NULL
    assert f("lgdwt_combine_p")(None, None, 0, nine.data_ptr(), C.byref(lp), nine.data_ptr(), st) == E_NULL
    assert f("lgdwt_combine_p")(nine.data_ptr(), None, 5, nine.data_ptr(), C.byref(lp), nine.data_ptr(), st) == E_SHAPE
    assert f("lgdwt_combine_p")(nine.data_ptr(), None, -1, nine.data_ptr(), C.byref(lp), nine.data_ptr(), st) == E_SHAPE
    # round-5 entries: depth regularisation, masked Adam, the sparse exchange's pack / unpack, the layout query
    one = torch.rand(1, 32, 32, device=device)
    part = torch.zeros(int(f("depth_l1_partials_count")(1024)), device=device)
    assert f("depth_l1_partials_count")(0) == 0 and part.numel() >= 1
    assert f("depth_l1")(None, one.data_ptr(), None, 1024, part.data_ptr(), C.c_float(1.0), None, None, st) == E_NULL
    assert f("depth_l1")(one.data_ptr(), None, None, 1024, part.data_ptr(), C.c_float(1.0), None, None, st) == E_NULL
    assert f("depth_l1")(one.data_ptr(), one.data_ptr(), None, 1024, None, C.c_float(1.0), None, None, st) == E_NULL   # no output asked for
    assert f("depth_l1")(one.data_ptr(), one.data_ptr(), None, 0, part.data_ptr(), C.c_float(1.0), None, None, st) == 0
    am = f("adam_step_masked")
    assert am(None, buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), 16, seg, 1, 0.9, 0.999, 1e-15, 1, None, buf.data_ptr(), st) == E_NULL
    assert am(buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), 16, seg, 9, 0.9, 0.999, 1e-15, 1, None, buf.data_ptr(), st) == E_SHAPE
    assert am(buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), 16, seg, 1, 0.9, 0.999, 1e-15, 0, None, buf.data_ptr(), st) == E_SHAPE
    widths = (C.c_int32 * 2)(3, 1)
    mask = torch.ones(4, dtype=torch.uint8, device=device)
    pos = torch.arange(4, dtype=torch.int32, device=device)
    for name in ("rows_pack", "rows_unpack"):
        assert f(name)(buf.data_ptr(), -1, 2, widths, mask.data_ptr(), pos.data_ptr(), 4, buf.data_ptr(), st) == E_SHAPE
        assert f(name)(buf.data_ptr(), 4, 9, widths, mask.data_ptr(), pos.data_ptr(), 4, buf.data_ptr(), st) == E_SHAPE
        assert f(name)(buf.data_ptr(), 4, 2, widths, None, pos.data_ptr(), 4, buf.data_ptr(), st) == E_NULL
        assert f(name)(buf.data_ptr(), 4, 2, widths, mask.data_ptr(), pos.data_ptr(), 0, buf.data_ptr(), st) == 0      # empty union
    assert f("struct_bytes")(1) == C.sizeof(GsGaussians) and f("struct_bytes")(-1) == 0 and f("struct_bytes")(7) == 0
    return t, v, g, s, bufs, wsb


def test_oracle_argument_errors(oracle):
    run_table(oracle.api, torch.device("cpu"))


@pytest.mark.gpu
def test_hip_argument_errors(hip):
    api = hip.api
    t, v, g, s, bufs, wsb = run_table(api, torch.device("cuda"))
    dev = torch.device("cuda")
    # device-library specifics: the FSGS generation has no anti-aliasing; a 4th channel needs its input and output
    v2 = GsView.from_buffer_copy(v)
    v2.antialiasing = 1
    img = torch.zeros(3, 32, 32, device=dev)
    one = torch.zeros(1, 32, 32, device=dev)
    assert api.raw("forward_render_fsgs")(C.byref(v2), C.byref(g), C.byref(s), img.data_ptr(), one.data_ptr(), one.data_ptr(), None) == E_UNSUPPORTED
    assert api.raw("forward_render_fsgs")(C.byref(v), C.byref(g), C.byref(s), img.data_ptr(), None, one.data_ptr(), None) == E_NULL
    assert api.raw("forward_render_x")(C.byref(v), C.byref(g), C.byref(s), img.data_ptr(), None, one.data_ptr(), None) == E_NULL
    # the model's split SH rows (GsGaussians.shs_rest): DC rows + at least one more coefficient; not for the plain backward
    radii = torch.zeros(8, dtype=torch.int32, device=dev)
    gs_ = GsGaussians.from_buffer_copy(g)
    gs_.shs_rest = t["sh"].data_ptr()
    gs_.M = 1
    assert api.raw("forward_geometry")(C.byref(v), C.byref(gs_), C.byref(s), radii.data_ptr(), None, None) == E_SHAPE
    gs_.M = 16
    grads = GsGrads()
    ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
    assert api.raw("backward")(C.byref(v), C.byref(gs_), radii.data_ptr(), C.byref(s), 0, img.data_ptr(), None, C.byref(grads),
                               ws.data_ptr(), wsb, None) == E_UNSUPPORTED
    torch.cuda.synchronize()
