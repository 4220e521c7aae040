"""HIP vs CPU-oracle comparison of one full-size view (forward state, image, every gradient tensor and the
intermediate per-Gaussian sums of the blend backward).  Shared by tests/test_gpu_fullsize.py and
tests/tools/c3_grad_probe.py.  Test infrastructure: drives both backends through the same glue
(gsplat_amd.raster.RasterBackend) the drop-in packages use.

What the blend backward accumulates per Gaussian (backward.cu:593-635, one float atomicAdd per pixel and slot in the
reference, in a run-dependent order) is kept by the oracle in double and rounded once; the product sums in fp32
registers per tile and adds the tile totals with float atomics.  `compare_view` reports, per tensor,
max|hip - oracle| / max|oracle| and rms(hip - oracle) / rms(oracle).
"""
import torch

GRAD_NAMES = ("means2D", "colors_precomp", "opacities", "means3D", "cov3D_precomp", "shs", "scales", "rotations")
ROW_SLOTS = ("mean2D.x", "mean2D.y", "conic.xx", "conic.xy", "conic.yy", "opacity", "r", "g", "b", "invdepth")


def _args(scene, cam, device, bg, antialiasing=False):
    def dev(t):
        return torch.empty(0) if t is None else t.to(device)
    return dict(bg=bg.to(device), means3D=dev(scene["means3D"]), colors=dev(scene.get("colors_precomp")),
                opacities=dev(scene["opacities"]), scales=dev(scene.get("scales")), rotations=dev(scene.get("rotations")),
                mod=scene.get("scale_modifier", 1.0), cov=dev(scene.get("cov3D_precomp")),
                view=cam.world_view_transform.to(device), proj=cam.full_proj_transform.to(device), tx=cam.tanfovx,
                ty=cam.tanfovy, H=cam.image_height, W=cam.image_width, sh=dev(scene.get("shs")),
                deg=scene.get("sh_degree", 0), campos=cam.camera_center.to(device), aa=antialiasing)


def forward(backend, scene, cam, device, bg, antialiasing=False):
    a = _args(scene, cam, device, bg, antialiasing)
    R, color, radii, geom, binning, img, invd = backend.rasterize_gaussians(
        a["bg"], a["means3D"], a["colors"], a["opacities"], a["scales"], a["rotations"], a["mod"], a["cov"], a["view"],
        a["proj"], a["tx"], a["ty"], a["H"], a["W"], a["sh"], a["deg"], a["campos"], False, a["aa"], False)
    return dict(args=a, R=R, color=color, radii=radii, geom=geom, binning=binning, img=img, invdepth=invd)


def backward(backend, fw, dL_dcolor, dL_dinvdepth=None):
    """-> (dict of the eight gradient tensors, [P,16] rows of the blend backward)"""
    a = fw["args"]
    dev = a["means3D"].device
    backend.keep_workspace = True
    try:
        out = backend.rasterize_gaussians_backward(
            a["bg"], a["means3D"], fw["radii"], a["colors"], a["opacities"], a["scales"], a["rotations"], a["mod"],
            a["cov"], a["view"], a["proj"], a["tx"], a["ty"], dL_dcolor.to(dev),
            None if dL_dinvdepth is None else dL_dinvdepth.to(dev), a["sh"], a["deg"], a["campos"], fw["geom"],
            fw["R"], fw["binning"], fw["img"], a["aa"], False)
        if dev.type == "cuda":
            torch.cuda.synchronize()
        ws = backend.last_workspace
    finally:
        backend.keep_workspace = False
        backend.last_workspace = None
    P = a["means3D"].shape[0]
    rows = ws[: P * 64].view(torch.float32).reshape(P, 16).cpu().clone()
    grads = {n: (None if t is None else t.detach().cpu()) for n, t in zip(GRAD_NAMES, out)}
    return grads, rows


def err_stats(h, o):
    h, o = h.double().flatten(), o.double().flatten()
    d = (h - o).abs()
    omax = max(float(o.abs().max()), 1e-30)
    orms = max(float(o.pow(2).mean().sqrt()), 1e-30)
    i = int(d.argmax())
    return dict(max_rel=float(d.max()) / omax, rms_rel=float(d.pow(2).mean().sqrt()) / orms, ref_max=omax, argmax=i)


def compare_grads(hg, og):
    return {k: err_stats(hg[k], og[k]) for k in og if og[k] is not None and hg.get(k) is not None}


def compare_rows(hr, orow, has_invdepth=False):
    out = {}
    for s, name in enumerate(ROW_SLOTS):
        if name == "invdepth" and not has_invdepth:
            continue
        out[name] = err_stats(hr[:, s], orow[:, s])
    return out
