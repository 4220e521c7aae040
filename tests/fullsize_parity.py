"""HIP vs CPU-oracle comparison of one full-size view (forward state, image, every gradient tensor and the
intermediate per-Gaussian sums of the blend backward).  Shared by tests/test_gpu_fullsize.py and
tests/tools/c3_grad_probe.py.  Test infrastructure: drives both backends through the same glue
(gsplat_amd.raster.RasterBackend) the drop-in packages use.

What the blend backward accumulates per Gaussian (backward.cu:593-635, one float atomicAdd per pixel and slot in the
reference, in a run-dependent order) is kept by the oracle in double (and rounded once where its stage 2 reads it); the
product sums in fp32 registers per tile and adds the tile totals into float64 rows.  `compare_view` reports, per tensor,
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
    """-> (dict of the eight gradient tensors, [P,16] float64 rows of the blend backward)"""
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
    rows = ws[: P * 128].view(torch.float64).reshape(P, 16).cpu().clone()  # float64 sums (include/gsplat.h, gs_backward_from_rows)
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


def exact_scale_rot_chain(scene, cam, rows, radii):
    """Float64 image of the conic sums under stage 2: (dL_dscales, dL_drotations) = d/d(scales, quaternions) of
    sum_i <rows_i[conic], conic_i(scales_i, q_i)>, by autograd over the closed-form EWA projection (non-anti-aliased
    path; the reference's 1/(det^2 + 1e-7) regulariser, backward.cu:252, is applied to the upstream).  Stage 2 is
    LINEAR in the rows, so  chain(rows_a) - chain(rows_b) = chain(rows_a - rows_b): what a difference of the blend sums
    becomes in the scale / rotation gradients when nothing else goes wrong."""
    from dense_reference import quat_to_rot
    f64 = torch.float64
    vis = radii.cpu() > 0
    means = scene["means3D"].to(f64)
    s = scene["scales"].to(f64).clone().requires_grad_(True)
    q = scene["rotations"].to(f64).clone().requires_grad_(True)
    mod = float(scene.get("scale_modifier", 1.0))
    V = cam.world_view_transform.to(f64).cpu()
    H, W = cam.image_height, cam.image_width
    fx, fy = W / (2 * cam.tanfovx), H / (2 * cam.tanfovy)
    P = means.shape[0]
    p_view = torch.cat([means, torch.ones((P, 1), dtype=f64)], dim=1) @ V
    tz = p_view[:, 2]
    tz = torch.where(vis, tz, torch.ones_like(tz))  # culled rows carry zero sums; keep their arithmetic finite
    limx, limy = 1.3 * cam.tanfovx, 1.3 * cam.tanfovy
    tx = (p_view[:, 0] / tz).clamp(-limx, limx) * tz
    ty = (p_view[:, 1] / tz).clamp(-limy, limy) * tz
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], dim=1).reshape(P, 2, 3)
    M = J @ V[:3, :3].transpose(0, 1)
    R = quat_to_rot(q)
    S = torch.diag_embed(mod * s)
    Sigma = R @ S @ S @ R.transpose(1, 2)
    cov = M @ Sigma @ M.transpose(1, 2)
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    reg = (det * det / (det * det + 1e-7)).detach()
    g = rows.to(f64)
    L = (reg * (g[:, 2] * (c / det) + 2.0 * g[:, 3] * (-b / det) + g[:, 4] * (a / det)))[vis].sum()
    ds, dq = torch.autograd.grad(L, [s, q])
    return ds / mod, dq  # the kernel returns dL/d(mod * scale) (backward.cu:372-375)
