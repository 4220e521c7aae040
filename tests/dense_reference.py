"""Independent dense float64 formulation of the differentiable splatting model.

Written from the math, not from the oracle: every pixel evaluates every Gaussian, masks stand in
for the skip rules, a global depth sort gives the blend order, exclusive cumulative products give
the transmittance, and torch autograd supplies all parameter gradients.  Integer decisions that are
not differentiable (radius, tile rectangle, visibility) are taken from the same closed-form rules
(forward.cu:179-268) evaluated here in float64.  Feasible for P <= ~300 and <= 64x64 images.
"""
import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def sh_basis(deg, d):
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    b = [torch.full_like(x, SH_C0)]
    if deg > 0:
        b += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2 * zz - xx - yy), SH_C2[3] * xz, SH_C2[4] * (xx - yy)]
    if deg > 2:
        b += [SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * xy * z, SH_C3[2] * y * (4 * zz - xx - yy),
              SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy), SH_C3[4] * x * (4 * zz - xx - yy),
              SH_C3[5] * z * (xx - yy), SH_C3[6] * x * (xx - 3 * yy)]
    return torch.stack(b, dim=1)  # [P, (deg+1)^2]


def quat_to_rot(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    return R


class _AAScale(torch.autograd.Function):
    """h = sqrt(max(2.5e-5, det(cov)/det(cov+0.3I))).  Its backward is the reference's closed form
    (backward.cu:208-246), which evaluates the derivative of the ratio at the ALREADY +0.3-SHIFTED
    diagonal (x = a+0.3, y = c+0.3) instead of at (a, c): a quirk of the reference that any parity
    target has to reproduce, so plain autograd cannot be used for this one factor."""

    @staticmethod
    def forward(ctx, a, b, c):
        det_cov = a * c - b * b
        det = (a + 0.3) * (c + 0.3) - b * b
        ratio = det_cov / det
        h = torch.sqrt(torch.clamp_min(ratio, 0.000025))
        ctx.save_for_backward(a, b, c, ratio, h)
        return h

    @staticmethod
    def backward(ctx, d_h):
        a, b, c, ratio, h = ctx.saved_tensors
        d_root = torch.where(ratio <= 0.000025, torch.zeros_like(d_h), d_h / (2 * h))
        w = 0.3
        x, y, z = a + 0.3, c + 0.3, b
        denom_f = d_root / (w * w + w * (x + y) + x * y - z * z) ** 2
        return w * (w * y + y * y + z * z) * denom_f, -2.0 * w * z * (w + x + y) * denom_f, \
            w * (w * x + x * x + z * z) * denom_f


def render(scene, cam, bg, antialiasing=False, want_invdepth=True):
    """scene tensors must be float64 leaves (requires_grad as desired).  Returns dict."""
    f64 = torch.float64
    means = scene["means3D"]
    P = means.shape[0]
    H, W = cam.image_height, cam.image_width
    V = cam.world_view_transform.to(f64)      # W2C^T
    PV = cam.full_proj_transform.to(f64)      # (P W2C)^T
    campos = cam.camera_center.to(f64)
    tanx, tany = cam.tanfovx, cam.tanfovy
    fx, fy = W / (2 * tanx), H / (2 * tany)
    mod = scene.get("scale_modifier", 1.0)

    hom = torch.cat([means, torch.ones((P, 1), dtype=f64)], dim=1)
    p_view = hom @ V
    p_hom = hom @ PV
    p_w = 1.0 / (p_hom[:, 3] + 1e-7)
    p_proj = p_hom[:, :3] * p_w[:, None]
    if scene.get("ndc_probe") is not None:  # zero leaf whose gradient is the reference's dL_dmeans2D[:, :2]
        p_proj = torch.cat([p_proj[:, :2] + scene["ndc_probe"], p_proj[:, 2:]], dim=1)
    tz = p_view[:, 2]
    visible = tz > 0.2

    if scene.get("cov3D_precomp") is not None:
        c6 = scene["cov3D_precomp"]
        Sigma = torch.stack([c6[:, 0], c6[:, 1], c6[:, 2], c6[:, 1], c6[:, 3], c6[:, 4], c6[:, 2], c6[:, 4], c6[:, 5]],
                            dim=1).reshape(P, 3, 3)
    else:
        R = quat_to_rot(scene["rotations"])
        S = torch.diag_embed(mod * scene["scales"])
        Sigma = R @ S @ S @ R.transpose(1, 2)

    limx, limy = 1.3 * tanx, 1.3 * tany
    txtz, tytz = p_view[:, 0] / tz, p_view[:, 1] / tz
    cx = (txtz < -limx) | (txtz > limx)
    cy = (tytz < -limy) | (tytz > limy)
    # the reference treats the clamped t.x = lim*t.z as a constant in its backward (backward.cu:310-313)
    tx = torch.where(cx, (txtz.clamp(-limx, limx) * tz).detach(), p_view[:, 0])
    ty = torch.where(cy, (tytz.clamp(-limy, limy) * tz).detach(), p_view[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], dim=1).reshape(P, 2, 3)
    Wr = V[:3, :3].transpose(0, 1)            # W2C rotation
    M = J @ Wr
    cov = M @ Sigma @ M.transpose(1, 2)
    a, b, c = cov[:, 0, 0], cov[:, 0, 1], cov[:, 1, 1]
    opac = scene["opacities"].reshape(-1)
    if antialiasing:
        opac = opac * _AAScale.apply(a, b, c)
    a = a + 0.3
    c = c + 0.3
    det = a * c - b * b
    visible = visible & (det != 0)
    det_safe = torch.where(det != 0, det, torch.ones_like(det))
    conic = torch.stack([c / det_safe, -b / det_safe, a / det_safe], dim=1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    lam2 = mid - torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(lam, lam2))).detach()
    px = ((p_proj[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((p_proj[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + 15) // 16, (H + 15) // 16

    def trunc_clamp(v, hi):
        return torch.clamp(torch.trunc(v), 0, hi)
    rminx = trunc_clamp((px.detach() - radius) / 16, gx)
    rminy = trunc_clamp((py.detach() - radius) / 16, gy)
    rmaxx = trunc_clamp((px.detach() + radius + 15) / 16, gx)
    rmaxy = trunc_clamp((py.detach() + radius + 15) / 16, gy)
    visible = visible & (((rmaxx - rminx) * (rmaxy - rminy)) > 0)
    radii = torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32)

    if scene.get("colors_precomp") is not None:
        rgb = scene["colors_precomp"]
    else:
        deg = scene["sh_degree"]
        d = means - campos[None]
        d = d / d.norm(dim=1, keepdim=True)
        B = sh_basis(deg, d)
        rgb = torch.einsum("pk,pkc->pc", B, scene["shs"][:, : (deg + 1) ** 2, :]) + 0.5
        rgb = torch.clamp_min(rgb, 0.0)

    # global blend order: depth ascending, ties by index (stable) - the (tile|depth) radix sort of
    # rasterizer_impl.cu:306-311 restricted to one tile
    order = torch.sort(tz.detach().float(), stable=True).indices
    order = order[visible[order]]
    ys, xs = torch.meshgrid(torch.arange(H, dtype=f64), torch.arange(W, dtype=f64), indexing="ij")
    pix_x, pix_y = xs.reshape(-1), ys.reshape(-1)           # [N]
    tile_x, tile_y = torch.div(pix_x, 16, rounding_mode="floor"), torch.div(pix_y, 16, rounding_mode="floor")

    o = order
    dx = px[o][None, :] - pix_x[:, None]                      # [N, G]
    dy = py[o][None, :] - pix_y[:, None]
    cn = conic[o]
    power = -0.5 * (cn[:, 0][None] * dx * dx + cn[:, 2][None] * dy * dy) - cn[:, 1][None] * dx * dy
    in_rect = (tile_x[:, None] >= rminx[o][None]) & (tile_x[:, None] < rmaxx[o][None]) & \
              (tile_y[:, None] >= rminy[o][None]) & (tile_y[:, None] < rmaxy[o][None])
    alpha = torch.clamp_max(opac[o][None] * torch.exp(power), 0.99)
    live = in_rect & (power <= 0) & (alpha >= 1.0 / 255.0)
    alpha = torch.where(live, alpha, torch.zeros_like(alpha))
    Tinc = torch.cumprod(1 - alpha, dim=1)
    stop = live & (Tinc < 0.0001)
    dead = torch.cumsum(stop.to(torch.int64), dim=1) > 0       # the stopping Gaussian itself is not blended
    alpha = torch.where(dead, torch.zeros_like(alpha), alpha)
    Tinc = torch.cumprod(1 - alpha, dim=1)
    Texc = torch.cat([torch.ones((alpha.shape[0], 1), dtype=f64), Tinc[:, :-1]], dim=1)
    w = alpha * Texc
    T_final = Tinc[:, -1] if alpha.shape[1] > 0 else torch.ones(alpha.shape[0], dtype=f64)
    color = w @ rgb[o] + T_final[:, None] * bg.to(f64)[None]
    invd = w @ (1.0 / tz[o])
    depth = w @ tz[o]            # FSGS generation: sum depth alpha T (-confidence forward.cu:361)
    alpha_img = w.sum(dim=1)     # ... and sum alpha T (:360)
    # n_contrib: 1-based position (within the tile's list = Gaussians whose rect holds the tile) of
    # the last blended Gaussian
    pos_in_tile = torch.cumsum(in_rect.to(torch.int64), dim=1)
    blended = (alpha > 0)
    n_contrib = torch.where(blended, pos_in_tile, torch.zeros_like(pos_in_tile)).max(dim=1).values \
        if alpha.shape[1] > 0 else torch.zeros(alpha.shape[0], dtype=torch.int64)
    return dict(color=color.t().reshape(3, H, W), invdepth=invd.reshape(1, H, W), radii=radii,
                depth=depth.reshape(1, H, W), alpha=alpha_img.reshape(1, H, W),
                final_T=T_final.reshape(H, W), n_contrib=n_contrib.reshape(H, W), means2D=torch.stack([px, py], 1))
