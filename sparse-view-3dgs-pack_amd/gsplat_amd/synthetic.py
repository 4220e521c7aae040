"""Seeded synthetic scenes and cameras - host-side numpy/torch only.

Cameras follow the reference's conventions exactly (restated, pinned by tests/golden):
  W2C from (R, t)            LGDWT-GS/utils/graphics_utils.py:38-51  (getWorld2View2, R stored transposed) -> world_to_camera
  projection                 LGDWT-GS/utils/graphics_utils.py:53-74  (getProjectionMatrix) -> perspective
  transposes / full proj /   LGDWT-GS/scene/cameras.py:86-89
  camera centre
Gaussian sets:
  init_like     what GaussianModel.create_from_pcd produces from a random point cloud
                (LGDWT-GS/scene/gaussian_model.py:149-176, dataset_readers.py:401-407)
  trained_like  anisotropic, random rotations / opacities / full SH - stresses sort and blend
"""
import math
from typing import NamedTuple

import numpy as np
import torch

C0 = 0.28209479177387814


class Camera(NamedTuple):
    image_height: int
    image_width: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor  # [4,4] = W2C^T
    full_proj_transform: torch.Tensor   # [4,4] = (P W2C)^T
    camera_center: torch.Tensor         # [3]

    @property
    def tanfovx(self):
        return math.tan(self.FoVx * 0.5)

    @property
    def tanfovy(self):
        return math.tan(self.FoVy * 0.5)


def fov2focal(fov, pixels):
    """Pinhole: half the image spans tan(fov / 2) focal lengths."""
    return 0.5 * pixels / math.tan(0.5 * fov)


def focal2fov(focal, pixels):
    return 2.0 * math.atan(0.5 * pixels / focal)


def world_to_camera(R_c2w, t):
    """4x4 world-to-camera matrix [R_c2w^T | t; 0 0 0 1] (float32).  The reference reaches the same matrix through two
    4x4 inversions with a zero re-centring in between (graphics_utils.py:38-51); the closed form is what those amount to."""
    M = np.eye(4, dtype=np.float64)
    M[:3, :3] = np.asarray(R_c2w, dtype=np.float64).T
    M[:3, 3] = np.asarray(t, dtype=np.float64)
    return M.astype(np.float32)


def perspective(znear, zfar, fovX, fovY):
    """Symmetric-frustum perspective matrix of the reference's convention (graphics_utils.py:53-74): x, y scaled by
    cot(fov / 2), w = z_view, depth mapped to z_ndc in [0, 1] - five non-zero entries."""
    P = torch.zeros(4, 4)
    P[0, 0] = 1.0 / math.tan(0.5 * fovX)
    P[1, 1] = 1.0 / math.tan(0.5 * fovY)
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -zfar * znear / (zfar - znear)
    P[3, 2] = 1.0
    return P


def make_camera(R, t, FoVx, FoVy, W, H, znear=0.01, zfar=100.0):
    """R: camera-to-world rotation (the reference stores it transposed), t: W2C translation."""
    wvt = torch.tensor(world_to_camera(R, t)).transpose(0, 1)
    proj = perspective(znear, zfar, FoVx, FoVy).transpose(0, 1)
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
    center = wvt.inverse()[3, :3]
    return Camera(int(H), int(W), float(FoVx), float(FoVy), wvt.contiguous(), full.contiguous(), center.contiguous())


def look_at_camera(eye, W, H, FoVx=0.6911112070083618, target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0)):
    """COLMAP-style camera (x right, y down, z forward) at `eye` looking at `target`."""
    eye = np.asarray(eye, dtype=np.float64)
    fwd = np.asarray(target, dtype=np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.asarray(up, dtype=np.float64))
    if np.linalg.norm(right) < 1e-8:
        right = np.cross(fwd, np.array([0.0, 1.0, 0.0]))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    c2w_R = np.stack([right, down, fwd], axis=1)  # columns = camera axes in world
    w2c_R = c2w_R.T
    t = -w2c_R @ eye
    FoVy = focal2fov(fov2focal(FoVx, W), H)
    return make_camera(c2w_R, t, FoVx, FoVy, W, H)


def orbit_cameras(W, H, n_azimuth=8, elevations_deg=(-10.0, 20.0, 45.0), radius=4.031, FoVx=0.6911112070083618):
    """NeRF-synthetic-like orbit: 8 azimuths x 3 elevations = 24 views."""
    cams = []
    for el in elevations_deg:
        for k in range(n_azimuth):
            az = 2 * math.pi * k / n_azimuth
            e = math.radians(el)
            eye = (radius * math.cos(e) * math.cos(az), radius * math.cos(e) * math.sin(az), radius * math.sin(e))
            cams.append(look_at_camera(eye, W, H, FoVx))
    return cams


def rgb2sh(rgb):
    return (rgb - 0.5) / C0


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def brute_force_knn_dist2(xyz, chunk=2048):
    """Exact mean squared distance to the 3 nearest other points (host, float32) - only used to
    build synthetic scales when no kNN operator is supplied."""
    P = xyz.shape[0]
    out = torch.empty(P, dtype=torch.float32)
    for s in range(0, P, chunk):
        d = torch.cdist(xyz[s:s + chunk].double(), xyz.double()) ** 2
        idx = torch.arange(s, min(s + chunk, P))
        d[torch.arange(idx.numel()), idx] = float("inf")
        out[s:s + chunk] = d.topk(3, dim=1, largest=False).values.mean(dim=1).float()
    return out


def _points(P, seed):
    rng = np.random.RandomState(seed)
    return torch.from_numpy((rng.random_sample((P, 3)) * 2.6 - 1.3).astype(np.float32)), rng


def init_like(P, seed=0, knn=None, sh_degree=3):
    """Parameters in their *activated* form, as render() hands them to the rasterizer."""
    xyz, rng = _points(P, seed)
    dist2 = (knn(xyz) if knn is not None else brute_force_knn_dist2(xyz)).clamp_min(1e-7)
    scales = torch.sqrt(dist2)[:, None].repeat(1, 3)  # exp(log(sqrt(dist2)))
    rots = torch.zeros((P, 4), dtype=torch.float32)
    rots[:, 0] = 1
    opac = torch.full((P, 1), 0.1, dtype=torch.float32)
    shs_np = rng.random_sample((P, 3)) / 255.0          # dataset_readers.py:406
    colour = torch.from_numpy((shs_np * C0 + 0.5).astype(np.float32))  # SH2RGB
    M = 16
    sh = torch.zeros((P, M, 3), dtype=torch.float32)
    sh[:, 0, :] = rgb2sh(colour)
    return dict(means3D=xyz, scales=scales, rotations=rots, opacities=opac, shs=sh, sh_degree=sh_degree)


def trained_like(P, seed=0, knn=None, sh_degree=3, scale_mult=0.5):
    xyz, rng = _points(P, seed)
    g = torch.Generator().manual_seed(seed)
    dist2 = (knn(xyz) if knn is not None else brute_force_knn_dist2(xyz)).clamp_min(1e-7)
    base = torch.log(scale_mult * torch.sqrt(dist2))[:, None]
    scales = torch.exp(base + 0.7 * torch.randn((P, 3), generator=g))
    q = torch.randn((P, 4), generator=g)
    rots = q / q.norm(dim=1, keepdim=True)
    opac = torch.sigmoid(2.0 * torch.randn((P, 1), generator=g))
    sh = torch.zeros((P, 16, 3), dtype=torch.float32)
    sh[:, 0, :] = torch.randn((P, 3), generator=g)
    sh[:, 1:, :] = 0.15 * torch.randn((P, 15, 3), generator=g)
    return dict(means3D=xyz, scales=scales.float(), rotations=rots.float(), opacities=opac.float(), shs=sh,
                sh_degree=sh_degree)


def ball_in_shell(P, seed=0, knn=None, sh_degree=3, ball_radius=0.8, shell_radius=(1.3, 1.35), shell_opacity=(0.02, 0.1)):
    """A scene with an UNSATURATED background, for timing what depth limits and early termination cannot help with:
    half of the Gaussians as trained_like inside a ball of `ball_radius` (every pixel that sees it saturates), the other
    half on a thin spherical shell around it with opacities in `shell_opacity` - from the orbit cameras the ball covers
    about half of a 1080p image and everything outside it sees two thin layers that never saturate (final T stays
    near 0.9), so those tiles - and their 3x3 neighbours - keep their whole lists on every visit."""
    rng = np.random.RandomState(seed)
    g = torch.Generator().manual_seed(seed)
    n_ball = P // 2
    n_shell = P - n_ball
    d = rng.standard_normal((P, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = np.empty(P)
    r[:n_ball] = ball_radius * rng.random_sample(n_ball) ** (1.0 / 3.0)
    r[n_ball:] = shell_radius[0] + (shell_radius[1] - shell_radius[0]) * rng.random_sample(n_shell)
    xyz = torch.from_numpy((d * r[:, None]).astype(np.float32))
    dist2 = (knn(xyz) if knn is not None else brute_force_knn_dist2(xyz)).clamp_min(1e-7)
    base = torch.log(0.5 * torch.sqrt(dist2))[:, None]
    scales = torch.exp(base + 0.7 * torch.randn((P, 3), generator=g))
    q = torch.randn((P, 4), generator=g)
    rots = q / q.norm(dim=1, keepdim=True)
    opac = torch.sigmoid(2.0 * torch.randn((P, 1), generator=g))
    opac[n_ball:] = shell_opacity[0] + (shell_opacity[1] - shell_opacity[0]) * torch.rand((n_shell, 1), generator=g)
    sh = torch.zeros((P, 16, 3), dtype=torch.float32)
    sh[:, 0, :] = torch.randn((P, 3), generator=g)
    sh[:, 1:, :] = 0.15 * torch.randn((P, 15, 3), generator=g)
    return dict(means3D=xyz, scales=scales.float(), rotations=rots.float(), opacities=opac.float(), shs=sh,
                sh_degree=sh_degree)


def morton_order(xyz, bits=10):
    """Permutation that sorts points along the 3-D Morton (Z-order) curve of their bounding box (`bits` per axis): rows that
    are neighbours in memory are neighbours in space.  xyz: [P, 3] tensor -> int64 tensor [P] on xyz's device (stable: ties keep their order)."""
    x = xyz.detach().to(torch.float64)   # (on the tensor's own device: a million points are a millisecond on the GPU)
    lo, hi = x.min(dim=0).values, x.max(dim=0).values
    q = ((x - lo) / (hi - lo).clamp_min(1e-30) * ((1 << bits) - 1)).round().to(torch.int64).clamp_(0, (1 << bits) - 1)
    code = torch.zeros((x.shape[0],), dtype=torch.int64, device=x.device)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return torch.sort(code, stable=True).indices


def spatially_ordered(scene):
    """The scene with its Gaussians (every [P, ...] tensor) in Morton order of their centres."""
    perm = morton_order(scene["means3D"])
    P = scene["means3D"].shape[0]
    return {k: (v[perm].contiguous() if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == P else v) for k, v in scene.items()}
