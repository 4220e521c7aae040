"""Shared helpers for the test-suite: running a scene through a rasterizer class and
threshold-aware comparisons."""
import math

import torch


def settings_for(Settings, cam, bg, sh_degree, device, antialiasing=False, scale_modifier=1.0, debug=False):
    return Settings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
        bg=bg.to(device), scale_modifier=scale_modifier, viewmatrix=cam.world_view_transform.to(device),
        projmatrix=cam.full_proj_transform.to(device), sh_degree=sh_degree, campos=cam.camera_center.to(device),
        prefiltered=False, debug=debug, antialiasing=antialiasing)


PARAMS = ("means3D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp")


def run_scene(Rasterizer, Settings, scene, cam, device, bg=None, antialiasing=False, dL_dcolor=None,
              dL_dinvdepth=None, backward=True):
    """Render `scene` (dict of activated tensors) and, if asked, backprop the given image gradients.
    Returns dict(color, radii, invdepth, grads{name: tensor}, means2D_grad)."""
    bg = torch.zeros(3) if bg is None else bg
    p = {}
    for k in PARAMS:
        v = scene.get(k)
        p[k] = None if v is None else v.detach().clone().to(device).requires_grad_(backward)
    means2D = torch.zeros_like(p["means3D"], requires_grad=backward)
    rs = settings_for(Settings, cam, bg, scene.get("sh_degree", 0), device, antialiasing,
                      scene.get("scale_modifier", 1.0))
    rast = Rasterizer(raster_settings=rs)
    color, radii, invdepth = rast(means3D=p["means3D"], means2D=means2D, opacities=p["opacities"], shs=p["shs"],
                                  colors_precomp=p["colors_precomp"], scales=p["scales"], rotations=p["rotations"],
                                  cov3D_precomp=p["cov3D_precomp"])
    out = dict(color=color.detach(), radii=radii.detach(), invdepth=invdepth.detach(), grads={})
    if backward:
        H, W = cam.image_height, cam.image_width
        if dL_dcolor is None:
            g = torch.Generator().manual_seed(99)
            dL_dcolor = torch.randn((3, H, W), generator=g)
        loss = (color * dL_dcolor.to(device)).sum()
        if dL_dinvdepth is not None:
            loss = loss + (invdepth * dL_dinvdepth.to(device)).sum()
        loss.backward()
        for k in PARAMS:
            if p[k] is not None:
                out["grads"][k] = p[k].grad.detach() if p[k].grad is not None else torch.zeros_like(p[k])
        out["grads"]["means2D"] = means2D.grad.detach() if means2D.grad is not None else torch.zeros_like(means2D)
    return out


def rel_err(a, b):
    """max |a-b| / max(1e-12, max|b|) - tensor-level relative error."""
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / max(1e-12, float(b.abs().max())))


def assert_close(a, b, tol, what=""):
    e = rel_err(a, b)
    assert e <= tol, "%s: relative error %.3e > %.1e" % (what, e, tol)


# ---- gradient comparison at the stated tolerance ---------------------------------------------------------------
TOL = 1e-4        # north_star: within 1e-4 relative (to the tensor's largest entry), fp32
# dL_dscales / dL_drotations pass through the conic -> cov2D -> cov3D -> (scale, quaternion) chain, which amplifies any
# fp32 rounding by the footprint's anisotropy (tests/test_gpu_fullsize.py).  Since round 4 the product
# evaluates that chain in float64 on float64 sums, and the GPU tests compare it with the oracle's double evaluation of the
# reference formula (conftest._gpu_parity_uses_the_exact_chain): the same TOL as every other tensor.  (Rounds 1-3: 5e-4.)
CHAIN_TOL = TOL
CHAIN_TENSORS = ("scales", "rotations")
MEASURED = {}
ROTATION_COLLAPSED = 1e-3


def rotation_scale_floor(grads, scene):
    """A scale for dL_drotations that does not collapse when the tensor is (nearly) zero: a rotation by a small angle moves
    Sigma = R S^2 R^T by ~angle x s^2, so |dL_dq| is at most of the order of |dL_ds| x s - and EXACTLY zero for an isotropic
    Gaussian (init-like scenes), where what an implementation returns is pure rounding residue of that magnitude times
    2^-24.  "1e-4 of the tensor's largest entry" is therefore read as 1e-4 of max(largest entry, largest |dL_ds| x s)."""
    ds, s = grads.get("scales"), scene.get("scales")
    if ds is None or s is None:
        return 0.0
    return float((ds.detach().abs().cpu().amax(1) * s.detach().abs().cpu().amax(1)).max())


def check_grads(hg, og, name, tol=TOL, chain_tol=CHAIN_TOL, scene=None):
    """Compare two dicts of gradient tensors (max |a-b| / max |b| per tensor); print, record under gpurun_out/, assert.
    scene (optional): the parameter dict, for rotation_scale_floor."""
    import json
    import os
    errs = {}
    for k in og:
        if og[k] is None or hg.get(k) is None:
            continue
        errs[k] = rel_err(hg[k], og[k])
        if k == "rotations" and scene is not None:
            # only where the tensor has collapsed (isotropic, init-like scene: largest entry below 1e-3 of what a rotation
            # gradient of this scene could be - measured there 1e-10 ... 1e-9 of the floor); everywhere else the stated scale
            floor = rotation_scale_floor(og, scene)
            if float(og[k].abs().max()) < ROTATION_COLLAPSED * floor:
                errs[k] = float((hg[k].double().cpu() - og[k].double().cpu()).abs().max()) / max(floor, 1e-12)
    MEASURED[name] = errs
    print("grads %s: %s" % (name, {k: "%.1e" % v for k, v in errs.items()}))
    try:
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        json.dump(MEASURED, open(os.path.join(out, "parity_small_scenes.json"), "w"), indent=1)
    except OSError:
        pass
    for k, e in errs.items():
        t = chain_tol if k in CHAIN_TENSORS else tol
        assert e <= t, "%s: dL_d%s rel err %.3e > %.0e" % (name, k, e, t)
    return errs


# ---- instance lists, independent of where a tile's list lies inside point_list ---------------------------------------
def canonical_lists(st):
    """-> (counts[T], keys[R]): keys = (tile << 32 | Gaussian id) of every list entry, tile after tile, each tile's entries
    in list order.  The LSD binning stores the lists in tile order; region binning (GsView.tile_cull = 2) places them
    wherever its regions reserved room - ranges[] is the only map.  Also checks that the ranges tile [0, R) exactly."""
    import numpy as np
    T = st["ranges"].reshape(-1, 2).long().numpy()
    counts = T[:, 1] - T[:, 0]
    pl = st["point_list"].long().numpy()
    assert (counts >= 0).all() and int(counts.sum()) == pl.size, (int(counts.sum()), pl.size)
    nz = counts > 0
    if nz.any():
        order = np.argsort(T[nz, 0], kind="stable")
        s0, c0 = T[nz, 0][order], counts[nz][order]
        assert s0[0] == 0 and (s0[1:] == s0[:-1] + c0[:-1]).all(), "tile ranges overlap or leave holes"
    tiles = np.repeat(np.arange(len(counts), dtype=np.int64), counts)
    first = np.cumsum(counts) - counts                       # canonical start of each tile
    src = np.repeat(T[:, 0] - first, counts) + np.arange(int(counts.sum()), dtype=np.int64)
    return counts, (tiles << 32) | pl[src]
