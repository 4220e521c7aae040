"""Generates tests/golden/*.npz by IMPORTING the reference's own Python helpers (CPU, this
container only - /root/reference does not exist on the GPU box, so the outputs are committed).

Run:  python tests/golden/make_golden.py
Fixtures are data only (inputs + the reference's outputs); no reference source is copied.

  sh_colour.npz   eval_sh (LGDWT-GS/utils/sh_utils.py) + the clamp of
                  LGDWT-GS/gaussian_renderer/__init__.py:76-80, with autograd gradients
                  -> pins computeColorFromSH forward (forward.cu:20-71) and backward (backward.cu:23-142)
  cameras.npz     getWorld2View2 / getProjectionMatrix / focal2fov / fov2focal
                  (LGDWT-GS/utils/graphics_utils.py) + the 4 lines of LGDWT-GS/scene/cameras.py:86-89
                  -> pins viewmatrix / projmatrix / campos conventions
  image_losses.npz  l1_loss, ssim (+autograd), psnr  (gaussian-splatting/utils/loss_utils.py,
                  LGDWT-GS/utils/image_utils.py) on seeded torch.rand images (recipe of
                  fused-ssim/tests/test.py:58-91 at small sizes)
  schedule.npz    get_expon_lr_func, inverse_sigmoid (LGDWT-GS/utils/general_utils.py), RGB2SH/SH2RGB
  nir_loss.npz    the multispectral step's loss (mult-dwtgs/train_nir.py:88-104) from the reference's own functions
                  (mult-dwtgs/utils/loss_utils.py: l1_loss, ssim, combined_nir_loss) with autograd gradients
                  -> pins gsplat_amd.trainer.NirCriterion
  geometry.npz    the reference's python covariance path - build_scaling_rotation / build_rotation / strip_symmetric
                  (LGDWT-GS/utils/general_utils.py:64-110; their hard-coded device="cuda" is redirected to the CPU) composed
                  as build_covariance_from_scaling_rotation does (LGDWT-GS/scene/gaussian_model.py:33-37: L = R S,
                  Sigma = L L^T, upper triangle), with autograd gradients w.r.t. scales and (unit) quaternions, and
                  geom_transform_points (LGDWT-GS/utils/graphics_utils.py:22-29: the "+1e-7" homogeneous divide)
                  -> pins computeCov3D forward (forward.cu:114-148) and backward (backward.cu:330-393), the projection
                  of preprocessCUDA (forward.cu:193-195), and - rendered with cov3D_precomp - the python-cov path of
                  LGDWT-GS/gaussian_renderer/__init__.py:64-68
  colmap/         a small COLMAP model (cameras.bin, images.bin, points3D.bin + the .txt forms) written by
                  gsplat_amd.io.write_colmap_binary from seeded data, and colmap_expected.npz = what the REFERENCE's
                  own readers (LGDWT-GS/scene/colmap_loader.py) return for those files, plus its qvec2rotmat /
                  rotmat2qvec -> pins the byte layout and conventions of gsplat_amd/io.py's readers
  lgdwt_loss.npz  the reference's OWN DWT-loss code - LGDWT-GS/utils/loss_utils.py: get_dwt_subbands (:106-153),
                  compute_elf_map (:336-366), compute_patch_dwt_loss (:368-442), l1_loss, ssim - imported and run here on
                  seeded images at 128x128, 131x260 and 256x384, values + autograd gradients, and the defaults of
                  LGDWT-GS/arguments/__init__.py:103-122 read from the imported OptimizationParams.  The module's top-level
                  `from pytorch_wavelets import DWTForward` (third party, absent offline) is satisfied by registering
                  tests/torch_loss_reference.DWTForward - an independent F.conv2d statement of that package's published
                  Haar analysis step - in sys.modules; everything AROUND the 2x2 Haar butterfly (band slicing, unfold
                  order, kthvalue index, ">=" selection, the HH weight, 1e-8, align_corners=False, l1 means) is the
                  reference's code running.  -> pins D1 (layout), D2, D3, D4 of SURVEY 8(a)
  depth_reg.npz   the depth-regularisation term of the step (LGDWT-GS/train.py:69, 204-216): the reference's own statements,
                  read from train.py by gen_depth_reg when it runs and executed on CPU stand-ins; weights of the schedule,
                  the term, its gradient w.r.t. the rendered inverse depth -> pins gs_depth_l1_* and Trainer's depth term
"""
import importlib.util
import math
import os

import numpy as np
import torch

REF = "/root/reference/fs3dgs_benchmark"
HERE = os.path.dirname(os.path.abspath(__file__))


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


sh_utils = load(REF + "/LGDWT-GS/utils/sh_utils.py", "ref_sh_utils")
graphics = load(REF + "/LGDWT-GS/utils/graphics_utils.py", "ref_graphics_utils")
image_utils = load(REF + "/LGDWT-GS/utils/image_utils.py", "ref_image_utils")
general = load(REF + "/LGDWT-GS/utils/general_utils.py", "ref_general_utils")
loss_utils = load(REF + "/gaussian-splatting/utils/loss_utils.py", "ref_loss_utils")


class _CpuTorch:
    """`torch` as seen by the reference's general_utils: three helpers there hard-code device="cuda"
    (general_utils.py:65,83,102); the same calls, with the tensors created on the CPU."""

    def __getattr__(self, name):
        return getattr(torch, name)

    @staticmethod
    def zeros(*a, **k):
        k.pop("device", None)
        return torch.zeros(*a, **k)


def gen_geometry():
    general.torch = _CpuTorch()
    try:
        g = torch.Generator().manual_seed(4321)
        P = 128
        scales = torch.exp(torch.randn((P, 3), generator=g) * 0.9 - 2.5)
        q = torch.randn((P, 4), generator=g)
        q = q / q.norm(dim=1, keepdim=True)  # the caller hands F.normalize(_rotation) to the rasterizer
        w6 = torch.randn((P, 6), generator=g)
        out = dict(scales=scales.numpy(), quats=q.numpy(), w6=w6.numpy())
        for tag, mod in (("m10", 1.0), ("m07", 0.7)):
            s = scales.clone().requires_grad_(True)
            r = q.clone().requires_grad_(True)
            L = general.build_scaling_rotation(mod * s, r)          # gaussian_model.py:34
            actual_covariance = L @ L.transpose(1, 2)               # :35
            symm = general.strip_symmetric(actual_covariance)       # :36
            (symm * w6).sum().backward()
            out["cov6_" + tag] = symm.detach().numpy()
            out["dscales_" + tag] = s.grad.numpy()
            out["dquats_" + tag] = r.grad.numpy()  # through build_rotation's own normalisation: tangential part only
        out["mods"] = np.array([1.0, 0.7])
        # homogeneous projection with the camera conventions of cameras.py:86-89
        rng = np.random.RandomState(11)
        R = np.linalg.qr(rng.randn(3, 3))[0]
        if np.linalg.det(R) < 0:
            R[:, 0] = -R[:, 0]
        t = np.array([0.3, -0.2, 4.0])
        W, H = 1237, 822
        FoVx = 0.69
        FoVy = graphics.focal2fov(graphics.fov2focal(FoVx, W), H)
        wvt = torch.tensor(graphics.getWorld2View2(R, t)).transpose(0, 1)
        proj = graphics.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=FoVx, fovY=FoVy).transpose(0, 1)
        full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
        pts = torch.rand((P, 3), generator=g) * 2.6 - 1.3
        out["points"] = pts.numpy()
        out["full_proj_transform"] = full.numpy()
        out["world_view_transform"] = wvt.numpy()
        out["p_proj"] = graphics.geom_transform_points(pts, full).numpy()
        out["p_view"] = graphics.geom_transform_points(pts, wvt).numpy()
        np.savez(os.path.join(HERE, "geometry.npz"), **out)
    finally:
        general.torch = torch


def gen_nir_loss():
    ref = load(REF + "/LGDWT-GS/mult-dwtgs/utils/loss_utils.py", "ref_nir_loss_utils")
    g = torch.Generator().manual_seed(77)
    H, W = 45, 61
    image = torch.rand((3, H, W), generator=g).requires_grad_(True)
    gt = torch.rand((3, H, W), generator=g)
    nir = (torch.rand((1, H, W), generator=g) * 1.2 - 0.1).requires_grad_(True)   # the NIR render is not clamped
    nir_gt = torch.rand((1, H, W), generator=g)
    lam, nir_weight = 0.2, 1.0
    Ll1 = ref.l1_loss(image, gt)
    loss = (1.0 - lam) * Ll1 + lam * (1.0 - ref.ssim(image, gt))
    nir_loss = ref.combined_nir_loss(nir, nir_gt, l1_weight=1.0, ssim_weight=0.2)
    total = loss + nir_weight * nir_loss
    total.backward()
    np.savez(os.path.join(HERE, "nir_loss.npz"), image=image.detach().numpy(), gt=gt.numpy(), nir=nir.detach().numpy(),
             nir_gt=nir_gt.numpy(), rgb_loss=loss.item(), nir_loss=nir_loss.item(), total=total.item(),
             d_image=image.grad.numpy(), d_nir=nir.grad.numpy())


def gen_colmap():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "..", "sparse-view-3dgs-pack_amd"))
    from gsplat_amd import io as gio
    ref = load(REF + "/LGDWT-GS/scene/colmap_loader.py", "ref_colmap_loader")
    rng = np.random.RandomState(7)
    d = os.path.join(HERE, "colmap")
    models = ["SIMPLE_PINHOLE", "PINHOLE", "SIMPLE_RADIAL", "OPENCV"]
    cams = {}
    for i, m in enumerate(models):
        n = gio.CAMERA_MODEL_NAMES[m].num_params
        cams[i + 1] = gio.ColmapCamera(i + 1, m, 640 + 16 * i, 480 + 8 * i, rng.uniform(300, 900, n))
    images = {}
    for i in range(6):
        q = rng.randn(4)
        q /= np.linalg.norm(q)
        npts = [0, 3, 5, 1, 0, 2][i]
        images[10 + i] = gio.ColmapImage(10 + i, q, rng.randn(3) * 2, 1 + i % 4, "img_%02d.png" % (5 - i),
                                         rng.uniform(0, 600, (npts, 2)), rng.randint(-1, 50, npts).astype(np.int64))
    pts = (rng.randn(9, 3), rng.randint(0, 256, (9, 3)), rng.uniform(0, 2, (9, 1)))
    gio.write_colmap_binary(d, cams, images, pts)
    with open(os.path.join(d, "cameras.txt"), "w") as f:   # only PINHOLE: the reference's text reader asserts it
        f.write("# Camera list with one line of data per camera:\n#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]\n")
        f.write("2 PINHOLE 656 488 %r %r %r %r\n" % tuple(float(x) for x in cams[2].params))
    with open(os.path.join(d, "images.txt"), "w") as f:
        f.write("# Image list with two lines of data per image:\n")
        for im in images.values():
            f.write("%d %s %s %d %s\n" % (im.id, " ".join(repr(float(x)) for x in im.qvec),
                                          " ".join(repr(float(x)) for x in im.tvec), im.camera_id, im.name))
            f.write(" ".join("%r %r %d" % (float(x), float(y), int(p)) for (x, y), p in zip(im.xys, im.point3D_ids)) + "\n")
    with open(os.path.join(d, "points3D.txt"), "w") as f:
        f.write("# 3D point list\n")
        for i in range(9):
            f.write("%d %s %d %d %d %r 1 2 3 4\n" % (i + 1, " ".join(repr(float(v)) for v in pts[0][i]),
                                                     *[int(v) for v in pts[1][i]], float(pts[2][i][0])))
    out = {}
    rc = ref.read_intrinsics_binary(os.path.join(d, "cameras.bin"))
    out["cam_ids"] = np.array(sorted(rc))
    for k in sorted(rc):
        out["cam%d_model" % k] = np.array(rc[k].model)
        out["cam%d_wh" % k] = np.array([rc[k].width, rc[k].height])
        out["cam%d_params" % k] = np.array(rc[k].params)
    for tag, ri in (("bin", ref.read_extrinsics_binary(os.path.join(d, "images.bin"))),
                    ("txt", ref.read_extrinsics_text(os.path.join(d, "images.txt")))):
        out["img_ids_" + tag] = np.array(sorted(ri))
        for k in sorted(ri):
            out["img%d_%s_qt" % (k, tag)] = np.concatenate((ri[k].qvec, ri[k].tvec))
            out["img%d_%s_cam" % (k, tag)] = np.array(ri[k].camera_id)
            out["img%d_%s_name" % (k, tag)] = np.array(ri[k].name)
            out["img%d_%s_xys" % (k, tag)] = np.asarray(ri[k].xys, dtype=np.float64).reshape(-1, 2)
            out["img%d_%s_p3d" % (k, tag)] = np.asarray(ri[k].point3D_ids, dtype=np.int64)
            out["img%d_%s_R" % (k, tag)] = ref.qvec2rotmat(ri[k].qvec)
            out["img%d_%s_q_back" % (k, tag)] = ref.rotmat2qvec(ref.qvec2rotmat(ri[k].qvec))
    rt = ref.read_intrinsics_text(os.path.join(d, "cameras.txt"))
    out["txtcam_params"] = np.array(rt[2].params)
    for tag, fn, name in (("bin", ref.read_points3D_binary, "points3D.bin"), ("txt", ref.read_points3D_text, "points3D.txt")):
        xyz, rgb, err = fn(os.path.join(d, name))
        out["pts_xyz_" + tag], out["pts_rgb_" + tag], out["pts_err_" + tag] = xyz, rgb, err
    np.savez(os.path.join(HERE, "colmap_expected.npz"), **out)


def gen_sh():
    out = {}
    g = torch.Generator().manual_seed(1234)
    P = 96
    means = (torch.rand((P, 3), generator=g) * 2.6 - 1.3)
    campos = torch.tensor([2.5, -3.0, 1.2])
    sh = torch.randn((P, 16, 3), generator=g) * 0.6
    sh[:, 0, :] = sh[:, 0, :] * 3.0 - 0.9  # make a good share of channels clamp at 0
    w = torch.randn((P, 3), generator=g)
    out["means"], out["campos"], out["sh"], out["w"] = means.numpy(), campos.numpy(), sh.numpy(), w.numpy()
    for deg in range(4):
        m = means.clone().requires_grad_(True)
        s = sh.clone().requires_grad_(True)
        shs_view = s.transpose(1, 2)  # [P,3,16] as in gaussian_renderer/__init__.py:76
        dir_pp = m - campos.repeat(P, 1)
        dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
        sh2rgb = sh_utils.eval_sh(deg, shs_view[..., :(deg + 1) ** 2], dir_pp_normalized)
        colors = torch.clamp_min(sh2rgb + 0.5, 0.0)
        (colors * w).sum().backward()
        out["colors_deg%d" % deg] = colors.detach().numpy()
        out["raw_deg%d" % deg] = (sh2rgb + 0.5).detach().numpy()
        out["dsh_deg%d" % deg] = s.grad.numpy()
        out["dmeans_deg%d" % deg] = (m.grad if m.grad is not None else torch.zeros_like(m)).numpy()
    np.savez(os.path.join(HERE, "sh_colour.npz"), **out)


def gen_cameras():
    out = {}
    rng = np.random.RandomState(7)
    Rs, ts, fx, fy, wvt, full, centers, WH = [], [], [], [], [], [], [], []
    for i in range(6):
        q = rng.randn(4)
        q /= np.linalg.norm(q)
        r, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                      [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                      [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])
        t = rng.randn(3) * 2
        W, H = [(400, 400), (800, 800), (1920, 1080), (1237, 822), (64, 48), (640, 360)][i]
        FoVx = 0.4 + 0.3 * rng.rand()
        FoVy = graphics.focal2fov(graphics.fov2focal(FoVx, W), H)
        world_view_transform = torch.tensor(graphics.getWorld2View2(R, t)).transpose(0, 1)
        projection_matrix = graphics.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=FoVx, fovY=FoVy).transpose(0, 1)
        full_proj_transform = (world_view_transform.unsqueeze(0).bmm(projection_matrix.unsqueeze(0))).squeeze(0)
        camera_center = world_view_transform.inverse()[3, :3]
        Rs.append(R); ts.append(t); fx.append(FoVx); fy.append(FoVy); WH.append((W, H))
        wvt.append(world_view_transform.numpy()); full.append(full_proj_transform.numpy())
        centers.append(camera_center.numpy())
    out = dict(R=np.array(Rs), t=np.array(ts), FoVx=np.array(fx), FoVy=np.array(fy), WH=np.array(WH),
               world_view_transform=np.array(wvt), full_proj_transform=np.array(full), camera_center=np.array(centers))
    np.savez(os.path.join(HERE, "cameras.npz"), **out)


def gen_losses():
    out = {}
    torch.manual_seed(0)
    for tag, (C, H, W) in {"a": (3, 37, 53), "b": (3, 64, 64), "c": (1, 20, 45)}.items():
        img1 = torch.rand((C, H, W))
        img2 = (img1 + 0.25 * torch.randn((C, H, W))).clamp(0, 1) if tag != "b" else torch.rand((C, H, W))
        x = img1.clone().requires_grad_(True)
        s = loss_utils.ssim(x, img2)
        s.backward()
        out["img1_" + tag], out["img2_" + tag] = img1.numpy(), img2.numpy()
        out["ssim_" + tag] = s.detach().numpy()
        out["dssim_" + tag] = x.grad.numpy()
        y = img1.clone().requires_grad_(True)
        l = loss_utils.l1_loss(y, img2)
        l.backward()
        out["l1_" + tag] = l.detach().numpy()
        out["dl1_" + tag] = y.grad.numpy()
        out["psnr_" + tag] = image_utils.psnr(img1[None], img2[None]).numpy()
    np.savez(os.path.join(HERE, "image_losses.npz"), **out)


def gen_schedule():
    f = general.get_expon_lr_func(lr_init=0.00016, lr_final=0.0000016, lr_delay_mult=0.01, max_steps=30000)
    steps = np.array([0, 1, 10, 100, 999, 1000, 7000, 15000, 29999, 30000, 40000])
    lr = np.array([f(int(s)) for s in steps])
    xs = torch.tensor([0.01, 0.1, 0.5, 0.9, 0.99])
    rgb = torch.tensor([[0.0, 0.5, 1.0], [0.25, 0.75, 0.1]])
    np.savez(os.path.join(HERE, "schedule.npz"), steps=steps, lr=lr, inv_sig_x=xs.numpy(),
             inv_sig=general.inverse_sigmoid(xs).numpy(), rgb=rgb.numpy(), rgb2sh=sh_utils.RGB2SH(rgb).numpy(),
             sh2rgb=sh_utils.SH2RGB(rgb).numpy())


def gen_lgdwt_loss():
    import argparse
    import sys
    import types
    sys.path.insert(0, os.path.dirname(HERE))
    import torch_loss_reference as tlr
    stub = types.ModuleType("pytorch_wavelets")
    stub.DWTForward = tlr.DWTForward
    sys.modules["pytorch_wavelets"] = stub
    try:
        ref = load(REF + "/LGDWT-GS/utils/loss_utils.py", "ref_lgdwt_loss_utils")
        args = load(REF + "/LGDWT-GS/arguments/__init__.py", "ref_lgdwt_arguments")
    finally:
        del sys.modules["pytorch_wavelets"]
    o = args.OptimizationParams(argparse.ArgumentParser())
    out = dict(
        band_names=np.array(["LL1", "LH1", "HL1", "HH1", "LL2", "LH2", "HL2", "HH2"]),
        dwt_weights=np.array([o.dwt_ll1_weight, o.dwt_lh1_weight, o.dwt_hl1_weight, o.dwt_hh1_weight,
                              o.dwt_ll2_weight, o.dwt_lh2_weight, o.dwt_hl2_weight, o.dwt_hh2_weight]),
        lambda_dssim=np.array(o.lambda_dssim), patch_dwt_weight=np.array(o.patch_dwt_weight),
        patch_size=np.array(o.patch_size), patch_percentile=np.array(o.patch_percentile),
        patch_lh1_weight=np.array(o.patch_dwt_lh1_weight), patch_hl1_weight=np.array(o.patch_dwt_hl1_weight),
        dwt_enable=np.array(o.dwt_enable), patch_dwt_enable=np.array(o.patch_dwt_enable),
        # LGDWT-GS/train.py:188-202 (not importable: needs CUDA + the scene stack); data, not code: the running mean
        # m <- 0.95 m + 0.05 base / (dwt + 1e-8) from m0 = 1 (train.py:78), scale = clamp(m, 0.1, 10)
        running_mean=np.array([0.95, 0.05, 1e-8, 1.0, 0.1, 10.0]))
    g = torch.Generator().manual_seed(20261004)
    allw = (1.0, 0.7, 1.3, 0.3, 0.5, 0.25, 0.9, 2.0)   # a second, all-non-zero weight set for the summed gradient
    out["dwt_weights_all"] = np.array(allw)
    for tag, (H, W), ps, (w_lh, w_hl), rs in (("a", (128, 128), 32, (1.0, 1.0), 1), ("b", (131, 260), 64, (1.0, 0.5), 1),
                                              ("c", (256, 384), 128, (1.0, 1.0), 4), ("d", (75, 141), 32, (0.7, 1.3), 1)):
        # images quantised to 8 bit like PILtoTorch (general_utils.py:21-27); rs: row stride at which the big arrays of
        # the largest case are kept (a fixture is a sample, the scalars cover every pixel)
        gt = torch.rand((1, 3, H, W), generator=g)
        gt[:, :, : H // 2] = gt[:, :, : H // 2] * 0.1 + 0.4       # a smooth half: high ELF, selected by the patch rule
        gt[:, :, :, W // 3: W // 2] *= 0.5
        gt = torch.round(gt * 255.0) / 255.0
        # (the render is continuous and here un-clamped: an 8-bit or clamped pred against an 8-bit gt makes band differences
        # cancel EXACTLY in places, where the sign() inside the L1 gradients is decided by the rounding of the Haar step
        # alone; one block where both images are 0 keeps the sign(0) = 0 case)
        pred = gt + 0.1 * torch.randn((1, 3, H, W), generator=g)
        gt[:, :, 8:16, 8:24] = 0.0
        pred[:, :, 8:16, 8:24] = 0.0
        out["gt_u8_" + tag] = torch.round(gt[0] * 255.0).to(torch.uint8).numpy()
        out["pred_" + tag] = pred[0].numpy()
        out["patch_args_" + tag] = np.array([ps, 0.2, w_lh, w_hl])
        out["row_stride_" + tag] = np.array(rs)
        # D1: the eight bands and the adjoint (cotangents: tests/torch_loss_reference.cotangent)
        x = pred.clone().requires_grad_(True)
        bands = ref.get_dwt_subbands(x)
        sum((bands[k] * tlr.cotangent(bands[k].shape, i)).sum() for i, k in enumerate(bands)).backward()
        for k in bands:
            out["band_%s_%s" % (k, tag)] = bands[k].detach()[0, :, ::rs].numpy()
        out["dbands_" + tag] = x.grad[0, :, ::rs].numpy()
        # D2: per-band L1 (train.py:132-164 weights them and adds them up); gradient of the all-bands weighted sum
        gb = ref.get_dwt_subbands(gt)
        x = pred.clone().requires_grad_(True)
        pb = ref.get_dwt_subbands(x)
        l1s = [ref.l1_loss(pb[k], gb[k]) for k in bands]
        sum(w * l for w, l in zip(allw, l1s)).backward()
        out["band_l1_" + tag] = np.array([float(l) for l in l1s])
        out["dl1_all_" + tag] = x.grad[0, :, ::rs].numpy()
        # D3
        elf = ref.compute_elf_map(gt)
        out["elf_" + tag] = elf[0, :, ::rs].numpy()
        # D4 (selected patches = those that receive a gradient: every selected patch of a noisy image does)
        x = pred.clone().requires_grad_(True)
        pl = ref.compute_patch_dwt_loss(x, gt, elf, patch_size=ps, percentile=0.2, lh1_weight=w_lh, hl1_weight=w_hl)
        pl.backward()
        out["patch_loss_" + tag] = pl.detach().numpy()
        out["dpatch_" + tag] = x.grad[0, :, ::rs].numpy()
        ny, nx = H // ps, W // ps
        gp = x.grad[0, :, : ny * ps, : nx * ps].abs().reshape(3, ny, ps, nx, ps).amax(dim=(0, 2, 4))
        out["patch_mask_" + tag] = (gp > 0).reshape(-1).numpy()
        # the other two terms of the base loss from the same module
        x = pred[0].clone().requires_grad_(True)
        l1 = ref.l1_loss(x, gt[0])
        ss = ref.ssim(x, gt[0])
        ((1.0 - o.lambda_dssim) * l1 + o.lambda_dssim * (1.0 - ss)).backward()
        out["l1_" + tag], out["ssim_" + tag] = l1.detach().numpy(), ss.detach().numpy()
        out["dbase_" + tag] = x.grad[:, ::rs].numpy()
    # too-small image: the patch term is 0 (loss_utils.py:386-387)
    small = torch.rand((1, 3, 40, 200), generator=g)
    out["patch_loss_small"] = ref.compute_patch_dwt_loss(small, small * 0.5, ref.compute_elf_map(small)).numpy()
    np.savez_compressed(os.path.join(HERE, "lgdwt_loss.npz"), **out)


def gen_depth_reg():
    """depth_reg.npz: the depth-regularisation term of the step, LGDWT-GS/train.py:69 (the weight schedule) and :204-216 (the
    term).  train.py cannot be imported (CUDA, the scene stack, argparse at import), so the generator READS those statements
    from the reference file when it runs, compiles them and executes them on CPU stand-ins for `viewpoint_cam` and `render_pkg`
    (tensors whose .cuda() is the identity): the reference's own statements compute the fixture - nothing of them is kept in
    this repository.  Outputs: the weights at a few iterations, and per case the pure term, the total loss increment and the
    autograd gradient with respect to the rendered inverse depth."""
    import argparse
    import textwrap
    import types
    src = open(REF + "/LGDWT-GS/train.py").read().split("\n")
    sched = [l for l in src if l.strip().startswith("depth_l1_weight = get_expon_lr_func(")]
    assert len(sched) == 1, sched
    a = next(i for i, l in enumerate(src) if l.strip() == "# Depth regularization")
    b = next(i for i in range(a, len(src)) if src[i].strip() == "loss.backward()")
    block = textwrap.dedent("\n".join(src[a + 1:b]))
    assert "Ll1depth_pure" in block and "depth_mask" in block and block.count("\n") < 20
    args = load(REF + "/LGDWT-GS/arguments/__init__.py", "ref_lgdwt_arguments_depth")
    opt = args.OptimizationParams(argparse.ArgumentParser())
    ns = dict(get_expon_lr_func=general.get_expon_lr_func, opt=opt, torch=torch)
    exec(compile(sched[0].strip(), "train.py:69", "exec"), ns)
    its = np.array([1, 100, 1000, 7000, 15000, 29999, 30000])
    out = dict(weight_init=np.array(opt.depth_l1_weight_init), weight_final=np.array(opt.depth_l1_weight_final),
               iterations=np.array(opt.iterations), weight_at=its, weight=np.array([ns["depth_l1_weight"](int(i)) for i in its]))

    class OnCpu:   # viewpoint_cam.invdepthmap.cuda() / .depth_mask.cuda()
        def __init__(self, t):
            self.t = t

        def cuda(self):
            return self.t
    g = torch.Generator().manual_seed(20261005)
    code = compile(block, "train.py:204-216", "exec")
    for tag, (H, W), it, reliable in (("a", (40, 56), 1, True), ("b", (75, 141), 7000, True), ("c", (32, 32), 100, False)):
        inv = (torch.rand((1, H, W), generator=g) * 0.8).requires_grad_(True)
        mono = torch.rand((1, H, W), generator=g) * 0.8
        mono[:, 3:9, 5:20] = inv.detach()[:, 3:9, 5:20]           # exact ties: sign(0) = 0
        mask = (torch.rand((1, H, W), generator=g) > 0.25).float()
        mask[:, : H // 4] *= 0.5                                   # the mask is a weight, not only 0 / 1 (cameras.py:63)
        cam = types.SimpleNamespace(depth_reliable=reliable, invdepthmap=OnCpu(mono), depth_mask=OnCpu(mask))
        ns2 = dict(ns, iteration=it, viewpoint_cam=cam, render_pkg={"depth": inv}, loss=torch.zeros(()), torch=torch)
        exec(code, ns2)
        loss = ns2["loss"]
        if loss.requires_grad:
            loss.backward()
        out["invdepth_" + tag], out["mono_" + tag], out["mask_" + tag] = inv.detach().numpy(), mono.numpy(), mask.numpy()
        out["iteration_" + tag], out["reliable_" + tag] = np.array(it), np.array(reliable)
        out["pure_" + tag] = np.array(float(ns2["Ll1depth_pure"]))
        out["loss_" + tag] = np.array(float(loss))
        out["logged_" + tag] = np.array(float(ns2["Ll1depth"]))
        out["grad_" + tag] = np.zeros_like(out["invdepth_" + tag]) if inv.grad is None else inv.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "depth_reg.npz"), **out)


if __name__ == "__main__":
    gen_depth_reg()
    gen_lgdwt_loss()
    gen_sh()
    gen_cameras()
    gen_losses()
    gen_schedule()
    gen_colmap()
    gen_nir_loss()
    gen_geometry()
    print("golden fixtures written to", HERE)
