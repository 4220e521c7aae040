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


if __name__ == "__main__":
    gen_sh()
    gen_cameras()
    gen_losses()
    gen_schedule()
    print("golden fixtures written to", HERE)
