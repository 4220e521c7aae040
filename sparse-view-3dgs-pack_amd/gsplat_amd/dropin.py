"""The reference's own training iteration, written against the DROP-IN packages only - what a maintainer who follows
INTEGRATION.md section 1 gets without touching anything else.

Shape: LGDWT-GS/train.py:97-288 (render -> L1, global 2-level DWT, patch DWT on the ELF-selected patches, SSIM, the
running-mean DWT scale from a host `.item()`, backward, densification statistics, `optimizer.step()`), with the model's
six parameter tensors and per-group Adam of LGDWT-GS/scene/gaussian_model.py:40-60,178-201 and the renderer glue of
LGDWT-GS/gaussian_renderer/__init__.py:18-128.  Every operator goes through the names the reference imports:

    diff_gaussian_rasterization.GaussianRasterizer / GaussianRasterizationSettings
    lgdwt_loss.l1_loss / get_dwt_subbands / compute_elf_map / compute_patch_dwt_loss      (= utils/loss_utils.py)
    fused_ssim.fused_ssim
    torch.optim.Adam(lr=0.0, eps=1e-15) over six groups   - or gsplat_amd.optim.FusedAdam, the same update in one kernel

None of the build's own step machinery (gsplat_amd.trainer) is used.  bench.py times this loop as its `drop_in_api` leg.
"""
import math

import torch

C0 = 0.28209479177387814


class DropInModel:
    """Six leaf tensors as GaussianModel keeps them (gaussian_model.py:40-60) + the activations of :102-135."""

    def __init__(self, scene, device):
        def leaf(t):
            return torch.nn.Parameter(t.to(device).contiguous().requires_grad_(True))
        self._xyz = leaf(scene["means3D"])
        self._features_dc = leaf(scene["shs"][:, :1, :])
        self._features_rest = leaf(scene["shs"][:, 1:, :])
        op = scene["opacities"].clamp(1e-6, 1 - 1e-6)
        self._opacity = leaf(torch.log(op / (1 - op)))
        self._scaling = leaf(torch.log(scene["scales"]))
        self._rotation = leaf(scene["rotations"])
        self.active_sh_degree = int(scene.get("sh_degree", 3))
        P = self._xyz.shape[0]
        self.max_radii2D = torch.zeros((P,), device=device)
        self.xyz_gradient_accum = torch.zeros((P, 1), device=device)
        self.denom = torch.zeros((P, 1), device=device)

    def groups(self, spatial_lr_scale=1.0):
        """gaussian_model.py:183-190"""
        return [{"params": [self._xyz], "lr": 0.00016 * spatial_lr_scale, "name": "xyz"},
                {"params": [self._features_dc], "lr": 0.0025, "name": "f_dc"},
                {"params": [self._features_rest], "lr": 0.0025 / 20.0, "name": "f_rest"},
                {"params": [self._opacity], "lr": 0.025, "name": "opacity"},
                {"params": [self._scaling], "lr": 0.005, "name": "scaling"},
                {"params": [self._rotation], "lr": 0.001, "name": "rotation"}]

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        """gaussian_model.py:471-473"""
        self.xyz_gradient_accum[update_filter] += torch.norm(viewspace_point_tensor.grad[update_filter, :2], dim=-1, keepdim=True)
        self.denom[update_filter] += 1


def render(cam, pc, bg_color, camera_key=None):
    """gaussian_renderer/__init__.py:18-128 (SH evaluated by the rasterizer, scales + rotations given, no exposure)."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    screenspace_points = torch.zeros_like(pc._xyz, dtype=pc._xyz.dtype, requires_grad=True, device=pc._xyz.device) + 0
    screenspace_points.retain_grad()
    settings = GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=math.tan(cam.FoVx * 0.5),
        tanfovy=math.tan(cam.FoVy * 0.5), bg=bg_color, scale_modifier=1.0, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=pc.active_sh_degree, campos=cam.camera_center, prefiltered=False,
        debug=False, antialiasing=False)
    rasterizer = GaussianRasterizer(raster_settings=settings)
    if camera_key is not None:          # the one optional addition: who this camera is
        rasterizer.camera_key = camera_key
    shs = torch.cat((pc._features_dc, pc._features_rest), dim=1)
    rendered_image, radii, depth_image = rasterizer(
        means3D=pc._xyz, means2D=screenspace_points, shs=shs, colors_precomp=None, opacities=torch.sigmoid(pc._opacity),
        scales=torch.exp(pc._scaling), rotations=torch.nn.functional.normalize(pc._rotation), cov3D_precomp=None)
    rendered_image = rendered_image.clamp(0, 1)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": (radii > 0).nonzero(),
            "radii": radii, "depth": depth_image}


class _Pipe:   # arguments/__init__.py PipelineParams defaults
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False
    antialiasing = False


_PIPE = _Pipe()


class DropInLoop:
    """One object = one training run's state (model, optimizer, the running mean of train.py:75)."""

    def __init__(self, scene, cameras, gt_images, device, dwt=True, patch=True, optimizer="torch", use_camera_key=False,
                 fused_criterion=False, raw_render=False):
        self.pc = DropInModel(scene, device)
        self.cameras, self.gts = cameras, gt_images
        self.bg = torch.zeros(3, device=device)
        self.dwt, self.patch = dwt, patch
        self.dwt_running_mean = 1.0
        self.lambda_dssim, self.patch_dwt_weight = 0.2, 0.1
        self.dwt_weights = {"LL1": 1.0, "LH1": 1.0, "HL1": 1.0, "HH1": 0.0, "LL2": 0.0, "LH2": 0.0, "HL2": 0.0, "HH2": 0.0}
        self.use_camera_key = use_camera_key
        # `from gsplat_amd.render_raw import render` instead of the reference's gaussian_renderer.render
        # (same signature and dict; the model's six raw tensors go to the library, no activation / cat kernels)
        self.raw_render = raw_render
        # INTEGRATION.md section 1, "optional, faster": the loss calls of train.py:128-202 replaced by ONE call of
        # lgdwt_loss.criterion() (same terms, the running mean kept on the device: no `.item()`), ELF masks cached per camera
        self.criterion, self.masks = None, {}
        if fused_criterion:
            import lgdwt_loss
            self.criterion = lgdwt_loss.criterion(dwt_enable=dwt, patch_dwt_enable=patch)
        if optimizer == "torch":
            self.optimizer = torch.optim.Adam(self.pc.groups(), lr=0.0, eps=1e-15)
        else:
            from .optim import FusedAdam
            self.optimizer = FusedAdam(self.pc.groups(), lr=0.0, eps=1e-15)

    def iteration(self, ci):
        """train.py:119-288 for camera `ci` (no densification, no logging); returns the loss as a Python float like the
        reference's progress bar does (`loss.item()`, train.py:224)."""
        from fused_ssim import fused_ssim
        from lgdwt_loss import compute_elf_map, compute_patch_dwt_loss, get_dwt_subbands, l1_loss
        pc = self.pc
        key = ("dropin", ci) if self.use_camera_key else None
        if self.raw_render:
            from .render_raw import render as render_raw
            pkg = render_raw(self.cameras[ci], pc, _PIPE, self.bg, camera_key=key)
        else:
            pkg = render(self.cameras[ci], pc, self.bg, camera_key=key)
        image, vsp, vis, radii = pkg["render"], pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"]
        gt = self.gts[ci]
        if self.criterion is not None:
            if self.patch and ci not in self.masks:
                self.masks[ci] = self.criterion.elf_mask(gt)
            if "render_unclamped" in pkg:   # render_raw: the whole criterion as ONE autograd node on the un-clamped render
                loss, _ = self.criterion.fused_call(pkg["render_unclamped"], gt, mask=self.masks.get(ci))
            else:
                loss, _ = self.criterion(image, gt, mask=self.masks.get(ci))
            return self._finish(loss, vsp, vis, radii)
        Ll1 = l1_loss(image, gt)
        dwt_loss = torch.tensor(0.0, device=image.device)
        if self.dwt:
            pb, gb = get_dwt_subbands(image.unsqueeze(0)), get_dwt_subbands(gt.unsqueeze(0))
            total = 0.0
            for name, w in self.dwt_weights.items():
                if w != 0.0:
                    total = total + w * l1_loss(pb[name], gb[name])
            dwt_loss = total
        patch_loss = torch.tensor(0.0, device=image.device)
        if self.patch:
            elf = compute_elf_map(gt.unsqueeze(0))
            patch_loss = compute_patch_dwt_loss(image.unsqueeze(0), gt.unsqueeze(0), elf, patch_size=128, percentile=0.2,
                                                lh1_weight=1.0, hl1_weight=1.0)
        ssim_value = fused_ssim(image.unsqueeze(0), gt.unsqueeze(0))
        base = (1.0 - self.lambda_dssim) * Ll1 + self.lambda_dssim * (1.0 - ssim_value)
        if self.dwt:
            ratio = (base.detach() / (dwt_loss.detach() + 1e-8)).item()        # the reference's host sync (train.py:191)
            self.dwt_running_mean = 0.95 * self.dwt_running_mean + 0.05 * ratio
            loss = base + float(max(0.1, min(10.0, self.dwt_running_mean))) * dwt_loss
        else:
            loss = base
        if self.patch:
            loss = loss + self.patch_dwt_weight * patch_loss
        return self._finish(loss, vsp, vis, radii)

    def _finish(self, loss, vsp, vis, radii):
        pc = self.pc
        loss.backward()
        with torch.no_grad():
            value = loss.item()     # train.py:224, the progress bar: the host waits for the backward HERE, before the statistics
            # train.py:266-268      # and the optimizer are enqueued - they run while the host sets up the next iteration
            vf = vis.squeeze(1)
            pc.max_radii2D[vf] = torch.max(pc.max_radii2D[vf], radii[vf].to(torch.float32))
            pc.add_densification_stats(vsp, vf)
            self.optimizer.step()   # train.py:284-285
            self.optimizer.zero_grad(set_to_none=True)
        return value
