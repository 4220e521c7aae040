"""Build-owned counterpart of the reference's training step (the reference's Python never travels).

  GaussianModel.get_*            LGDWT-GS/scene/gaussian_model.py:40-60,102-135  (activations)
  training_setup / Adam groups   LGDWT-GS/scene/gaussian_model.py:178-201        (per-group LRs, eps 1e-15)
  render()                       LGDWT-GS/gaussian_renderer/__init__.py:18-128   (argument / return contract)
  one iteration                  LGDWT-GS/train.py:97-218,279-288                 (render, losses, backward, Adam)

MI355X-first differences:
  * the six parameter tensors are views into ONE flat fp32 buffer and their gradients views into one
    flat gradient buffer, so that the data-parallel exchange is a single RCCL all-reduce of 59 floats per
    Gaussian with no packing copies;
  * cameras are sharded over ranks (one camera per GPU per step); every rank holds a full replica and
    applies the identical Adam step after the all-reduce, so replicas stay bit-identical;
  * densification statistics (xyz_gradient_accum, denom: sum; max_radii2D: max) are reduced too
    (gaussian_model.py:471-473, train.py:266-268).
"""
import math

import torch
import torch.distributed as dist

# flat layout: one segment per field; the SH block keeps DC and rest interleaved [P,16,3] exactly as the
# rasterizer reads it, so get_features is a view (the reference concatenates _features_dc/_features_rest
# every step: gaussian_model.py:119-123).  The two learning rates of that block alternate with period 48.
FIELDS = (("xyz", 3), ("features", 48), ("opacity", 1), ("scaling", 3), ("rotation", 4))
FLOATS_PER_GAUSSIAN = sum(n for _, n in FIELDS)  # 59
# LGDWT-GS/arguments/__init__.py:79-86 (position_lr_init, feature_lr, opacity_lr, scaling_lr, rotation_lr)
LRS = {"xyz": 0.00016, "f_dc": 0.0025, "f_rest": 0.0025 / 20.0, "opacity": 0.025, "scaling": 0.005, "rotation": 0.001}


def expon_lr(step, lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """get_expon_lr_func of LGDWT-GS/utils/general_utils.py:29-62 (pinned by tests/golden/schedule.npz)."""
    import numpy as np
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    if lr_delay_steps > 0:
        delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
    else:
        delay_rate = 1.0
    t = np.clip(step / max_steps, 0, 1)
    log_lerp = np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)
    return float(delay_rate * log_lerp)


class FlatAdam:
    """torch.optim.Adam(lr=0.0, eps=1e-15) with the reference's per-group learning rates
    (gaussian_model.py:183-193), as ONE fused pass over the flat buffers through gs_adam_step."""

    def __init__(self, api, model, betas=(0.9, 0.999), eps=1e-15):
        from .capi import GsAdamSeg
        self.api, self.model, self.betas, self.eps = api, model, betas, eps
        self.exp_avg = torch.zeros_like(model.flat)
        self.exp_avg_sq = torch.zeros_like(model.flat)
        self.t = 0
        self.lr = dict(LRS)
        self.lr["xyz"] = LRS["xyz"] * model.spatial_lr_scale
        self._Seg = GsAdamSeg

    def segments(self):
        P = self.model.P
        segs = (self._Seg * len(FIELDS))()
        off = 0
        for k, (name, n) in enumerate(FIELDS):
            segs[k].begin, segs[k].end = off, off + P * n
            if name == "features":
                segs[k].lr_a, segs[k].lr_b, segs[k].period, segs[k].split = self.lr["f_dc"], self.lr["f_rest"], 48, 3
            else:
                segs[k].lr_a, segs[k].lr_b, segs[k].period, segs[k].split = self.lr[name], 0.0, 0, 0
            off += P * n
        return segs

    def step(self):
        import ctypes as C
        m = self.model
        self.t += 1
        segs = self.segments()
        stream = C.c_void_p(torch.cuda.current_stream(m.flat.device).cuda_stream) if m.flat.is_cuda else None
        self.api.call("adam_step", m.flat.data_ptr(), m.flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                      self.exp_avg_sq.data_ptr(), m.flat.numel(), segs, len(FIELDS), self.betas[0], self.betas[1],
                      self.eps, self.t, stream)


class GaussianModelLite:
    """Raw (pre-activation) parameters of P Gaussians at max SH degree 3."""

    def __init__(self, scene, device, spatial_lr_scale=1.0, api=None):
        """scene: dict of ACTIVATED tensors as produced by gsplat_amd.synthetic (means3D, scales,
        rotations, opacities, shs[P,16,3]) - converted back to raw form as create_from_pcd would hold them.
        api: C-ABI implementation that provides adam_step (None: torch.optim.Adam on the same flat buffers)."""
        P = scene["means3D"].shape[0]
        self.P = P
        self.device = device
        self.spatial_lr_scale = spatial_lr_scale
        self.max_sh_degree = 3
        self.active_sh_degree = scene.get("sh_degree", 3)
        self.flat = torch.zeros((P * FLOATS_PER_GAUSSIAN,), dtype=torch.float32, device=device)
        self.flat_grad = torch.zeros_like(self.flat)
        self.params = {}
        off = 0
        shapes = {"xyz": (P, 3), "features": (P, 16, 3), "opacity": (P, 1), "scaling": (P, 3), "rotation": (P, 4)}
        for name, n in FIELDS:
            view = self.flat[off:off + P * n].view(shapes[name])
            p = torch.nn.Parameter(view, requires_grad=True)
            p.grad = self.flat_grad[off:off + P * n].view(shapes[name])
            self.params[name] = p
            off += P * n
        with torch.no_grad():
            self.params["xyz"].copy_(scene["means3D"])
            self.params["features"].copy_(scene["shs"])
            op = scene["opacities"].clamp(1e-6, 1 - 1e-6)
            self.params["opacity"].copy_(torch.log(op / (1 - op)))
            self.params["scaling"].copy_(torch.log(scene["scales"]))
            self.params["rotation"].copy_(scene["rotations"])
        if api is not None:
            self.optimizer = FlatAdam(api, self)
        else:
            # plain torch Adam on the same storage (used to pin FlatAdam): f_dc / f_rest as strided views
            f = self.params["features"]
            groups = [{"params": [self.params["xyz"]], "lr": LRS["xyz"] * spatial_lr_scale, "name": "xyz"},
                      {"params": [self.params["opacity"]], "lr": LRS["opacity"], "name": "opacity"},
                      {"params": [self.params["scaling"]], "lr": LRS["scaling"], "name": "scaling"},
                      {"params": [self.params["rotation"]], "lr": LRS["rotation"], "name": "rotation"}]
            self.optimizer = _TorchAdamWithFeatureSplit(groups, f, LRS["f_dc"], LRS["f_rest"])
        self.xyz_gradient_accum = torch.zeros((P, 1), device=device)
        self.denom = torch.zeros((P, 1), device=device)
        self.max_radii2D = torch.zeros((P,), device=device)

    def update_learning_rate(self, iteration, position_lr_final=0.0000016, delay_mult=0.01, max_steps=30000):
        """gaussian_model.py:213-223: exponential decay of the xyz learning rate."""
        lr = expon_lr(iteration, LRS["xyz"] * self.spatial_lr_scale, position_lr_final * self.spatial_lr_scale,
                      lr_delay_mult=delay_mult, max_steps=max_steps)
        if isinstance(self.optimizer, FlatAdam):
            self.optimizer.lr["xyz"] = lr
        else:
            self.optimizer.set_xyz_lr(lr)
        return lr

    # activations, gaussian_model.py:102-135
    @property
    def get_xyz(self):
        return self.params["xyz"]

    @property
    def get_scaling(self):
        return torch.exp(self.params["scaling"])

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self.params["rotation"])

    @property
    def get_opacity(self):
        return torch.sigmoid(self.params["opacity"])

    @property
    def get_features(self):
        return self.params["features"]

    def zero_grad(self):
        """set_to_none, as the reference does (train.py:283-288): autograd then ASSIGNS the new gradients
        instead of adding them to a zeroed buffer (no 236 B/Gaussian memset, no read-modify-write)."""
        for p in self.params.values():
            p.grad = None

    def grad_views(self):
        P, off, out = self.P, 0, {}
        shapes = {"xyz": (P, 3), "features": (P, 16, 3), "opacity": (P, 1), "scaling": (P, 3), "rotation": (P, 4)}
        for name, n in FIELDS:
            out[name] = self.flat_grad[off:off + P * n].view(shapes[name])
            off += P * n
        return out

    def arm_grad_arena(self, backend):
        """Ask the rasterizer backward to write dL_dmeans3D / dL_dsh (51 of the 59 floats per Gaussian) straight
        into the flat gradient buffer: those two parameters reach the rasterizer without an activation in
        between, so autograd hands the very same tensors to the leaves and nothing is copied."""
        v = self.grad_views()
        backend.grad_arena = {"means3D": v["xyz"], "sh": v["features"]}

    def collect_grads(self):
        """After backward: make every .grad a view of the flat buffer (copy only what autograd produced elsewhere:
        the 8 floats per Gaussian behind exp / sigmoid / normalize, or everything when no arena was armed)."""
        for name, view in self.grad_views().items():
            p = self.params[name]
            g = p.grad
            if g is None:
                view.zero_()
            elif g.data_ptr() != view.data_ptr():
                view.copy_(g)
            p.grad = view

    def add_densification_stats(self, viewspace_grad, visible_mask):
        # gaussian_model.py:471-473, written without boolean indexing (no host sync); rows of culled
        # Gaussians have zero gradient and zero mask, so the result is identical
        vm = visible_mask.to(torch.float32).unsqueeze(-1)
        self.xyz_gradient_accum += torch.norm(viewspace_grad[:, :2], dim=-1, keepdim=True) * vm
        self.denom += vm


class _TorchAdamWithFeatureSplit:
    """Reference optimiser on the flat storage: torch.optim.Adam for the four plain groups and for f_dc / f_rest
    as separate contiguous copies that are written back (the fused torch kernels need dense tensors)."""

    def __init__(self, groups, features, lr_dc, lr_rest):
        self.features = features
        self.f_dc = features.detach()[:, :1, :].clone().requires_grad_(True)
        self.f_rest = features.detach()[:, 1:, :].clone().requires_grad_(True)
        groups = groups + [{"params": [self.f_dc], "lr": lr_dc, "name": "f_dc"},
                           {"params": [self.f_rest], "lr": lr_rest, "name": "f_rest"}]
        self.opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15)

    def set_xyz_lr(self, lr):
        for g in self.opt.param_groups:
            if g["name"] == "xyz":
                g["lr"] = lr

    def step(self):
        self.f_dc.grad = self.features.grad[:, :1, :].contiguous()
        self.f_rest.grad = self.features.grad[:, 1:, :].contiguous()
        self.opt.step()
        with torch.no_grad():
            self.features[:, :1, :] = self.f_dc
            self.features[:, 1:, :] = self.f_rest


def render(viewpoint_camera, pc, Rasterizer, Settings, bg_color, scaling_modifier=1.0, antialiasing=False,
           debug=False, filter_as_indices=True, clamp=True):
    """= render() of LGDWT-GS/gaussian_renderer/__init__.py:18-128 (SH evaluated by the rasterizer, scale +
    rotation given, no exposure): returns {render, viewspace_points, visibility_filter, radii, depth}."""
    screenspace_points = torch.zeros_like(pc.get_xyz, dtype=pc.get_xyz.dtype, requires_grad=True) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    rs = Settings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=debug, antialiasing=antialiasing)
    rasterizer = Rasterizer(raster_settings=rs)
    rendered_image, radii, depth_image = rasterizer(
        means3D=pc.get_xyz, means2D=screenspace_points, shs=pc.get_features, colors_precomp=None,
        opacities=pc.get_opacity, scales=pc.get_scaling, rotations=pc.get_rotation, cov3D_precomp=None)
    if clamp:  # gaussian_renderer/__init__.py:119; the fused criterion applies (and differentiates) it itself
        rendered_image = rendered_image.clamp(0, 1)
    return {"render": rendered_image, "viewspace_points": screenspace_points,
            # the reference returns indices ((radii > 0).nonzero(): a host sync); the step loop asks for the
            # equivalent boolean mask instead and stays asynchronous
            "visibility_filter": (radii > 0).nonzero() if filter_as_indices else (radii > 0),
            "radii": radii, "depth": depth_image}


def camera_to(cam, device):
    return cam._replace(world_view_transform=cam.world_view_transform.to(device),
                        full_proj_transform=cam.full_proj_transform.to(device),
                        camera_center=cam.camera_center.to(device))


class Trainer:
    """One process per GPU; `step(k)` renders this rank's camera of global step k and updates the model."""

    def __init__(self, model, cameras, gt_images, criterion, Rasterizer, Settings, bg, rank=0, world_size=1,
                 optimizer_step=True, masks=None):
        self.model, self.cameras, self.gts, self.criterion = model, cameras, gt_images, criterion
        self.Rasterizer, self.Settings, self.bg = Rasterizer, Settings, bg
        self.rank, self.world_size = rank, world_size
        self.optimizer_step = optimizer_step
        self.masks = masks  # per-camera ELF patch masks (depend on the ground truth only): cached
        self.last = None

    def camera_index(self, k):
        return (k * self.world_size + self.rank) % len(self.cameras)

    def step(self, k):
        m = self.model
        ci = self.camera_index(k)
        m.zero_grad()
        backend = getattr(getattr(self.Rasterizer, "_fn", None), "_impl", None)
        backend = getattr(backend, "backend", None)
        if backend is not None:
            m.arm_grad_arena(backend)
        fused = getattr(self.criterion, "fused", False)
        pkg = render(self.cameras[ci], m, self.Rasterizer, self.Settings, self.bg, filter_as_indices=False,
                     clamp=not fused)
        mask = None if self.masks is None else self.masks[ci]
        if fused:
            loss, parts = self.criterion.fused_call(pkg["render"], self.gts[ci], mask=mask)
        else:
            loss, parts = self.criterion(pkg["render"], self.gts[ci], mask=mask)
        loss.backward()
        radii = pkg["radii"]
        with torch.no_grad():
            m.collect_grads()
            # train.py:268 (radii are 0 for culled Gaussians, so a plain maximum equals the masked update)
            torch.maximum(m.max_radii2D, radii.to(torch.float32), out=m.max_radii2D)
            m.add_densification_stats(pkg["viewspace_points"].grad, pkg["visibility_filter"])
            if self.world_size > 1:
                self.all_reduce()
            if self.optimizer_step:
                m.optimizer.step()
        self.last = dict(loss=loss.detach(), radii=radii, parts=parts)
        return loss.detach()

    def all_reduce(self):
        """The one exchange step of the data-parallel path: sum of the 59-floats-per-Gaussian gradient
        buffer (236 B x P) + densification statistics, RCCL over xGMI (gloo in the CPU tests)."""
        m = self.model
        work = [dist.all_reduce(m.flat_grad, op=dist.ReduceOp.SUM, async_op=True),
                dist.all_reduce(m.xyz_gradient_accum, op=dist.ReduceOp.SUM, async_op=True),
                dist.all_reduce(m.denom, op=dist.ReduceOp.SUM, async_op=True),
                dist.all_reduce(m.max_radii2D, op=dist.ReduceOp.MAX, async_op=True)]
        for w in work:
            w.wait()
