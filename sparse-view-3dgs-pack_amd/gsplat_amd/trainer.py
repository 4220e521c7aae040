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

FIELDS = (("xyz", 3), ("f_dc", 3), ("f_rest", 45), ("opacity", 1), ("scaling", 3), ("rotation", 4))
FLOATS_PER_GAUSSIAN = sum(n for _, n in FIELDS)  # 59


class GaussianModelLite:
    """Raw (pre-activation) parameters of P Gaussians at max SH degree 3."""

    def __init__(self, scene, device, spatial_lr_scale=1.0):
        """scene: dict of ACTIVATED tensors as produced by gsplat_amd.synthetic (means3D, scales,
        rotations, opacities, shs[P,16,3]) - converted back to raw form as create_from_pcd would hold them."""
        P = scene["means3D"].shape[0]
        self.P = P
        self.device = device
        self.max_sh_degree = 3
        self.active_sh_degree = scene.get("sh_degree", 3)
        self.flat = torch.zeros((P * FLOATS_PER_GAUSSIAN,), dtype=torch.float32, device=device)
        self.flat_grad = torch.zeros_like(self.flat)
        self.params = {}
        off = 0
        shapes = {"xyz": (P, 3), "f_dc": (P, 1, 3), "f_rest": (P, 15, 3), "opacity": (P, 1), "scaling": (P, 3),
                  "rotation": (P, 4)}
        for name, n in FIELDS:
            view = self.flat[off:off + P * n].view(shapes[name])
            p = torch.nn.Parameter(view, requires_grad=True)
            p.grad = self.flat_grad[off:off + P * n].view(shapes[name])
            self.params[name] = p
            off += P * n
        with torch.no_grad():
            self.params["xyz"].copy_(scene["means3D"])
            self.params["f_dc"].copy_(scene["shs"][:, 0:1, :])
            self.params["f_rest"].copy_(scene["shs"][:, 1:, :])
            op = scene["opacities"].clamp(1e-6, 1 - 1e-6)
            self.params["opacity"].copy_(torch.log(op / (1 - op)))
            self.params["scaling"].copy_(torch.log(scene["scales"]))
            self.params["rotation"].copy_(scene["rotations"])
        lrs = {"xyz": 0.00016 * spatial_lr_scale, "f_dc": 0.0025, "f_rest": 0.0025 / 20.0, "opacity": 0.025,
               "scaling": 0.005, "rotation": 0.001}
        groups = [{"params": [self.params[n]], "lr": lrs[n], "name": n} for n, _ in FIELDS]
        # fused=True: one multi-tensor kernel per step instead of ~10 foreach passes over the 59 floats/Gaussian
        self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15, fused=(device.type == "cuda"))
        self.xyz_gradient_accum = torch.zeros((P, 1), device=device)
        self.denom = torch.zeros((P, 1), device=device)
        self.max_radii2D = torch.zeros((P,), device=device)

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
        return torch.cat((self.params["f_dc"], self.params["f_rest"]), dim=1)

    def zero_grad(self):
        # gradients live in one persistent flat buffer (views), so "set_to_none" is replaced by a memset
        self.flat_grad.zero_()

    def add_densification_stats(self, viewspace_grad, visible_mask):
        # gaussian_model.py:471-473, written without boolean indexing (no host sync); rows of culled
        # Gaussians have zero gradient and zero mask, so the result is identical
        vm = visible_mask.to(torch.float32).unsqueeze(-1)
        self.xyz_gradient_accum += torch.norm(viewspace_grad[:, :2], dim=-1, keepdim=True) * vm
        self.denom += vm


def render(viewpoint_camera, pc, Rasterizer, Settings, bg_color, scaling_modifier=1.0, antialiasing=False,
           debug=False, filter_as_indices=True):
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
        pkg = render(self.cameras[ci], m, self.Rasterizer, self.Settings, self.bg, filter_as_indices=False)
        mask = None if self.masks is None else self.masks[ci]
        loss, parts = self.criterion(pkg["render"], self.gts[ci], mask=mask)
        loss.backward()
        radii = pkg["radii"]
        with torch.no_grad():
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
