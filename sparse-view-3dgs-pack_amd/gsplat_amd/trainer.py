"""Build-owned counterpart of the reference's training step (the reference's Python never travels).

  GaussianModel.get_*            LGDWT-GS/scene/gaussian_model.py:40-60,102-135  (activations)
  training_setup / Adam groups   LGDWT-GS/scene/gaussian_model.py:178-201        (per-group LRs, eps 1e-15)
  render()                       LGDWT-GS/gaussian_renderer/__init__.py:18-128   (argument / return contract)
  one iteration                  LGDWT-GS/train.py:97-218,279-288                 (render, losses, backward, Adam)
  densify / prune / reset        LGDWT-GS/scene/gaussian_model.py:258-261,316-473 (clone, split, prune, moments)
  schedule                       LGDWT-GS/train.py:99-111,262-288, arguments/__init__.py:78-100

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
LRS = {"xyz": 0.00016, "f_dc": 0.0025, "f_rest": 0.0025 / 20.0, "opacity": 0.025, "scaling": 0.005, "rotation": 0.001,
       # multispectral variant (mult-dwtgs/scene/gaussian_model.py:266-280): albedo at position_lr_init (unscaled,
       # no schedule), the global gain at feature_lr
       "nir_albedo": 0.00016, "nir_gain": 0.0025}
NIR_FIELD = ("nir_albedo", 1)
SHARD_UNIT = 4 * 840  # float4 x lcm(1..8)


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
        self.alloc_moments()
        self.t = 0
        # torch keeps Adam's step count per parameter; a group that is skipped (no gradient after its tensor
        # was replaced) falls behind the others
        self.seg_steps = {name: 0 for name, _ in model.fields}
        if getattr(model, "with_nir", False):
            self.seg_steps["nir_gain"] = 0   # the global gain (not a per-Gaussian field): stepped with every optimizer step
        self.lr = dict(LRS)
        self.lr["xyz"] = LRS["xyz"] * model.spatial_lr_scale
        self._Seg = GsAdamSeg
        # optimizer_type = "sparse_adam" (arguments/__init__.py:101, train.py:282-284: `visible = radii > 0;
        # optimizer.step(visible, N)`): only the Gaussians visible in the step's view(s) are stepped; parameters and moments of
        # the others keep their bits.  (The optimizer class the reference would use, SparseGaussianAdam, lives in the 3dgs_accel
        # branch of the rasterizer, which the reference does not vendor - its pinned branch is dr_aa; the update applied to a
        # visible row here is torch.optim.Adam's, bias correction by the group's step count included.)
        self.sparse = False

    def alloc_moments(self):
        """Zeroed moment buffers for the model's current size, padded like the parameters (the sharded optimizer
        all-gathers them in place before a re-layout or a checkpoint)."""
        m = self.model
        self.exp_avg_padded = torch.zeros_like(m.flat_padded)
        self.exp_avg_sq_padded = torch.zeros_like(m.flat_padded)
        self.exp_avg = self.exp_avg_padded[:m.flat.numel()]
        self.exp_avg_sq = self.exp_avg_sq_padded[:m.flat.numel()]
        self._dormant, self._dormant_valid = None, False

    # Dormant blocks (GsStepState.dormant, include/gsplat.h): one byte per 256 consecutive Gaussians, 1 = every Adam moment
    # of every row of the block is +0, so the zero-gradient update is a no-op and gs_backward_step / gs_step_uninstanced
    # skip the block's parameter and moment traffic.  DERIVED from the moments (never assumed), kept current by the fused
    # kernels (a block that receives a gradient is cleared on the device) and recomputed after anything else wrote the
    # moments (invalidate_dormant).  Pays when the rows are in spatial order (GaussianModelLite(spatial_order=True)): the
    # Gaussians no camera reaches - more than half of bench.py's scene - then fill whole blocks.  GS_DORMANT_BLOCKS=0: off.
    USE_DORMANT = __import__("os").environ.get("GS_DORMANT_BLOCKS", "1") != "0"
    DORMANT_BLOCK = 256

    def invalidate_dormant(self):
        self._dormant_valid = False

    def dormant_flags(self):
        """The current flags (uint8 [ceil(P / 256)], same tensor until the model is re-laid out), recomputed from the moment
        buffers when something other than the fused step may have written them."""
        m = self.model
        B = self.DORMANT_BLOCK
        nb = (m.P + B - 1) // B
        if getattr(self, "_dormant", None) is None or self._dormant.numel() != nb or self._dormant.device != m.flat.device:
            self._dormant = torch.zeros((nb,), dtype=torch.uint8, device=m.flat.device)
            self._dormant_valid = False
        if not self._dormant_valid:
            with torch.no_grad():
                live = torch.zeros((nb * B,), dtype=torch.bool, device=m.flat.device)
                for buf in (self.exp_avg, self.exp_avg_sq):
                    for name, view in self.field_views(buf).items():
                        live[:m.P] |= (view.view(torch.int32) != 0).any(dim=1)   # (bit patterns: -0.0 is not +0)
                self._dormant.copy_((~live.view(nb, B).any(dim=1)).to(torch.uint8))
            self._dormant_valid = True
        return self._dormant

    def segments(self, skip=()):
        P = self.model.P
        names = [(name, n) for name, n in self.model.fields]
        segs = (self._Seg * len(names))()
        off, k = 0, 0
        for name, n in names:
            if name not in skip:
                segs[k].begin, segs[k].end = off, off + P * n
                if name == "features":
                    segs[k].lr_a, segs[k].lr_b, segs[k].period, segs[k].split = self.lr["f_dc"], self.lr["f_rest"], 48, 3
                else:
                    segs[k].lr_a, segs[k].lr_b, segs[k].period, segs[k].split = self.lr[name], 0.0, 0, 0
                segs[k].step = self.seg_steps[name]
                segs[k].row_width = n
                k += 1
            off += P * n
        return segs, k

    def step(self, skip=(), row_mask=None):
        """skip: fields without a gradient this iteration (their tensor was replaced after backward: torch's
        optimizer.step() leaves such a parameter, its moments and its step count alone).
        row_mask (sparse_adam): float [P], > 0 = the Gaussian is stepped."""
        self.begin_step(skip)
        self.step_range(0, self.model.flat.numel(), skip, row_mask=row_mask)

    def begin_step(self, skip=()):
        self.t += 1
        for name, _ in self.model.fields:
            if name not in skip:
                self.seg_steps[name] += 1
        if "nir_gain" in self.seg_steps:
            self.seg_steps["nir_gain"] += 1

    def _nir_into(self, st, grads_out):
        """The 4th channel's part of a GsStepState (multispectral model): raw albedo row, global gain, moments, rates."""
        m = self.model
        if not m.with_nir:
            return
        st.extra = m.params["nir_albedo"].data_ptr()
        st.gain = m.nir_gain.data_ptr()
        st.lr_extra, st.lr_gain = self.lr["nir_albedo"], self.lr["nir_gain"]
        if grads_out:
            st.grad_out_extra = m.grad_views()["nir_albedo"].data_ptr()
            st.grad_out_gain = m.gain_grad_word.data_ptr()
            st.step_extra = st.step_gain = 1
            return
        ea, eas = self.field_views(self.exp_avg), self.field_views(self.exp_avg_sq)
        st.extra_m, st.extra_v = ea["nir_albedo"].data_ptr(), eas["nir_albedo"].data_ptr()
        gs = m.nir_gain_optimizer.state[m.nir_gain]
        st.gain_m, st.gain_v = gs["exp_avg"].data_ptr(), gs["exp_avg_sq"].data_ptr()
        st.step_extra = self.seg_steps["nir_albedo"] if self.seg_steps["nir_albedo"] > 0 else self.t
        st.step_gain = self.seg_steps["nir_gain"]

    def grads_out_request(self):
        """The GsStepState of the data-parallel form of gs_backward_step: raw rows in, gradients OUT into the flat gradient
        buffer (nothing to pack before the exchange), this view's statistic increments into `stat_delta`, the validity
        flag into the exchange buffer's tail.  Counters are not touched (exchange_and_step advances them)."""
        from .capi import GsStepState
        m = self.model
        st = GsStepState()
        p, gv = self.field_views(m.flat), m.grad_views()
        for k, name in enumerate(self.ROWS):
            setattr(st, name, p[name].data_ptr())
            st.grad_out[k] = gv[name].data_ptr()
            st.step[k] = 1
        st.beta1, st.beta2, st.eps = self.betas[0], self.betas[1], self.eps
        st.max_radii2D = m.max_radii2D.data_ptr()
        st.xyz_gradient_accum, st.denom = m.stat_delta[0].data_ptr(), m.stat_delta[1].data_ptr()
        st.fail_flag = m.fail_flag.data_ptr()
        st.grad_mask = m.grad_mask.data_ptr()
        self._nir_into(st, True)
        return st

    def step_range(self, lo, hi, skip=(), grads=None, gate=None, row_mask=None):
        """The update of elements [lo, hi) of the flat buffers (lo a multiple of 4): the data-parallel step applies
        Adam chunk by chunk as the chunks of the gradient all-reduce arrive.  The segment table is shifted by -lo so
        that the kernel's element index i stands for element lo + i (a negative `begin` keeps the phase of the
        interleaved SH learning rates)."""
        import ctypes as C
        m = self.model
        segs, nseg = self.segments(skip)
        if nseg == 0 or hi <= lo:
            return
        self._dormant_valid = False   # (gs_adam_step writes moments without keeping the dormant-block flags)
        for k in range(nseg):
            segs[k].begin -= lo
            segs[k].end -= lo
        stream = C.c_void_p(torch.cuda.current_stream(m.flat.device).cuda_stream) if m.flat.is_cuda else None
        # grads: a tensor holding the gradients of elements [lo, hi) (default: that slice of the flat gradient buffer)
        gptr = m.flat_grad.data_ptr() + 4 * lo if grads is None else grads.data_ptr()
        if self.sparse and row_mask is None:
            raise RuntimeError("sparse_adam: the step needs the view's visibility (row_mask)")
        if row_mask is not None:   # (sparse_adam, or one part of a step that is applied in two parts: the sparse exchange)
            self.api.call("adam_step_masked", m.flat.data_ptr() + 4 * lo, gptr,
                          self.exp_avg.data_ptr() + 4 * lo, self.exp_avg_sq.data_ptr() + 4 * lo, hi - lo, segs, nseg,
                          self.betas[0], self.betas[1], self.eps, self.t, None if gate is None else gate.data_ptr(),
                          row_mask.data_ptr(), stream)
            return
        if gate is not None:  # a device float: the update is a no-op on the device when it is non-zero (gs_adam_step_gated)
            self.api.call("adam_step_gated", m.flat.data_ptr() + 4 * lo, gptr,
                          self.exp_avg.data_ptr() + 4 * lo, self.exp_avg_sq.data_ptr() + 4 * lo, hi - lo, segs, nseg,
                          self.betas[0], self.betas[1], self.eps, self.t, gate.data_ptr(), stream)
            return
        self.api.call("adam_step", m.flat.data_ptr() + 4 * lo, gptr,
                      self.exp_avg.data_ptr() + 4 * lo, self.exp_avg_sq.data_ptr() + 4 * lo, hi - lo, segs, nseg,
                      self.betas[0], self.betas[1], self.eps, self.t, stream)

    ROWS = ("xyz", "features", "opacity", "scaling", "rotation")
    LR_CLASSES = (("xyz", 0), ("f_dc", 1), ("f_rest", 1), ("opacity", 2), ("scaling", 3), ("rotation", 4))

    def step_coefficients(self, skip=()):
        """The 11 step-dependent constants of gs_backward_step for the CURRENT counters (call after begin_step):
        lr / (1 - beta1^t) per learning-rate class, 1 / sqrt(1 - beta2^t) per row - in the float arithmetic of
        gs_adam_step (double, rounded once; the product lr * 1/(1 - beta1^t) in float).  Always 15 floats: words 11..14 are
        the same two constants of the multispectral model's 4th-channel row and of its gain (zero without them)."""
        import numpy as np
        t = [self.seg_steps[name] if self.seg_steps[name] > 0 else self.t for name in self.ROWS]
        t = [max(x, 1) for x in t]
        b1, b2 = (float(np.float32(b)) for b in self.betas)  # the C ABI takes the betas as float
        out = np.zeros(15, dtype=np.float32)
        for c, (name, row) in enumerate(self.LR_CLASSES):
            out[c] = np.float32(self.lr[name]) * np.float32(1.0 / (1.0 - b1 ** t[row]))
        for k in range(5):
            out[6 + k] = np.float32(1.0 / math.sqrt(1.0 - b2 ** t[k]))
        if getattr(self.model, "with_nir", False):
            for k, name in enumerate(("nir_albedo", "nir_gain")):
                tk = max(self.seg_steps[name] if self.seg_steps[name] > 0 else self.t, 1)
                out[11 + 2 * k] = np.float32(self.lr[name]) * np.float32(1.0 / (1.0 - b1 ** tk))
                out[12 + 2 * k] = np.float32(1.0 / math.sqrt(1.0 - b2 ** tk))
        return out

    def fused_request(self, skip=(), coef_dev=None):
        """The GsStepState of ONE optimizer step (counters advanced as step() would) for gs_backward_step: raw parameter
        rows, both moment buffers, per-row step counts (0 = skipped), learning rates, view statistics.
        coef_dev: device tensor of 11 floats to read the step-dependent constants from (graph replay)."""
        from .capi import GsStepState
        m = self.model
        self.begin_step(skip)
        st = GsStepState()
        if coef_dev is not None:
            st.coef_dev = coef_dev.data_ptr()
        names = ("xyz", "features", "opacity", "scaling", "rotation")
        # (the eighteen row addresses only change when the buffers do: a re-layout, a restore - not every step)
        key = (m.flat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), m.P)
        ptrs = getattr(self, "_row_ptrs", None)
        if ptrs is None or ptrs[0] != key:
            p, ea, eas = self.field_views(m.flat), self.field_views(self.exp_avg), self.field_views(self.exp_avg_sq)
            ptrs = self._row_ptrs = (key, [(p[n].data_ptr(), ea[n].data_ptr(), eas[n].data_ptr()) for n in names])
        for k, name in enumerate(names):
            pp, pm, pv = ptrs[1][k]
            setattr(st, name, pp)
            st.m[k], st.v[k] = pm, pv
            st.step[k] = 0 if name in skip else (self.seg_steps[name] if self.seg_steps[name] > 0 else self.t)
        for k, name in enumerate(("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")):
            st.lr[k] = self.lr[name]
        st.beta1, st.beta2, st.eps = self.betas[0], self.betas[1], self.eps
        st.max_radii2D, st.xyz_gradient_accum, st.denom = (m.max_radii2D.data_ptr(), m.xyz_gradient_accum.data_ptr(),
                                                           m.denom.data_ptr())
        self._nir_into(st, False)
        if m.with_nir and "nir_albedo" in skip:
            st.step_extra = 0
        if self.USE_DORMANT and m.flat.is_cuda:
            st.dormant = self.dormant_flags().data_ptr()
        st.sparse = 1 if self.sparse else 0
        return st

    def field_views(self, buf):
        P, off, out = self.model.P, 0, {}
        for name, n in self.model.fields:
            out[name] = buf[off:off + P * n].view(P, n)
            off += P * n
        return out


def _stream_of(t):
    import ctypes as C
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream) if t.is_cuda else None


class _FusedActivations(torch.autograd.Function):
    """scales, rotations, opacities = exp(_scaling), normalize(_rotation), sigmoid(_opacity) in one kernel each way
    (gs_activations_fwd / gs_activations_bwd); the backward writes straight into the model's flat gradient buffer
    and hands those views to autograd, so nothing is copied afterwards."""

    @staticmethod
    def forward(ctx, scaling, rotation, opacity, model):
        api = model.optimizer.api
        P = scaling.shape[0]
        scales, rots, opac = torch.empty_like(scaling), torch.empty_like(rotation), torch.empty_like(opacity)
        api.call("activations_fwd", scaling.data_ptr(), rotation.data_ptr(), opacity.data_ptr(), P, scales.data_ptr(),
                 rots.data_ptr(), opac.data_ptr(), _stream_of(scaling))
        ctx.model = model
        ctx.save_for_backward(scaling, rotation, opacity)
        ctx.set_materialize_grads(False)
        return scales, rots, opac

    @staticmethod
    def backward(ctx, g_s, g_r, g_o):
        if g_s is None and g_r is None and g_o is None:
            # nothing flowed back (the fused train step has already consumed the gradients inside gs_backward_step)
            return None, None, None, None
        scaling, rotation, opacity = ctx.saved_tensors
        model = ctx.model
        P = scaling.shape[0]
        v = model.grad_views()
        g_s = torch.zeros_like(scaling) if g_s is None else g_s.contiguous()
        g_r = torch.zeros_like(rotation) if g_r is None else g_r.contiguous()
        g_o = torch.zeros_like(opacity) if g_o is None else g_o.contiguous()
        model.optimizer.api.call("activations_bwd", scaling.data_ptr(), rotation.data_ptr(), opacity.data_ptr(), P,
                                 g_s.data_ptr(), g_r.data_ptr(), g_o.data_ptr(), v["scaling"].data_ptr(),
                                 v["rotation"].data_ptr(), v["opacity"].data_ptr(), _stream_of(scaling))
        return v["scaling"], v["rotation"], v["opacity"], None


class GaussianModelLite:
    """Raw (pre-activation) parameters of P Gaussians at max SH degree 3."""

    SPATIAL_ORDER_MIN_P = 100_000

    def __init__(self, scene, device, spatial_lr_scale=1.0, api=None, with_nir=False, spatial_order=None):
        """scene: dict of ACTIVATED tensors as produced by gsplat_amd.synthetic (means3D, scales,
        rotations, opacities, shs[P,16,3]) - converted back to raw form as create_from_pcd would hold them.
        api: C-ABI implementation that provides adam_step (None: torch.optim.Adam on the same flat buffers).
        spatial_order: keep the rows in Morton order of the centres (here, and again after every densification): rows that
        are neighbours in memory are neighbours in space, so whole wavefronts / 256-row blocks are culled, depth-limited
        away, stepped or skipped (FlatAdam.dormant_flags) together.  The model is the same set of Gaussians; only which row
        holds which one differs from the reference's [survivors, clones, split samples] order.  Default (None): on from
        SPATIAL_ORDER_MIN_P = 100 000 Gaussians - below, a step is launch-bound and the row order buys nothing -;
        GS_SPATIAL_ORDER=0 / 1 forces it off / on for models that do not say."""
        if spatial_order is None:
            env = __import__("os").environ.get("GS_SPATIAL_ORDER", "")
            spatial_order = (env == "1") if env in ("0", "1") else int(scene["means3D"].shape[0]) >= self.SPATIAL_ORDER_MIN_P
        self.spatial_order = bool(spatial_order)
        if self.spatial_order:
            from . import synthetic
            scene = synthetic.spatially_ordered(scene)
        P = scene["means3D"].shape[0]
        self.P = P
        self.device = device
        self.spatial_lr_scale = spatial_lr_scale
        self.max_sh_degree = 3
        self.active_sh_degree = scene.get("sh_degree", 3)
        self.percent_dense = 0.01  # arguments/__init__.py:91
        # multispectral variant: one more per-Gaussian parameter (raw NIR albedo, sigmoid-activated) and a global gain
        # (mult-dwtgs/scene/gaussian_model.py:183-186,266-280,408-423).  The reference only creates them in load_ply
        # (SURVEY Q5): here they exist from the start, initialised as its load_ply does - albedo = the DC coefficient
        # it would broadcast (channel 0 is the one its render keeps), gain = 1.
        self.with_nir = bool(with_nir)
        self.fields = FIELDS + ((NIR_FIELD,) if self.with_nir else ())
        self.width = sum(n for _, n in self.fields)
        self.nir_gain = None
        self.exposure = None
        self.exposure_optimizer = None
        self._allocate(P)
        with torch.no_grad():
            self.params["xyz"].copy_(scene["means3D"])
            self.params["features"].copy_(scene["shs"])
            op = scene["opacities"].clamp(1e-6, 1 - 1e-6)
            self.params["opacity"].copy_(torch.log(op / (1 - op)))
            self.params["scaling"].copy_(torch.log(scene["scales"]))
            self.params["rotation"].copy_(scene["rotations"])
            if self.with_nir:
                self.params["nir_albedo"].copy_(scene["shs"][:, 0, 0:1])
        if self.with_nir:
            self.nir_gain = torch.nn.Parameter(torch.tensor(1.0, dtype=torch.float32, device=device))
            self.nir_gain_optimizer = torch.optim.Adam([self.nir_gain], lr=LRS["nir_gain"], eps=1e-15)
            # the moments exist from the start: the fused multispectral step (gs_backward_step_x) updates them in place, the
            # un-fused one goes through torch's optimizer - ONE state for both.  The step count lives with the main
            # optimizer's (FlatAdam.seg_steps["nir_gain"]): the gain steps whenever that one does.
            self.nir_gain_optimizer.state[self.nir_gain] = {
                "step": torch.tensor(0.0, dtype=torch.float32), "exp_avg": torch.zeros_like(self.nir_gain.data),
                "exp_avg_sq": torch.zeros_like(self.nir_gain.data)}
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

    def enable_exposure(self, n_cameras):
        """Per-camera exposure (LGDWT-GS/scene/gaussian_model.py:173-176,201: a [n_cams, 3, 4] parameter starting at
        [I | 0] with its own Adam; rate from get_expon_lr_func(0.01, 0.001, delay_steps 0, delay_mult 0) over the
        run, arguments/__init__.py:87-90, set by update_learning_rate).  Used by render(..., use_trained_exp=True)."""
        eye = torch.eye(3, 4, device=self.device)[None].repeat(int(n_cameras), 1, 1)
        self.exposure = torch.nn.Parameter(eye.requires_grad_(True))
        self.exposure_optimizer = torch.optim.Adam([self.exposure])

    def get_exposure(self, camera_index):
        return self.exposure[camera_index]

    @staticmethod
    def _shapes(P):
        return {"xyz": (P, 3), "features": (P, 16, 3), "opacity": (P, 1), "scaling": (P, 3), "rotation": (P, 4),
                "nir_albedo": (P, 1)}

    def _allocate(self, P):
        """(Re)create the flat parameter / gradient buffers for P Gaussians and the views into them."""
        # every re-allocation (densification, restore, load_ply) gives the model new buffers: whoever froze addresses of the
        # old ones (GraphedStep) compares this counter
        self.generation = getattr(self, "generation", 0) + 1
        self.P = P
        W = self.width
        # parameters and gradients are padded to a multiple of SHARD_UNIT floats so that the flat buffers split into
        # equal, float4-aligned shards for any world size up to 8 (sharded optimizer: reduce-scatter / all-gather
        # run in place on the padded buffers); `flat` / `flat_grad` are the un-padded views everything else uses
        n = P * W
        n_pad = (n + SHARD_UNIT - 1) // SHARD_UNIT * SHARD_UNIT
        self.flat_padded = torch.zeros((n_pad,), dtype=torch.float32, device=self.device)
        self.flat = self.flat_padded[:n]
        # gradient buffer + a [2, P] tail for this step's densification-statistic increments: the data-parallel
        # exchange is then ONE all-reduce (SUM) over gradients and increments together
        # ... and four more floats: word 0 = "some rank's view was invalid" (GsStepState.fail_flag: lists that proved too
        # short under depth limits, or a binning overflow), summed with everything else
        self.exchange = torch.zeros((n_pad + 2 * P + 4,), dtype=torch.float32, device=self.device)
        self.flat_grad = self.exchange[:n]
        self.grad_padded = self.exchange[:n_pad]
        self.stat_delta = self.exchange[n_pad:n_pad + 2 * P].view(2, P)
        self.stat_tail = self.exchange[n_pad:]            # statistics + flag: what travels beside the gradients
        self.fail_flag = self.exchange[n_pad + 2 * P:n_pad + 2 * P + 1]
        self.gain_grad_word = self.exchange[n_pad + 2 * P + 1:n_pad + 2 * P + 2]   # multispectral model: dL/dgain of this view
        # data-parallel step: 1 where this view's gradient row of the Gaussian may be non-zero (sparse exchange)
        self.grad_mask = torch.zeros((P,), dtype=torch.uint8, device=self.device)
        # ... and the union of the ranks' masks, exchanged EARLY (as soon as the forward knows which Gaussians have instances,
        # Trainer._begin_union), with its inclusive prefix - 1 (the packed row of every union member)
        self.union_mask = torch.zeros((P,), dtype=torch.uint8, device=self.device)
        self.union_pos = torch.zeros((P,), dtype=torch.int32, device=self.device)
        # the union as the row masks of gs_adam_step_masked (float, > 0 = stepped): members / everybody else
        self.union_rows_f = torch.zeros((2, P), dtype=torch.float32, device=self.device)
        self.params = {}
        off = 0
        shapes = self._shapes(P)
        for name, n in self.fields:
            view = self.flat[off:off + P * n].view(shapes[name])
            p = torch.nn.Parameter(view, requires_grad=True)
            p.grad = self.flat_grad[off:off + P * n].view(shapes[name])
            self.params[name] = p
            off += P * n

    # ------------------------------------------------------------------ files (formats: gsplat_amd/io.py)
    @classmethod
    def create_from_pcd(cls, points, colors, device, spatial_lr_scale=1.0, api=None, knn=None, spatial_order=None):
        """create_from_pcd (gaussian_model.py:149-176): DC = RGB2SH(colour), rest 0, scale = sqrt of the mean squared
        distance to the 3 nearest neighbours (clamped at 1e-7) on all axes, identity rotation, opacity 0.1, active
        SH degree 0.  knn: callable [P,3] -> [P] (simple_knn's distCUDA2); default = the float64 brute force."""
        from . import synthetic
        pts = torch.as_tensor(points, dtype=torch.float32)
        col = torch.as_tensor(colors, dtype=torch.float32)
        P = pts.shape[0]
        dist2 = (knn(pts) if knn is not None else synthetic.brute_force_knn_dist2(pts)).clamp_min(1e-7).cpu()
        shs = torch.zeros((P, 16, 3))
        shs[:, 0, :] = synthetic.rgb2sh(col)
        scene = dict(means3D=pts, shs=shs, scales=torch.sqrt(dist2)[:, None].repeat(1, 3),
                     rotations=torch.tensor([[1.0, 0.0, 0.0, 0.0]]).repeat(P, 1), opacities=torch.full((P, 1), 0.1),
                     sh_degree=0)
        return cls(scene, device, spatial_lr_scale=spatial_lr_scale, api=api, spatial_order=spatial_order)

    def save_ply(self, path):
        """save_ply (gaussian_model.py:240-256): raw parameters, the reference's 62-property layout."""
        from . import io as gio
        p = {k: v.detach().cpu().numpy() for k, v in self.params.items()}
        extra = {"nir_albedo": p["nir_albedo"]} if self.with_nir else None
        gio.save_gaussians_ply(path, p["xyz"], p["features"], p["opacity"], p["scaling"], p["rotation"], extra=extra)

    def load_ply(self, path):
        """load_ply (gaussian_model.py:263-314): replaces the parameters (fresh Adam state), active degree = max."""
        from . import io as gio
        d = gio.load_gaussians_ply(path, self.max_sh_degree)
        api = getattr(self.optimizer, "api", None)
        self._allocate(d["xyz"].shape[0])
        if self.with_nir and "nir_albedo" not in d:
            # mult-dwtgs/scene/gaussian_model.py:417-421: no NIR property in the file -> start from the DC coefficient
            d["nir_albedo"] = d["features"][:, 0, 0:1].copy()
        with torch.no_grad():
            for name in self.params:
                self.params[name].copy_(torch.from_numpy(d[name]).reshape(self.params[name].shape))
        if api is not None:
            self.optimizer = FlatAdam(api, self)
        self.xyz_gradient_accum = torch.zeros((self.P, 1), device=self.device)
        self.denom = torch.zeros((self.P, 1), device=self.device)
        self.max_radii2D = torch.zeros((self.P,), device=self.device)
        self.active_sh_degree = self.max_sh_degree

    def capture(self):
        """Checkpoint payload (the role of GaussianModel.capture, gaussian_model.py:62-76): flat parameters, Adam
        moments and step counts, densification statistics.  A model that is being trained is checkpointed through
        Trainer.checkpoint(), which first settles a pending depth-limit verdict and gathers sharded Adam moments."""
        opt = self.optimizer
        return dict(P=self.P, active_sh_degree=self.active_sh_degree, flat=self.flat.detach().cpu().clone(),
                    exp_avg=opt.exp_avg.cpu().clone(), exp_avg_sq=opt.exp_avg_sq.cpu().clone(), t=opt.t,
                    seg_steps=dict(opt.seg_steps), lr=dict(opt.lr), xyz_gradient_accum=self.xyz_gradient_accum.cpu().clone(),
                    denom=self.denom.cpu().clone(), max_radii2D=self.max_radii2D.cpu().clone(),
                    spatial_lr_scale=self.spatial_lr_scale,
                    nir_gain=None if self.nir_gain is None else self.nir_gain.detach().cpu().clone(),
                    nir_gain_optimizer=None if self.nir_gain is None else self.nir_gain_optimizer.state_dict())

    def restore(self, state):
        opt = self.optimizer
        self._allocate(int(state["P"]))
        with torch.no_grad():
            self.flat.copy_(state["flat"])
        opt.alloc_moments()
        opt.exp_avg.copy_(state["exp_avg"])
        opt.exp_avg_sq.copy_(state["exp_avg_sq"])
        opt.t, opt.seg_steps, opt.lr = int(state["t"]), dict(state["seg_steps"]), dict(state["lr"])
        self.active_sh_degree = int(state["active_sh_degree"])
        self.xyz_gradient_accum = state["xyz_gradient_accum"].to(self.device).clone()
        self.denom = state["denom"].to(self.device).clone()
        self.max_radii2D = state["max_radii2D"].to(self.device).clone()
        self.spatial_lr_scale = state["spatial_lr_scale"]
        if self.nir_gain is not None and state.get("nir_gain") is not None:
            with torch.no_grad():
                self.nir_gain.copy_(state["nir_gain"])
            self.nir_gain_optimizer.load_state_dict(state["nir_gain_optimizer"])
        if self.with_nir and "nir_gain" not in opt.seg_steps:
            # a checkpoint from before the gain's step count moved into seg_steps: it lived in the gain optimizer's own state
            try:
                opt.seg_steps["nir_gain"] = int(state["nir_gain_optimizer"]["state"][0]["step"])
            except (KeyError, IndexError, TypeError):
                opt.seg_steps["nir_gain"] = int(state["t"])

    def oneupSHdegree(self):
        """gaussian_model.py:145-147"""
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ------------------------------------------------------------------ densification
    def _relayout(self, src_idx, new_rows):
        """New model = rows `src_idx` of the current one (parameters AND Adam moments kept) followed by
        `new_rows` (dict field -> [n_new, width], moments zero): the flat-buffer form of _prune_optimizer +
        cat_tensors_to_optimizer (gaussian_model.py:331-393)."""
        opt = self.optimizer
        if not isinstance(opt, FlatAdam):
            raise NotImplementedError("densification needs the flat Adam state")
        old_p = {name: self.params[name].detach().reshape(self.P, n) for name, n in self.fields}
        old_m, old_v = opt.field_views(opt.exp_avg), opt.field_views(opt.exp_avg_sq)
        n_new = next(iter(new_rows.values())).shape[0] if new_rows else 0
        P2 = int(src_idx.numel()) + n_new
        cat_p, cat_m, cat_v = {}, {}, {}
        for name, n in self.fields:
            add = new_rows[name].reshape(n_new, n) if n_new else old_p[name][:0]
            cat_p[name] = torch.cat((old_p[name][src_idx], add), dim=0)
            z = torch.zeros_like(add)
            cat_m[name] = torch.cat((old_m[name][src_idx], z), dim=0)
            cat_v[name] = torch.cat((old_v[name][src_idx], z), dim=0)
        self._allocate(P2)
        opt.alloc_moments()
        new_m, new_v = opt.field_views(opt.exp_avg), opt.field_views(opt.exp_avg_sq)
        with torch.no_grad():
            for name, n in self.fields:
                self.params[name].reshape(P2, n).copy_(cat_p[name])
                new_m[name].copy_(cat_m[name])
                new_v[name].copy_(cat_v[name])
        for p in self.params.values():  # replaced tensors have no gradient until the next backward
            p.grad = None

    @staticmethod
    def build_rotation(r):
        """general_utils.py:78-99 (quaternion normalised here, unlike the rasterizer)."""
        q = r / torch.sqrt((r * r).sum(dim=1, keepdim=True))
        w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                         2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                         2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)), dim=1)
        return R.view(-1, 3, 3)

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, radii=None, generator=None, N=2,
                          decisions=None):
        """gaussian_model.py:395-467 in one re-layout: clone small Gaussians with a large view-space gradient,
        split large ones into N samples of their own distribution (scale / (0.8 N)), drop the split originals,
        then prune by opacity / world size.  Row order = the reference's: [survivors, clones, split samples].
        `generator`: CPU generator for the split samples (drawn on the CPU so every rank and every backend
        sees the same numbers).  Returns (n_clone, n_split, n_pruned).
        Note: the reference zeroes max_radii2D inside densification_postfix (:383) BEFORE the screen-size test of
        :458 reads it, so that test never fires; reproduced by construction.
        `decisions` (parity instrument, normally None): a dict.  Without the key "replay" the three DISCRETE decisions of this
        call - the clone mask, the split mask (both [P]) and the prune mask (over [survivors, clones, samples]) - are
        recorded into it (CPU bool tensors, plus the statistic `g` and the threshold they were taken with).  With
        decisions["replay"] = such a record, the recorded masks are USED instead of this model's own, and what this model
        would have decided is reported beside them (decisions["own"], decisions["disagree"]): two implementations whose
        statistics differ in the last bits take the same discrete trajectory, which separates threshold chaos from
        arithmetic differences (tests/test_gpu_psnr_parity.py)."""
        with torch.no_grad():
            P0 = self.P
            grads = self.xyz_gradient_accum / self.denom
            grads[grads.isnan()] = 0.0
            g = grads.reshape(P0)
            scaling = self.get_scaling
            max_scale = scaling.max(dim=1).values
            small = max_scale <= self.percent_dense * extent
            clone = (g.abs() >= max_grad) & small        # torch.norm over the single column (:435)
            split = (g >= max_grad) & ~small             # padded_grad of the clones is 0: never selected (:399-402)
            replay = decisions.get("replay") if decisions is not None else None
            if decisions is not None:
                own = dict(clone=clone.cpu(), split=split.cpu(), g=g.detach().cpu().clone(), max_grad=float(max_grad),
                           max_scale=max_scale.detach().cpu().clone(), scale_bound=float(self.percent_dense * extent))
                if replay is None:
                    decisions.update(own)
                else:
                    decisions["own"] = own
                    clone, split = replay["clone"].to(clone.device), replay["split"].to(split.device)
            ci = clone.nonzero().squeeze(1)
            si = split.nonzero().squeeze(1)
            ns = int(si.numel())
            raw = {name: self.params[name].detach().reshape(P0, n) for name, n in self.fields}
            new = {name: [raw[name][ci]] for name, _ in self.fields}
            if ns:
                stds = scaling[si].repeat(N, 1)
                if generator is None:
                    generator = torch.Generator().manual_seed(0)
                noise = torch.randn((ns * N, 3), generator=generator, dtype=torch.float32).to(stds.device)
                samples = noise * stds
                rots = self.build_rotation(raw["rotation"][si]).repeat(N, 1, 1)
                new_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + raw["xyz"][si].repeat(N, 1)
                new["xyz"].append(new_xyz)
                new["scaling"].append(torch.log(scaling[si].repeat(N, 1) / (0.8 * N)))
                for name, _ in self.fields:
                    if name not in ("xyz", "scaling"):
                        new[name].append(raw[name][si].repeat(N, 1))
            new = {name: torch.cat(v, dim=0) for name, v in new.items()}
            n_new = new["xyz"].shape[0]
            # final prune (:455-462) evaluated on [survivors, clones, samples]
            keep_old = ~split
            op_all = torch.cat((torch.sigmoid(raw["opacity"][keep_old]), torch.sigmoid(new["opacity"])), dim=0).squeeze(1)
            prune = op_all < min_opacity
            if max_screen_size:
                sc_all = torch.cat((max_scale[keep_old], torch.exp(new["scaling"]).max(dim=1).values), dim=0)
                prune = prune | (sc_all > 0.1 * extent)  # big_points_vs is all-False (max_radii2D was just zeroed)
            if decisions is not None:
                if replay is None:
                    decisions["prune"] = prune.cpu()
                    decisions["opacity"] = op_all.cpu()
                else:
                    decisions["own"]["prune"], decisions["own"]["opacity"] = prune.cpu(), op_all.cpu()
                    prune = replay["prune"].to(prune.device)
                    o = decisions["own"]
                    decisions["disagree"] = dict(clone=int((o["clone"] != replay["clone"]).sum()),
                                                 split=int((o["split"] != replay["split"]).sum()),
                                                 prune=int((o["prune"] != replay["prune"]).sum()))
            n_keep_old = int(keep_old.sum())
            src = keep_old.nonzero().squeeze(1)[~prune[:n_keep_old]]
            keep_new = ~prune[n_keep_old:]
            new = {name: v[keep_new] for name, v in new.items()}
            self._relayout(src, new)
            if getattr(self, "spatial_order", False):   # the new rows go where their neighbours are
                from . import synthetic
                self._relayout(synthetic.morton_order(self.params["xyz"]), {})
            self.xyz_gradient_accum = torch.zeros((self.P, 1), device=self.device)
            self.denom = torch.zeros((self.P, 1), device=self.device)
            self.max_radii2D = torch.zeros((self.P,), device=self.device)
            return int(ci.numel()), ns, int(prune.sum())

    def reset_opacity(self):
        """gaussian_model.py:258-261: opacity <- min(opacity, 0.01), moments of that group zeroed."""
        with torch.no_grad():
            op = torch.sigmoid(self.params["opacity"])
            new = torch.minimum(op, torch.full_like(op, 0.01))
            self.params["opacity"].copy_(torch.log(new / (1 - new)))
            opt = self.optimizer
            opt.field_views(opt.exp_avg)["opacity"].zero_()
            opt.field_views(opt.exp_avg_sq)["opacity"].zero_()

    def update_learning_rate(self, iteration, position_lr_final=0.0000016, delay_mult=0.01, max_steps=30000,
                             exposure_max_steps=None):
        """gaussian_model.py:213-223: exponential decay of the xyz learning rate (over position_lr_max_steps) and of the
        exposure rate (over training_args.iterations, gaussian_model.py:208-211: `exposure_max_steps`)."""
        lr = expon_lr(iteration, LRS["xyz"] * self.spatial_lr_scale, position_lr_final * self.spatial_lr_scale,
                      lr_delay_mult=delay_mult, max_steps=max_steps)
        if isinstance(self.optimizer, FlatAdam):
            self.optimizer.lr["xyz"] = lr
        else:
            self.optimizer.set_xyz_lr(lr)
        if self.exposure_optimizer is not None:  # gaussian_model.py:215-217
            elr = expon_lr(iteration, 0.01, 0.001, lr_delay_steps=0, lr_delay_mult=0.0,
                           max_steps=max_steps if exposure_max_steps is None else exposure_max_steps)
            for group in self.exposure_optimizer.param_groups:
                group["lr"] = elr
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

    def fused_activations(self):
        """(scales, rotations, opacities) through the fused kernels, or None when the model has no C-ABI optimizer
        (then the caller uses the get_* properties)."""
        api = getattr(self.optimizer, "api", None)
        if api is None or not hasattr(api, "_activations_fwd"):
            return None
        return _FusedActivations.apply(self.params["scaling"], self.params["rotation"], self.params["opacity"], self)

    def zero_grad(self):
        """set_to_none, as the reference does (train.py:283-288): autograd then ASSIGNS the new gradients
        instead of adding them to a zeroed buffer (no 236 B/Gaussian memset, no read-modify-write)."""
        for p in self.params.values():
            p.grad = None

    def grad_views(self):
        P, off, out = self.P, 0, {}
        shapes = self._shapes(P)
        for name, n in self.fields:
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

    def update_view_statistics(self, radii, viewspace_grad, into_delta=False):
        """train.py:266-268 + add_densification_stats in one kernel (gs_densify_stats) when available.
        into_delta: write this view's increments of xyz_gradient_accum / denom into `stat_delta` (zeroed first)
        instead of the running totals - the data-parallel step sums the increments over ranks before adding them."""
        api = getattr(self.optimizer, "api", None)
        if into_delta:
            self.stat_delta.zero_()
        acc = self.stat_delta[0] if into_delta else self.xyz_gradient_accum
        den = self.stat_delta[1] if into_delta else self.denom
        if api is not None and hasattr(api, "_densify_stats") and viewspace_grad is not None and viewspace_grad.is_contiguous():
            api.call("densify_stats", radii.data_ptr(), viewspace_grad.data_ptr(), self.P, self.max_radii2D.data_ptr(),
                     acc.data_ptr(), den.data_ptr(), _stream_of(radii))
            return
        if into_delta:
            vm = (radii > 0).to(torch.float32)
            torch.maximum(self.max_radii2D, radii.to(torch.float32), out=self.max_radii2D)
            acc += torch.norm(viewspace_grad[:, :2], dim=-1) * vm
            den += vm
            return
        # train.py:268 (radii are 0 for culled Gaussians, so a plain maximum equals the masked update)
        torch.maximum(self.max_radii2D, radii.to(torch.float32), out=self.max_radii2D)
        self.add_densification_stats(viewspace_grad, radii > 0)

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
           debug=False, filter_as_indices=True, clamp=True, fused=False, use_trained_exp=False, camera_index=None,
           raw_activations=False, camera_key=None):
    """= render() of LGDWT-GS/gaussian_renderer/__init__.py:18-128 (SH evaluated by the rasterizer, scale +
    rotation given, no exposure): returns {render, viewspace_points, visibility_filter, radii, depth}.
    raw_activations: hand the rasterizer the RAW scaling / rotation / opacity rows (the caller has told the backend,
    RasterBackend.raw_activations: the kernels activate them on the fly) - the fused train step's form."""
    act = pc.fused_activations() if fused and not raw_activations else None
    if raw_activations:
        act = (pc.params["scaling"], pc.params["rotation"], pc.params["opacity"])
        screenspace_points = torch.empty_like(pc.get_xyz).requires_grad_(True)
    elif act is None:
        screenspace_points = torch.zeros_like(pc.get_xyz, dtype=pc.get_xyz.dtype, requires_grad=True) + 0
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass
        act = (pc.get_scaling, pc.get_rotation, pc.get_opacity)
    else:  # the step loop's lean form: the leaf only carries the view-space gradient - the rasterizer never reads
        # means2D (rasterize_points.cu takes no such argument) - so it is not even filled
        screenspace_points = torch.empty_like(pc.get_xyz).requires_grad_(True)
    scales, rotations, opacities = act
    rs = Settings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=debug, antialiasing=antialiasing)
    rasterizer = Rasterizer(raster_settings=rs)
    if camera_key is not None:  # stable camera identity for the backend's per-camera state (GaussianRasterizer.camera_key)
        rasterizer.camera_key = camera_key
        rasterizer.camera_limits = False   # (the train step asks for depth limits itself: Trainer.depth_limit)
    rendered_image, radii, depth_image = rasterizer(
        means3D=pc.get_xyz, means2D=screenspace_points, shs=pc.get_features, colors_precomp=None,
        opacities=opacities, scales=scales, rotations=rotations, cov3D_precomp=None)
    if use_trained_exp:  # gaussian_renderer/__init__.py:112-115 (training only): 3x3 colour matrix + offset per camera
        exposure = pc.get_exposure(camera_index)
        rendered_image = torch.matmul(rendered_image.permute(1, 2, 0), exposure[:3, :3]).permute(2, 0, 1) + \
            exposure[:3, 3, None, None]
    if clamp:  # gaussian_renderer/__init__.py:119; the fused criterion applies (and differentiates) it itself
        rendered_image = rendered_image.clamp(0, 1)
    return {"render": rendered_image, "viewspace_points": screenspace_points,
            # the reference returns indices ((radii > 0).nonzero(): a host sync); the step loop asks for the
            # equivalent boolean mask instead and stays asynchronous
            # (filter_as_indices=None: not at all - the fused statistics kernel reads radii itself)
            "visibility_filter": None if filter_as_indices is None else
            ((radii > 0).nonzero() if filter_as_indices else (radii > 0)),
            "radii": radii, "depth": depth_image}


def camera_to(cam, device):
    return cam._replace(world_view_transform=cam.world_view_transform.to(device),
                        full_proj_transform=cam.full_proj_transform.to(device),
                        camera_center=cam.camera_center.to(device))


class TrainOptions:
    """The schedule constants of LGDWT-GS/arguments/__init__.py:78-100 (OptimizationParams)."""

    def __init__(self, **kw):
        self.iterations = 30_000
        self.position_lr_final = 0.0000016
        self.position_lr_delay_mult = 0.01
        self.position_lr_max_steps = 30_000
        self.densification_interval = 100
        self.opacity_reset_interval = 3000
        self.densify_from_iter = 500
        self.densify_until_iter = 15_000
        self.densify_grad_threshold = 0.0002
        self.min_opacity = 0.005          # train.py:273
        self.size_threshold = 20          # train.py:272
        self.sh_increase_interval = 1000  # train.py:102
        self.depth_l1_weight_init = 1.0   # arguments/__init__.py:98-99; train.py:69: expon schedule over `iterations`
        self.depth_l1_weight_final = 0.01
        self.optimizer_type = "default"   # arguments/__init__.py:101 ("sparse_adam": Trainer(optimizer_type=...))
        self.white_background = False
        self.cameras_extent = 1.0         # scene.cameras_extent = getNerfppNorm radius (dataset_readers.py:48-69)
        self.seed = 0
        for k, v in kw.items():
            if not hasattr(self, k):
                raise TypeError("unknown option %s" % k)
            setattr(self, k, v)


def cameras_extent(camera_centers):
    """getNerfppNorm (dataset_readers.py:48-69): 1.1 x the largest distance of a camera centre from their mean."""
    c = torch.stack([torch.as_tensor(x, dtype=torch.float64).reshape(3) for x in camera_centers])
    return float((c - c.mean(dim=0, keepdim=True)).norm(dim=1).max() * 1.1)


class Trainer:
    """One process per GPU; `step(k)` renders this rank's camera of global step k and updates the model;
    `train_iteration(it, opt)` is one iteration of the reference's loop with its schedule."""

    def __init__(self, model, cameras, gt_images, criterion, Rasterizer, Settings, bg, rank=0, world_size=1,
                 optimizer_step=True, masks=None, sharded_optimizer=None, sparse_exchange=None, optimizer_type="default"):
        # sharded_optimizer (default: env GS_SHARDED_ADAM=1): reduce-scatter the gradients, Adam on this rank's 1/N of
        # the rows, all-gather the parameters - instead of all-reduce + the full Adam pass on every replica
        import os
        # Default for N > 1: sharded (GS_SHARDED_ADAM=0: all-reduce).  Byte model: both forms move 2 (N-1)/N x 244 B per
        # Gaussian over each GPU's links (an all-reduce IS a reduce-scatter + all-gather); the sharded form does it in two
        # large collectives instead of four chunks, runs 1/N of the 28 B/element optimizer pass, and owes an all-gather of
        # the moments only before a densification or a checkpoint.
        self.sharded_optimizer = (os.environ.get("GS_SHARDED_ADAM", "1") == "1") if sharded_optimizer is None \
            else bool(sharded_optimizer)
        # N > 1: move only the gradient rows of the Gaussians some rank's view reached (_exchange_and_step_sparse; replaces
        # the sharded / all-reduce forms when on).  GS_SPARSE_EXCHANGE=1 / sparse_exchange=True
        self.sparse_exchange = (os.environ.get("GS_SPARSE_EXCHANGE", "0") == "1") if sparse_exchange is None \
            else bool(sparse_exchange)
        self.last_exchange = None
        # optimizer_type = "sparse_adam" (train.py:68, 282-284): only the Gaussians with radii > 0 in the step's view - on N
        # GPUs: in SOME rank's view - are stepped (FlatAdam.sparse).  The moments then cannot be sharded by element range
        # (a shard would need the visibility of rows it does not own the statistics of): all-reduce or sparse exchange.
        if optimizer_type not in ("default", "sparse_adam"):
            raise ValueError("optimizer_type must be 'default' or 'sparse_adam'")
        self.optimizer_type = optimizer_type
        if optimizer_type == "sparse_adam":
            if not isinstance(model.optimizer, FlatAdam):
                raise RuntimeError("sparse_adam needs the flat Adam state (a model built with an api)")
            model.optimizer.sparse = True
            self.sharded_optimizer = False
        self.model, self.cameras, self.gts, self.criterion = model, cameras, gt_images, criterion
        self.Rasterizer, self.Settings, self.bg = Rasterizer, Settings, bg
        self.rank, self.world_size = rank, world_size
        self.optimizer_step = optimizer_step
        self.masks = masks  # per-camera ELF patch masks (depend on the ground truth only): cached
        # Depth regularisation (train.py:204-216): per camera None or (mono_invdepth [1,H,W], depth_mask [1,H,W] or None) - the
        # reference's viewpoint_cam.invdepthmap / depth_mask of a camera with depth_reliable (scene/cameras.py:60-80) - and the
        # current weight (train_iteration sets it from the schedule of train.py:69; 0 = term off)
        self.depth_priors = None
        self.depth_l1_weight = 0.0
        self.last = None
        # who this trainer is in the backend's per-camera state (keys ("trainer", uid, camera index)): a counter, not id() -
        # the address of a dead trainer is handed to the next one, which would then inherit its cameras' hints and limits
        Trainer._uids = getattr(Trainer, "_uids", 0) + 1
        self.uid = Trainer._uids
        # the backend keeps per-camera state under ("trainer", uid, camera): it goes when this trainer does
        be = getattr(getattr(getattr(self.Rasterizer, "_fn", None), "_impl", None), "backend", None)
        if be is not None and hasattr(be, "drop_camera_entries"):
            import weakref
            weakref.finalize(self, be.drop_camera_entries, ("trainer", self.uid))
        self._stack = []    # train.py:106-111: cameras are drawn without replacement
        self._rng = None

    def camera_index(self, k):
        return (k * self.world_size + self.rank) % len(self.cameras)

    def step(self, k):
        return self._step_camera(self.camera_index(k), self.optimizer_step, ())

    def draw_cameras(self, seed):
        """train.py:106-111 for `world_size` ranks: one global step pops world_size cameras from the shared stack
        (same python RNG on every rank), rank r takes the r-th."""
        import random
        if self._rng is None:
            self._rng = random.Random(seed)
        mine = None
        for r in range(self.world_size):
            if not self._stack:
                self._stack = list(range(len(self.cameras)))
            ci = self._stack.pop(self._rng.randint(0, len(self._stack) - 1))
            if r == self.rank:
                mine = ci
        return mine

    densify_decisions = None

    def train_iteration(self, iteration, opt):
        """One iteration (1-based) of LGDWT-GS/train.py:97-288: LR schedule, SH ramp, camera draw, render + loss +
        backward, densification statistics, densify / prune / opacity reset on schedule, Adam.
        The reference replaces the parameter tensors when it densifies (their gradients are gone), so
        optimizer.step() of that iteration changes nothing; after reset_opacity only the opacity group is
        skipped.  Returns dict(loss, densified=(n_clone, n_split, n_pruned) or None, reset=bool, P)."""
        # a step whose depth limits failed is repeated HERE, before this iteration's learning rate / SH degree exist: the
        # repeat must see the schedule state of the iteration it belongs to
        self.sync()
        m = self.model
        m.update_learning_rate(iteration, opt.position_lr_final, opt.position_lr_delay_mult, opt.position_lr_max_steps,
                               exposure_max_steps=opt.iterations)
        if iteration % opt.sh_increase_interval == 0:
            m.oneupSHdegree()
        if self.depth_priors is not None:   # train.py:69
            self.depth_l1_weight = expon_lr(iteration, opt.depth_l1_weight_init, opt.depth_l1_weight_final, max_steps=opt.iterations)
        ci = self.draw_cameras(opt.seed)
        # What the schedule will do after the backward is known beforehand, so the optimizer step can run inside the
        # step (fused into the backward on one GPU, overlapped with the all-reduce on several).  Its order against
        # reset_opacity is free: the reset replaces the opacity row and its moments, the step skips exactly that row.
        in_densify = iteration < opt.densify_until_iter
        will_densify = in_densify and iteration > opt.densify_from_iter and iteration % opt.densification_interval == 0
        will_reset = in_densify and (iteration % opt.opacity_reset_interval == 0 or
                                     (opt.white_background and iteration == opt.densify_from_iter))
        do_step = iteration < opt.iterations and not will_densify
        # the exposure tensor is never replaced by densification, so its optimizer steps on those iterations too
        # (train.py:279-281: every iteration below opt.iterations)
        loss = self._step_camera(ci, do_step, ("opacity",) if will_reset else (), exposure_step=iteration < opt.iterations)
        if will_densify or will_reset or iteration >= opt.iterations:
            self.sync()  # (depth-limited steps: the model is about to be read / re-laid out)
        densified, reset = None, False
        if will_densify:
            self.gather_optimizer_state()
            thr = opt.size_threshold if iteration > opt.opacity_reset_interval else None
            gen = torch.Generator().manual_seed(opt.seed * 1000003 + iteration)
            # (densify_decisions: parity instrument, see GaussianModelLite.densify_and_prune - a callable iteration -> dict)
            dec = self.densify_decisions(iteration) if self.densify_decisions is not None else None
            densified = m.densify_and_prune(opt.densify_grad_threshold, opt.min_opacity, opt.cameras_extent, thr,
                                            self.last["radii"], generator=gen, decisions=dec)
        if will_reset:
            m.reset_opacity()
            reset = True
        return dict(loss=loss, densified=densified, reset=reset, P=m.P, camera=ci)

    # single-GPU steps run the optimizer inside the rasterizer backward (gs_backward_step); GS_FUSED_STEP=0 keeps the
    # separate activation-backward / statistics / Adam kernels (the path every multi-GPU step takes)
    FUSED_STEP = __import__("os").environ.get("GS_FUSED_STEP", "1") != "0"

    def _fused_step_ok(self, backend, optimizer_step):
        m = self.model
        return (self.FUSED_STEP and optimizer_step and self.world_size == 1 and backend is not None
                and isinstance(m.optimizer, FlatAdam) and (not m.with_nir or self.RENDERS_NIR) and m.flat.is_cuda
                and hasattr(backend.api, "_backward_step") and hasattr(backend, "fused_step"))

    RENDERS_NIR = False   # (TrainerNIR: the step renders and steps the model's 4th channel)

    def _fused_dp_ok(self, backend, optimizer_step):
        """N > 1: the data-parallel form of the fused backward (GsStepState.grad_out): raw rows in, the 59 gradient floats per
        Gaussian written straight into the exchange buffer together with this view's statistic increments and validity
        flag - no activation kernels, no activated copies, no packing; the optimizer then runs gated by the reduced flag."""
        m = self.model
        return (self.FUSED_STEP and optimizer_step and self.world_size > 1 and backend is not None
                and isinstance(m.optimizer, FlatAdam) and (not m.with_nir or self.RENDERS_NIR) and m.flat.is_cuda
                and hasattr(backend.api, "_backward_step") and hasattr(backend.api, "_adam_step_gated")
                and hasattr(backend, "fused_step"))

    # Depth-limited instance lists (RasterBackend.depth_limit_request) for the fused single-GPU step: None = off,
    # "deferred" = on, with the forward's verdict ("were the cut lists long enough?") collected one step later, when it
    # costs no wait.  A step whose limits failed changed nothing on the device (gs_backward_step checks the same flag),
    # so it is simply taken again without limits; until then its loss tensor holds the value of the invalid image - it
    # is overwritten in place.  Anything that reads the model between steps goes through sync() first.
    depth_limit = None
    # the fused step hands the rasterizer the raw scaling / rotation / opacity rows (GsGaussians.raw_activations): the
    # activation kernel and its three output tensors disappear from the step; False keeps them
    RAW_ACTIVATIONS = True

    def sync(self):
        """Settle the verdict of the last depth-limited step (redoing the step if its limits failed), and that of the last
        replay of a GraphedStep bound to this trainer."""
        g = getattr(self, "_graphed", None)
        if g is not None:
            g.settle()
        p, self._pending = getattr(self, "_pending", None), None
        if p is None:
            return
        if "flag_host" in p:
            # data-parallel step: the verdict is the REDUCED flag - the same number on every rank, so that all ranks repeat
            # the step (its collectives included) or none does; this rank's own verdict only updates its hints and limits
            p["event"].synchronize()
            if p["verdict"] is not None:
                p["verdict"]()
            if float(p["flag_host"][0]) == 0.0:
                return
        elif p["verdict"]():
            return
        opt = self.model.optimizer
        opt.t, opt.seg_steps = p["counters"][0], dict(p["counters"][1])
        if p["running_mean"] is not None:
            self.criterion.dwt_running_mean.copy_(p["running_mean"])
        p["loss"].copy_(self._step_camera(p["ci"], True, p["skip"]))  # (the camera's limits are invalid now: full lists)
        self.sync()

    def checkpoint(self):
        """model.capture() of a model in training: the pending depth-limit verdict is settled (so Adam's counters count
        only steps that happened) and, with the sharded optimizer, every rank's moments are gathered first."""
        self.sync()
        self.gather_optimizer_state()
        return self.model.capture()

    def _backend(self):
        be = getattr(getattr(self.Rasterizer, "_fn", None), "_impl", None)
        return getattr(be, "backend", None)

    # The single-GPU fused step without the autograd engine: its graph is two custom nodes in a row (rasterizer, criterion) whose
    # gradients end inside gs_backward_step, so the step calls their forward / backward bodies itself (a stand-in context
    # object each) - no graph construction, no engine thread hand-off: ~0.15 ms less host time per step, which is what
    # keeps the eager step GPU-bound on slow hosts.  Same kernels, same arguments, same bits.  GS_MANUAL_BACKWARD=0: autograd.
    MANUAL_BACKWARD = __import__("os").environ.get("GS_MANUAL_BACKWARD", "1") != "0"

    class _ManualCtx:
        """What a torch.autograd.Function's forward / backward use of their context, and nothing else."""

        def __init__(self):
            self.saved_tensors = ()

        def save_for_backward(self, *tensors):
            self.saved_tensors = tensors

        def mark_non_differentiable(self, *a):
            pass

        def set_materialize_grads(self, v):
            pass

    def _manual_step_ok(self, fused_step, fused):
        fn = getattr(self.Rasterizer, "_fn", None)
        return (self.MANUAL_BACKWARD and fused_step and fused and self.model.exposure is None and fn is not None
                and getattr(getattr(fn, "_impl", None), "backend", None) is not None
                and type(self)._render_view is Trainer._render_view and type(self)._criterion_backward is Trainer._criterion_backward)

    def _manual_step(self, ci, mask, backend, deferred):
        """render -> criterion -> backward of the fused single-GPU step, by hand (see MANUAL_BACKWARD) -> (pkg, loss, parts,
        the forward's deferred verdict or None)"""
        m, cam = self.model, self.cameras[ci]
        fn = self.Rasterizer._fn
        rs = self.Settings(
            image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=math.tan(cam.FoVx * 0.5),
            tanfovy=math.tan(cam.FoVy * 0.5), bg=self.bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform,
            projmatrix=cam.full_proj_transform, sh_degree=m.active_sh_degree, campos=cam.camera_center, prefiltered=False,
            debug=False, antialiasing=False)
        backend.camera_key = ("trainer", self.uid, ci)   # (as GaussianRasterizer.forward does with its camera_key)
        backend.camera_key_limits = False
        empty = getattr(self, "_empty_cpu", None)
        if empty is None:
            empty = self._empty_cpu = torch.Tensor([])
        rctx, lctx = self._ManualCtx(), self._ManualCtx()
        with torch.no_grad():
            p = m.params
            color, radii, depth = fn.forward(rctx, p["xyz"], None, p["features"], empty, p["opacity"], p["scaling"], p["rotation"],
                                             empty, rs)
            verdict = backend.take_deferred() if deferred else None
            loss, parts = self.criterion.fused_call(color, self.gts[ci], mask=mask, manual_ctx=lctx)
            grad_depth = None
            prior = self._depth_prior(ci)
            if prior is not None:   # train.py:204-216: one launch gives the term and its inverse-depth image gradient
                dl, grad_depth = self.criterion.ops.depth_l1_step(depth, prior[0], prior[1], self.depth_l1_weight)
                loss = loss + dl
                parts = dict(parts, depth_l1=dl)
            self._arm_side_launch(backend.launch_uninstanced_early)
            from .losses import FusedLGDWTLoss
            grad_img = FusedLGDWTLoss.backward(lctx, self.criterion.ops.unit_grad(loss.device), None)[1]
            fn.backward(rctx, grad_img, None, grad_depth)
        pkg = {"render": color, "viewspace_points": None, "visibility_filter": None, "radii": radii, "depth": depth}
        return pkg, loss, parts, verdict

    # -- what differs between the RGB step and the multispectral one (TrainerNIR)
    def _render_view(self, ci, fused, raw):
        m = self.model
        return render(self.cameras[ci], m, self.Rasterizer, self.Settings, self.bg, filter_as_indices=None,
                      clamp=not fused, fused=True, use_trained_exp=m.exposure is not None, camera_index=ci,
                      raw_activations=raw, camera_key=("trainer", self.uid, ci))

    def _depth_prior(self, ci):
        """(mono_invdepth, depth_mask) of camera ci when the depth term is on for it (train.py:206), else None"""
        if self.depth_priors is None or not self.depth_l1_weight > 0:
            return None
        return self.depth_priors[ci]

    def _depth_term(self, pkg, ci, loss, parts):
        prior = self._depth_prior(ci)
        if prior is None:
            return loss, parts
        dl = self.depth_l1_weight * self.criterion.ops.depth_l1(pkg["depth"], prior[0], prior[1])
        return loss + dl, dict(parts, depth_l1=dl.detach())

    def _criterion_backward(self, pkg, ci, mask, fused, side_launch):
        """-> (loss, parts); runs the backward.  side_launch: callable that issues the two-phase step's side launch (or None)"""
        if fused:
            loss, parts = self.criterion.fused_call(pkg["render"], self.gts[ci], mask=mask)
            loss, parts = self._depth_term(pkg, ci, loss, parts)
            # two-phase step: the Adam stream of the Gaussians without instances starts on its side stream from inside the
            # criterion's backward (RasterBackend.UNINST_AT) - it needs nothing of the loss - and runs beside the blend
            self._arm_side_launch(side_launch)
            # (seeded with the criterion's cached constant 1: no fill kernel for the implicit seed, no multiply by it)
            torch.autograd.backward(loss, self.criterion.ops.unit_grad(loss.device))
        else:
            loss, parts = self.criterion(pkg["render"], self.gts[ci], mask=mask)
            loss, parts = self._depth_term(pkg, ci, loss, parts)
            loss.backward()
        return loss, parts

    def _arm_side_launch(self, side_launch):
        ops = self.criterion.ops
        ops.before_last_backward_kernel = None
        if side_launch is None:
            return
        backend = self._backend()
        if backend.UNINST_AT == "ssim_backward":      # ... between the criterion's two backward kernels
            ops.before_last_backward_kernel = side_launch
        else:
            side_launch()

    def _unfused_tail(self, pkg, radii, optimizer_step, skip):
        m = self.model
        self.last_radii = radii
        with torch.no_grad():
            m.collect_grads()
            m.update_view_statistics(radii, pkg["viewspace_points"].grad, into_delta=self.world_size > 1)
            if self.world_size > 1:   # (a culled Gaussian's gradient row is all zero, rasterize_points.cu:163-172)
                torch.gt(radii, 0, out=m.grad_mask.view(torch.bool))
            self.exchange_and_step(optimizer_step, skip)

    def _step_camera(self, ci, optimizer_step, skip, exposure_step=None):
        self.sync()
        m = self.model
        if exposure_step is None:
            exposure_step = optimizer_step
        m.zero_grad()
        backend = self._backend()
        fused_step = self._fused_step_ok(backend, optimizer_step)
        fused_dp = self._fused_dp_ok(backend, optimizer_step)
        # (not with a trainable exposure: its torch optimizer has stepped on the invalid image by the time the verdict
        # arrives, and a repeat would step it twice)
        deferred = (fused_step or fused_dp) and self.depth_limit == "deferred" and getattr(self, "_coef_dev", None) is None \
            and m.exposure is None
        if deferred:
            counters = (m.optimizer.t, dict(m.optimizer.seg_steps))
            backend.depth_limit_request = "defer"
        if fused_dp:
            backend.raw_activations = self.RAW_ACTIVATIONS
            backend.fused_step = m.optimizer.grads_out_request()
            if not self.RAW_ACTIVATIONS:
                raise RuntimeError("the data-parallel fused backward takes the raw parameter rows (Trainer.RAW_ACTIVATIONS)")
        elif fused_step:
            backend.raw_activations = self.RAW_ACTIVATIONS
            backend.fused_step = m.optimizer.fused_request(skip, coef_dev=getattr(self, "_coef_dev", None))
            rows = getattr(self, "rows_override", None)  # parity tests: blend sums to use instead of stage 1
            if rows is not None:
                backend.fused_step.rows_override = rows.data_ptr()
        elif backend is not None and hasattr(backend, "grad_arena") and not m.with_nir:
            m.arm_grad_arena(backend)
        fused = getattr(self.criterion, "fused", False)
        mask = None if self.masks is None else self.masks[ci]
        rm = None
        if self.RAW_ACTIVATIONS and self._manual_step_ok(fused_step, fused):
            # (the deferred verdict of THIS forward is picked up inside: the backend parks it when the forward returns)
            pkg, loss, parts, verdict = self._manual_step(ci, mask, backend, deferred)
        else:
            pkg = self._render_view(ci, fused, (fused_step or fused_dp) and self.RAW_ACTIVATIONS)
            verdict = backend.take_deferred() if deferred else None
            if self.world_size > 1 and self.sparse_exchange:
                # which rows this view's gradient can touch is known NOW: the exchange of the masks starts behind the forward
                # (side stream), under the criterion and the backward
                if fused_dp and hasattr(backend, "export_row_mask") and backend.export_row_mask(m.union_mask):
                    self._begin_union()
                else:
                    self._begin_union(pkg["radii"] > 0)
            if verdict is not None and not fused:
                rm = getattr(self.criterion, "dwt_running_mean", None)
                rm = None if rm is None else rm.clone()
            loss, parts = self._criterion_backward(pkg, ci, mask, fused,
                                                   backend.launch_uninstanced_early if (fused_step and fused) else None)
        radii = pkg["radii"]
        if fused_dp:
            # gradients, statistic increments and the validity flag are in the exchange buffer: reduce, then the gated
            # optimizer; every rank learns the reduced flag one step later (sync) and repeats the step if it is set
            if backend.fused_step is not None:
                backend.fused_step = None
                raise RuntimeError("fused data-parallel step armed but the rasterizer backward did not run")
            with torch.no_grad():
                self.exchange_and_step(optimizer_step, skip, gate=m.fail_flag)
            if deferred:
                host = self._flag_blocks[self._flag_i % len(self._flag_blocks)] if getattr(self, "_flag_blocks", None) else None
                if host is None:
                    self._flag_blocks = [torch.zeros((1,), dtype=torch.float32).pin_memory() for _ in range(4)]
                    self._flag_i = 0
                    host = self._flag_blocks[0]
                self._flag_i += 1
                host.copy_(m.fail_flag, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(m.flat.device))
                if fused:
                    rm = parts.get("running_mean_before")
                self._pending = dict(verdict=verdict, flag_host=host, event=ev, ci=ci, skip=skip, counters=counters,
                                     running_mean=rm, loss=loss.detach())
        elif verdict is not None:
            if fused:  # (the criterion's combine kernel leaves the running mean it started from in its output: no clone)
                rm = parts.get("running_mean_before")
            self._pending = dict(verdict=verdict, ci=ci, skip=skip, counters=counters, running_mean=rm, loss=loss.detach())
        if fused_dp:
            pass
        elif fused_step:
            # gs_backward_step has already applied the activation backward, the view statistics and Adam
            if backend.fused_step is not None:
                backend.fused_step = None
                raise RuntimeError("fused train step armed but the rasterizer backward did not run")
        else:
            self._unfused_tail(pkg, radii, optimizer_step, skip)
        if m.exposure_optimizer is not None:
            # train.py:280-281: the exposure optimizer steps with the main one; a camera's row is only touched by the
            # rank that rendered it, the others hold a zero gradient for it (N > 1: summed like every other gradient)
            if m.exposure.grad is not None:
                if self.world_size > 1:
                    dist.all_reduce(m.exposure.grad, op=dist.ReduceOp.SUM)
                if exposure_step:
                    m.exposure_optimizer.step()
            m.exposure_optimizer.zero_grad(set_to_none=True)
        self.last = dict(loss=loss.detach(), radii=radii, parts=parts)
        return loss.detach()

    DP_CHUNKS = 4

    def exchange_and_step(self, optimizer_step, skip=(), gate=None):
        """The one exchange step of the data-parallel path and the optimizer step.  The exchange buffer = the 59-
        floats-per-Gaussian gradients (236 B x P) followed by this step's increments of the densification statistics
        (8 B x P) is summed over ranks in DP_CHUNKS consecutive all-reduces (RCCL over xGMI; gloo in the CPU tests)
        issued back to back; as soon as chunk k has arrived its slice of the flat Adam update runs while chunks
        k+1.. are still on the links, so the optimizer (0.3 ms at 1 M Gaussians) hides behind the communication.
        A max over max_radii2D travels alongside.  The summed increments are then added to the running totals, so
        every replica holds the statistics of ALL cameras of the step."""
        m = self.model
        opt = m.optimizer
        chunked = isinstance(opt, FlatAdam)
        if self.world_size <= 1:
            if optimizer_step:
                if getattr(opt, "sparse", False):   # train.py:283: visible = radii > 0
                    opt.step(skip, row_mask=(self.last_radii > 0).to(torch.float32))
                else:
                    opt.step(*([skip] if skip else []))
            return
        timed = getattr(self, "exchange_events", None) is not None and m.flat.is_cuda
        if timed:  # bench.py: how long the compute stream sees the exchange + optimizer take
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(m.flat.device))
        try:
            return self._exchange_and_step(optimizer_step, skip, gate)
        finally:
            if timed:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record(torch.cuda.current_stream(m.flat.device))
                self.exchange_events.append((e0, e1))

    def _exchange_and_step(self, optimizer_step, skip=(), gate=None):
        m = self.model
        opt = m.optimizer
        chunked = isinstance(opt, FlatAdam)
        if self.sparse_exchange and chunked:
            return self._exchange_and_step_sparse(optimizer_step, skip, gate)
        if self.sharded_optimizer and chunked:
            return self._exchange_and_step_sharded(optimizer_step, skip, gate)
        n_grad = m.flat_grad.numel()
        nch = self.DP_CHUNKS if chunked else 1
        n_stat_end = m.exchange.numel() - 4     # (the last four floats: validity flag + padding)
        wflag = None
        if gate is not None:  # the flag first, on its own: every chunk's gated update needs it
            wflag = dist.all_reduce(m.exchange[n_stat_end:], op=dist.ReduceOp.SUM, async_op=True)
        bounds = [(i * n_grad // nch) // 4 * 4 for i in range(nch)] + [n_stat_end]
        works = [dist.all_reduce(m.exchange[bounds[i]:bounds[i + 1]], op=dist.ReduceOp.SUM, async_op=True)
                 for i in range(nch)]
        if wflag is not None:
            wflag.wait()
        wmax = dist.all_reduce(m.max_radii2D, op=dist.ReduceOp.MAX, async_op=True)
        if optimizer_step and chunked:
            opt.begin_step(skip)
        sparse = chunked and opt.sparse
        for i, w in enumerate(works):
            w.wait()
            if optimizer_step and chunked and not sparse:
                opt.step_range(bounds[i], min(bounds[i + 1], n_grad), skip, gate=gate)
        if optimizer_step and sparse:
            # sparse_adam: the rows to step are those visible in some rank's view - the summed `denom` increments, which
            # travel behind the gradients in the last chunk: one update when everything has arrived
            opt.step_range(0, n_grad, skip, gate=gate, row_mask=m.stat_delta[1])
        if optimizer_step and not chunked:
            opt.step()
        wmax.wait()
        self._add_statistics(gate)

    def _add_statistics(self, gate):
        """running totals += the summed increments of all ranks' views - unless some rank's view was invalid (gate != 0):
        the step is about to be repeated, nothing of it may stay."""
        m = self.model
        if gate is None:
            m.xyz_gradient_accum += m.stat_delta[0].unsqueeze(1)
            m.denom += m.stat_delta[1].unsqueeze(1)
        else:
            ok = (gate == 0).to(torch.float32)
            m.xyz_gradient_accum += (m.stat_delta[0] * ok).unsqueeze(1)
            m.denom += (m.stat_delta[1] * ok).unsqueeze(1)

    # ---- the visibility-sparse exchange
    SPLIT_SPARSE_ADAM = True   # the optimizer of the sparse exchange in two parts, the first under the collective (False: one pass)

    def _begin_union(self, mask_src=None):
        """Start the exchange of the row masks NOW: `union_mask` <- this rank's mask (mask_src, or what the forward's geometry
        state says: Gaussians with instances), all-reduced (MAX) over the ranks, then its prefix (the packed row of every
        member) and its size K.  On the GPU all of it is enqueued on a SIDE stream right behind the forward - the collective is
        1 byte per Gaussian - while the criterion and the backward run on the main stream; K travels to a pinned host word,
        and the host reads it only when the backward has been enqueued (_exchange_and_step_sparse): off the critical path.
        No-op unless the sparse exchange is on."""
        if not (self.sparse_exchange and self.world_size > 1):
            return
        m = self.model
        dev = m.flat.device

        def work():
            if mask_src is not None:
                m.union_mask.copy_(mask_src.view(torch.uint8) if mask_src.dtype == torch.bool else mask_src)
            dist.all_reduce(m.union_mask, op=dist.ReduceOp.MAX)
            torch.cumsum(m.union_mask, dim=0, dtype=torch.int32, out=m.union_pos)
            m.union_pos.sub_(1)
            m.union_rows_f[0].copy_(m.union_mask)             # (uint8 -> float: 1.0 for the members)
            m.union_rows_f[1].copy_(m.union_mask == 0)
        if dev.type == "cuda":
            main = torch.cuda.current_stream(dev)
            side = getattr(self, "_union_stream", None)
            if side is None:
                side = self._union_stream = torch.cuda.Stream(device=dev)
                self._union_k = torch.zeros((1,), dtype=torch.int32).pin_memory()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                work()
                self._union_k.copy_(m.union_pos[-1:], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(side)
            self._union = dict(event=ev, stream=side)
        else:
            work()
            self._union = dict(event=None, stream=None, K=int(m.union_pos[-1]) + 1 if m.P else 0)

    def _exchange_and_step_sparse(self, optimizer_step, skip=(), gate=None):
        """Only Gaussians that emitted instances in SOME rank's view of this step have a non-zero gradient row - with
        depth-limited lists a fifth of them per view, and never more than the ~40 % any camera reaches - so: the union of the
        ranks' row masks (1 byte per Gaussian, exchanged early: _begin_union), the union's rows of the field-major gradient
        buffer packed into one contiguous buffer by ONE kernel (gs_rows_pack), THAT all-reduced, the sums scattered back
        (gs_rows_unpack), and the dense gated Adam on every replica (rows outside the union hold the exact sum already: zero
        on every rank - their update is the zero-gradient one, identical everywhere, no traffic).  Statistics + flag
        (8 B/Gaussian) and max_radii2D travel dense as in the other forms.  With two ranks the sums are the dense
        all-reduce's bit for bit (a + b commutes); with more the reduction order is the library's choice in either form.  The
        host reads the union's size (the collective's message length) from a pinned word the side stream filled while the
        backward ran."""
        import ctypes as C
        m, opt = self.model, self.model.optimizer
        P, W = m.P, m.width
        n_grad = m.flat_grad.numel()
        u = getattr(self, "_union", None)
        self._union = None
        if u is None:   # (nobody started it early: the backward's own mask, now)
            self._begin_union(m.grad_mask)
            u, self._union = self._union, None
        ws = dist.all_reduce(m.stat_tail, op=dist.ReduceOp.SUM, async_op=True)   # statistic increments + validity flag
        wmax = dist.all_reduce(m.max_radii2D, op=dist.ReduceOp.MAX, async_op=True)
        if u["event"] is not None:
            u["event"].synchronize()          # (recorded behind the mask exchange on the side stream: long done)
            K = int(self._union_k[0]) + 1
            torch.cuda.current_stream(m.flat.device).wait_stream(u["stream"])
        else:
            K = u["K"]
        if K:
            need = (K * W + 3) // 4 * 4
            buf = getattr(self, "_packed", None)
            if buf is None or buf.numel() < need or buf.device != m.flat.device:
                buf = self._packed = torch.empty((int(need * 1.25) + 1024,), dtype=torch.float32, device=m.flat.device)
            widths = (C.c_int32 * len(m.fields))(*[n for _, n in m.fields])
            stream = C.c_void_p(torch.cuda.current_stream(m.flat.device).cuda_stream) if m.flat.is_cuda else None
            opt.api.call("rows_pack", m.flat_grad.data_ptr(), P, len(m.fields), widths, m.union_mask.data_ptr(),
                         m.union_pos.data_ptr(), K, buf.data_ptr(), stream)
            wrows = dist.all_reduce(buf[:K * W], op=dist.ReduceOp.SUM, async_op=True)
        self.last_exchange = dict(union_rows=K, rows=P, sparse_bytes=4 * K * W + P + 4 * int(m.stat_tail.numel()),
                                  dense_bytes=4 * (n_grad + int(m.stat_tail.numel())))
        if gate is not None or opt.sparse:
            ws.wait()  # the gate is the reduced flag; sparse_adam: the summed visibility is the rows to step
        # The optimizer in two parts.  Rows OUTSIDE the union need nothing from the links: their summed gradient is zero on
        # every rank, so their (zero-gradient, gated) update runs NOW, on the compute stream, while the union's rows are on the
        # links - more than half of the dense Adam pass hides under the collective.  The union's rows follow when their sums
        # have been scattered back.  Every row is stepped exactly once with the same bias corrections: the same bits as one
        # dense pass (tests/test_dp_gloo.py compares with the all-reduce form).  sparse_adam keeps its single masked pass (its
        # rows - visible in SOME rank's view - are known from the summed statistics only).
        split = optimizer_step and not opt.sparse and self.SPLIT_SPARSE_ADAM
        if optimizer_step:
            opt.begin_step(skip)
        if split:
            opt.step_range(0, n_grad, skip, gate=gate, row_mask=m.union_rows_f[1])
        if K:
            wrows.wait()
            opt.api.call("rows_unpack", m.flat_grad.data_ptr(), P, len(m.fields), widths, m.union_mask.data_ptr(),
                         m.union_pos.data_ptr(), K, buf.data_ptr(), stream)
        if split:
            if K:
                opt.step_range(0, n_grad, skip, gate=gate, row_mask=m.union_rows_f[0])
        elif optimizer_step:
            opt.step_range(0, n_grad, skip, gate=gate, row_mask=m.stat_delta[1] if opt.sparse else None)
        ws.wait()
        wmax.wait()
        self._add_statistics(gate)

    def gather_optimizer_state(self):
        """Sharded optimizer only: every rank holds current Adam moments for ITS shard; before anything that needs them
        all (densification re-layout, checkpoint) they are all-gathered in place.  No-op otherwise."""
        m, opt = self.model, self.model.optimizer
        if not (self.sharded_optimizer and self.world_size > 1 and isinstance(opt, FlatAdam)) or self.sparse_exchange:
            return   # (all-reduce and sparse forms: every replica holds all moments)
        S = m.flat_padded.numel() // self.world_size
        for buf in (opt.exp_avg_padded, opt.exp_avg_sq_padded):
            dist.all_gather_into_tensor(buf, buf[self.rank * S:(self.rank + 1) * S])

    def _exchange_and_step_sharded(self, optimizer_step, skip=(), gate=None):
        """The sharded form of exchange_and_step: rank r owns elements [r S, (r+1) S) of the padded flat
        buffers.  reduce-scatter (SUM) leaves the summed gradients of its shard on each rank, Adam runs on that shard
        only (1/N of the 28 B/element pass), all-gather returns the updated parameters to every replica; the view
        statistics (8 B/Gaussian) and max_radii2D are all-reduced alongside.  Same bytes on the links as the all-reduce
        (which is a reduce-scatter + all-gather inside RCCL), 1/N of the optimizer pass.  Both collectives run in place
        on the padded buffers.  Replicas stay bit-identical: every parameter is computed once, by its owner.  With two
        ranks the sums equal the all-reduce path's bit for bit (a + b is commutative); with more the reduction order
        is the library's choice in both paths."""
        m, opt = self.model, self.model.optimizer
        world, rank = self.world_size, self.rank
        n, n_pad = m.flat.numel(), m.flat_padded.numel()
        S = n_pad // world
        lo, hi = rank * S, min((rank + 1) * S, n)
        gshard = m.grad_padded[rank * S:(rank + 1) * S]
        wg = dist.reduce_scatter_tensor(gshard, m.grad_padded, op=dist.ReduceOp.SUM, async_op=True)
        ws = dist.all_reduce(m.stat_tail, op=dist.ReduceOp.SUM, async_op=True)   # statistic increments + validity flag
        wmax = dist.all_reduce(m.max_radii2D, op=dist.ReduceOp.MAX, async_op=True)
        wg.wait()
        if gate is not None:
            ws.wait()  # the gate is the reduced flag
        if optimizer_step:
            opt.begin_step(skip)
            if hi > lo:
                opt.step_range(lo, hi, skip, grads=gshard, gate=gate)
            dist.all_gather_into_tensor(m.flat_padded, m.flat_padded[rank * S:(rank + 1) * S])
        ws.wait()
        wmax.wait()
        self._add_statistics(gate)


# ----------------------------------------------------------------------------------------------------------------
# multispectral (RGB + NIR) step: counterpart of LGDWT-GS/mult-dwtgs/train_nir.py:81-125 on the fused 4-channel pass
# ----------------------------------------------------------------------------------------------------------------
def render_rgb_nir(viewpoint_camera, pc, Settings, bg_color, scaling_modifier=1.0, antialiasing=False, debug=False,
                   two_pass_rasterizer=None, clamp=True, raw_activations=False, camera_key=None):
    """= render() + render_nir() of mult-dwtgs/gaussian_renderer/__init__.py:18-258: {render (clamped to [0,1] as
    :119 does), nir (channel 0 of the NIR pass, not clamped), viewspace_points, visibility_filter, radii, depth}.
    Default: ONE pass with the NIR albedo as a 4th blended channel (gsplat_amd.nir).  two_pass_rasterizer = a
    `GaussianRasterizer` class: the reference's two 3-channel passes through it (used as the parity target).
    raw_activations (the fused train step's form; the caller has told the backend, RasterBackend.raw_activations): the
    rasterizer gets the RAW scaling / rotation / opacity / albedo rows and the gain and activates them in its kernels.
    clamp=False: the fused criterion clamps (and differentiates the clamp of) the render itself."""
    from .nir import GaussianRasterizerX, nir_colors
    rs = Settings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=debug, antialiasing=antialiasing)
    if raw_activations:
        if two_pass_rasterizer is not None:
            raise RuntimeError("raw activations serve the fused 4-channel pass only")
        ssp = torch.empty_like(pc.get_xyz).requires_grad_(True)
        rast = GaussianRasterizerX(rs)
        if camera_key is not None:
            rast.camera_key, rast.camera_limits = camera_key, False
        color, radii, depth, nir_img = rast(
            means3D=pc.get_xyz, means2D=ssp, opacities=pc.params["opacity"], extra=pc.params["nir_albedo"],
            shs=pc.get_features, scales=pc.params["scaling"], rotations=pc.params["rotation"], extra_gain=pc.nir_gain)
        return {"render": color.clamp(0, 1) if clamp else color, "nir": nir_img, "viewspace_points": ssp,
                "viewspace_points_nir": None, "visibility_filter": None, "radii": radii, "depth": depth}
    act = pc.fused_activations()
    if act is None:
        act = (pc.get_scaling, pc.get_rotation, pc.get_opacity)
    scales, rotations, opacities = act
    nir = nir_colors(torch.sigmoid(pc.params["nir_albedo"]), pc.nir_gain)  # get_nir_albedo * clamp(gain), render_nir:166-169
    ssp = torch.zeros_like(pc.get_xyz, requires_grad=True)
    if two_pass_rasterizer is None:
        rast = GaussianRasterizerX(rs)
        if camera_key is not None:
            rast.camera_key, rast.camera_limits = camera_key, False
        color, radii, depth, nir_img = rast(
            means3D=pc.get_xyz, means2D=ssp, opacities=opacities, extra=nir, shs=pc.get_features, scales=scales,
            rotations=rotations)
        ssp2 = None
    else:
        rast = two_pass_rasterizer(raster_settings=rs)
        color, radii, depth = rast(means3D=pc.get_xyz, means2D=ssp, shs=pc.get_features, opacities=opacities,
                                   scales=scales, rotations=rotations)
        ssp2 = torch.zeros_like(pc.get_xyz, requires_grad=True)
        img3, _, _ = rast(means3D=pc.get_xyz, means2D=ssp2, shs=None, colors_precomp=nir[:, None].repeat(1, 3),
                          opacities=opacities, scales=scales, rotations=rotations)
        nir_img = img3[0:1]
    return {"render": color.clamp(0, 1) if clamp else color, "nir": nir_img, "viewspace_points": ssp,
            "viewspace_points_nir": ssp2, "visibility_filter": radii > 0, "radii": radii, "depth": depth}


class NirCriterion:
    """train_nir.py:88-104: (1 - l) L1 + l (1 - SSIM) on RGB, plus nir_weight * (L1 + 0.2 (1 - SSIM)) on the NIR image
    (mult-dwtgs/utils/loss_utils.py:93-144; the reference evaluates the single-channel SSIM on three identical
    copies, whose mean is the single-channel value).
    fused=True (device kernels available): both terms as FusedLGDWTLoss nodes - the RGB one on the un-clamped render
    (with the global / patch DWT terms when rgb_criterion asks for them), the NIR one through the same kernels on one
    channel with its own two weights (GsLgdwtParams.custom_base), no clamp; no torch pass over an image."""

    def __init__(self, ops, lambda_dssim=0.2, nir_weight=1.0, nir_l1_weight=1.0, nir_ssim_weight=0.2, rgb_criterion=None,
                 fused=False):
        """rgb_criterion: an LGDWTCriterion to use for the RGB term instead of train_nir.py's plain L1 + SSIM - the
        multispectral step WITH the global / patch DWT terms of LGDWT-GS/train.py:128-202 (BASELINE configs[4] names a
        patch-DWT loss; the reference's train_nir.py itself has none)."""
        from .losses import LGDWTCriterion
        self.ops, self.lambda_dssim, self.nir_weight = ops, lambda_dssim, nir_weight
        self.nir_l1_weight, self.nir_ssim_weight = nir_l1_weight, nir_ssim_weight
        self.rgb_criterion = rgb_criterion
        self.fused = bool(fused)
        if self.fused:
            if rgb_criterion is None:
                self.rgb_criterion = LGDWTCriterion(ops, lambda_dssim=lambda_dssim, dwt_enable=False, patch_dwt_enable=False)
            self._nir = LGDWTCriterion(ops, dwt_enable=False, patch_dwt_enable=False)
            self._nir.custom_base = (nir_weight * nir_l1_weight, nir_weight * nir_ssim_weight)
            self._nir.clamp = False

    # (the running mean of the RGB criterion's DWT scale: what a step that has to be taken back restores)
    @property
    def dwt_running_mean(self):
        return None if self.rgb_criterion is None else self.rgb_criterion.dwt_running_mean

    @dwt_running_mean.setter
    def dwt_running_mean(self, v):
        if self.rgb_criterion is not None:
            self.rgb_criterion.dwt_running_mean = v

    def elf_mask(self, gt_image):
        return self.rgb_criterion.elf_mask(gt_image)

    def fused_call(self, raw_image, gt_image, nir_pred, nir_gt, mask=None):
        """-> ((rgb_loss, nir_loss), parts): two scalars to seed the backward with (their sum is the step's loss)"""
        rgb, parts = self.rgb_criterion.fused_call(raw_image, gt_image, mask=mask)
        nir, nparts = self._nir.fused_call(nir_pred, nir_gt)
        parts = dict(parts)
        parts.update(rgb=rgb.detach(), nir=nir.detach(), nir_l1=nparts["l1"], nir_ssim=nparts["ssim"])
        return (rgb, nir), parts

    def __call__(self, image, gt_image, nir_pred, nir_gt, mask=None):
        o = self.ops
        if self.rgb_criterion is not None:
            rgb, _ = self.rgb_criterion(image, gt_image, mask=mask)
        else:
            l1 = o.l1_loss(image, gt_image)
            rgb = (1.0 - self.lambda_dssim) * l1 + self.lambda_dssim * (1.0 - o.ssim(image, gt_image))
        nir = self.nir_l1_weight * o.l1_loss(nir_pred, nir_gt) + self.nir_ssim_weight * (1.0 - o.ssim(nir_pred, nir_gt))
        return rgb + self.nir_weight * nir, dict(rgb=rgb.detach(), nir=nir.detach())


class TrainerNIR(Trainer):
    """Trainer for a model created with with_nir=True: one fused 4-channel pass per view, RGB + NIR losses, Adam over
    the 60-float rows (59 + raw NIR albedo) and the global gain.  With a fused NirCriterion on the GPU the step is the RGB
    trainer's: raw parameter rows, region-binned depth-limited lists with the deferred verdict, the criterion as two fused
    nodes, gs_backward_step_x (backward + activation backward + statistics + Adam over the six rows and the gain in the
    per-Gaussian kernels, two-phase with the side stream), hipGraph replay, and the data-parallel form."""
    RENDERS_NIR = True

    def __init__(self, model, cameras, gt_images, nir_images, criterion, Settings, bg, two_pass_rasterizer=None, **kw):
        from .nir import GaussianRasterizerX
        super().__init__(model, cameras, gt_images, criterion, None if two_pass_rasterizer is not None else GaussianRasterizerX,
                         Settings, bg, **kw)
        self.nirs = nir_images
        self.two_pass = two_pass_rasterizer

    def _fused_step_ok(self, backend, optimizer_step):
        return self.two_pass is None and getattr(self.criterion, "fused", False) and super()._fused_step_ok(backend, optimizer_step)

    def _fused_dp_ok(self, backend, optimizer_step):
        return self.two_pass is None and getattr(self.criterion, "fused", False) and super()._fused_dp_ok(backend, optimizer_step)

    def _render_view(self, ci, fused, raw):
        m = self.model
        if m.nir_gain.grad is not None:
            m.nir_gain.grad = None
        return render_rgb_nir(self.cameras[ci], m, self.Settings, self.bg, two_pass_rasterizer=self.two_pass,
                              clamp=not fused, raw_activations=raw,
                              camera_key=None if self.two_pass is not None else ("trainer", self.uid, ci))

    def _criterion_backward(self, pkg, ci, mask, fused, side_launch):
        kw = {} if mask is None else {"mask": mask}
        if fused:
            (rgb, nir), parts = self.criterion.fused_call(pkg["render"], self.gts[ci], pkg["nir"], self.nirs[ci], **kw)
            self._arm_side_launch(side_launch)
            unit = self.criterion.ops.unit_grad(rgb.device)
            torch.autograd.backward([rgb, nir], [unit, unit])
            with torch.no_grad():
                loss = rgb + nir
        else:
            loss, parts = self.criterion(pkg["render"], self.gts[ci], pkg["nir"], self.nirs[ci], **kw)
            loss.backward()
        return loss, parts

    def _unfused_tail(self, pkg, radii, optimizer_step, skip):
        m = self.model
        self.last_radii = radii
        with torch.no_grad():
            m.collect_grads()
            vg = pkg["viewspace_points"].grad
            if pkg["viewspace_points_nir"] is not None and pkg["viewspace_points_nir"].grad is not None:
                vg = vg + pkg["viewspace_points_nir"].grad
            m.update_view_statistics(radii, vg.contiguous(), into_delta=self.world_size > 1)
            if self.world_size > 1:
                dist.all_reduce(m.nir_gain.grad, op=dist.ReduceOp.SUM)
                torch.gt(radii, 0, out=m.grad_mask.view(torch.bool))
            self.exchange_and_step(optimizer_step, skip)
            if optimizer_step:
                # the reference keeps the global gain in the main Adam (mult-dwtgs/scene/gaussian_model.py:266-280): it
                # is stepped whenever optimizer.step() runs - and, like every group, not on a densification iteration.
                # Its step count is the main optimizer's (FlatAdam.seg_steps["nir_gain"], advanced by the step above)
                opt = m.optimizer
                if isinstance(opt, FlatAdam):
                    m.nir_gain_optimizer.state[m.nir_gain]["step"].fill_(float(opt.seg_steps["nir_gain"] - 1))
                m.nir_gain_optimizer.step()

    def exchange_and_step(self, optimizer_step, skip=(), gate=None):
        super().exchange_and_step(optimizer_step, skip, gate)
        if gate is not None and optimizer_step and self.world_size > 1:
            # fused data-parallel step: the gain's gradient travelled in the exchange tail (summed with the flag); its
            # Adam step, gated like every other row's, on every rank
            m, opt = self.model, self.model.optimizer
            gs = m.nir_gain_optimizer.state[m.nir_gain]
            segs = (opt._Seg * 1)()
            segs[0].begin, segs[0].end, segs[0].lr_a, segs[0].step = 0, 1, opt.lr["nir_gain"], opt.seg_steps["nir_gain"]
            gbuf = getattr(self, "_gain_grad_buf", None)   # (the kernel wants 16-byte aligned pointers: the word is not)
            if gbuf is None:
                gbuf = self._gain_grad_buf = torch.zeros((4,), dtype=torch.float32, device=m.flat.device)
            gbuf[:1].copy_(m.gain_grad_word)
            opt.api.call("adam_step_gated", m.nir_gain.data_ptr(), gbuf.data_ptr(), gs["exp_avg"].data_ptr(),
                         gs["exp_avg_sq"].data_ptr(), 1, segs, 1, opt.betas[0], opt.betas[1], opt.eps, opt.t,
                         gate.data_ptr(), _stream_of(m.flat))


# ----------------------------------------------------------------------------------------------------------------
# hipGraph replay of the steady-state single-GPU step
# ----------------------------------------------------------------------------------------------------------------
def _status_check(tag, num_rendered, overflow, trunc_failed):
    """include/gsplat.h GS_STATUS_CHECK"""
    m = 0xFFFFFFFF
    return (((tag & m) * 2654435761) & m) ^ (((num_rendered & m) * 40503) & m) ^ ((overflow << 30) & m) ^ ((trunc_failed << 31) & m) \
        ^ 0x5bd1e995


class GraphedStep:
    """Captures the train step of a Trainer (forward, fused criterion, gs_backward_step) into hipGraphs - ONE PER CAMERA -
    and replays them: the kernel launches of a step, their Python glue and the forward's host wait become one graph launch.
    Everything that belongs to a camera is baked into its own graph - matrices, ground truth, patch mask, the criterion's
    per-camera accumulator and the backend's per-camera tile order and depth limits (the same entry the eager step of that
    camera uses, RasterBackend.camera_entry, read and rewritten in place by the captured kernels) - so a replay copies
    nothing: only Adam's step-dependent constants (`coef`, GsStepState.coef_dev) and the replay's tag are uploaded.  (One
    graph fed through static buffers copied the 25 MB ground truth and a dozen small tensors per replay: 35 us of a 1.1 ms
    step at 1080p, most of a C1 step.)  The graphs share one memory pool: replays run one at a time on one stream and no
    intermediate outlives its replay.
    Everything frozen at capture is checked before a replay (model buffers, image size, field of view, active SH degree,
    the camera's tensors, the binning capacity); when it no longer holds the camera is captured again.  A view that needs
    more binning capacity than was captured, or whose depth limits failed, did nothing on the device (gs_backward_step
    skips on either flag): the step is then taken eagerly (settle).  SINGLE-THREADED by contract: while a capture is open
    nothing else in the process may free device memory (the cyclic collector is fenced in _capture; a tensor another thread
    drops by refcount is not).  Results are those of the eager fused step: same
    kernels, same arguments."""

    def __init__(self, trainer, capacity_margin=1.5, warmup=3, capacity=None):
        """capacity: binning capacity (instances) to capture with; default = capacity_margin x the largest view the
        backend has seen recently.  warmup: un-captured steps before the FIRST capture (allocator and library state
        settle); every later camera takes one."""
        self.tr = trainer
        trainer._graphed = self   # Trainer.sync() settles the last replay too
        self.fixed_capacity = capacity
        self.capacity_margin = capacity_margin
        self.warmup = warmup
        self.graphs = {}          # camera index -> dict(graph, key, loss, rm_before)
        self.pool = None
        self.capacity = None
        self.coef = None
        self.replays = self.eager_steps = self.captures = 0
        self.fail_kinds = {"limits": 0, "capacity": 0}   # why replays had to be repeated eagerly
        self._replay_pending = None
        self.s_loss = None

    # (compatibility: "is anything captured", the key of the last camera stepped)
    @property
    def graph(self):
        return next(iter(self.graphs.values()))["graph"] if self.graphs else None

    # -- what a capture froze
    def _key(self, ci):
        tr = self.tr
        m, cam = tr.model, tr.cameras[ci]
        mask = None if tr.masks is None else tr.masks[ci]
        # m.generation: a re-layout / restore / load_ply with the SAME number of Gaussians still replaces every buffer the
        # captured kernels point into; the statistic tensors are re-created by densify_and_prune on their own
        return (m.P, m.generation, m.denom.data_ptr(), m.max_radii2D.data_ptr(), m.xyz_gradient_accum.data_ptr(),
                m.active_sh_degree, int(cam.image_height), int(cam.image_width), float(cam.FoVx), float(cam.FoVy),
                cam.world_view_transform.data_ptr(), cam.full_proj_transform.data_ptr(), cam.camera_center.data_ptr(),
                tr.gts[ci].data_ptr(), None if mask is None else mask.data_ptr(), self.capacity, bool(tr.depth_limit),
                self._entry_ptrs(ci), self._backend().rows_epoch,
                m.optimizer.dormant_flags().data_ptr() if m.optimizer.USE_DORMANT else None)

    def _entry_ptrs(self, ci):
        """The addresses a capture of camera `ci` bakes in from the backend's per-camera entry (tile order, depth limits,
        slack): the entry the backend holds NOW - should it ever be replaced, the key changes and the camera is captured again."""
        ent = self.camera_entry(ci)
        return None if ent is None else (ent["order"].data_ptr(), ent["limit"].data_ptr(), ent["slack"].data_ptr())

    def _drop_graph(self, ci):
        g = self.graphs.pop(ci, None)
        if g is not None and g.get("entry") is not None:
            g["entry"]["pinned"] = max(0, g["entry"].get("pinned", 0) - 1)

    def _drop_all_graphs(self):
        for ci in list(self.graphs):
            self._drop_graph(ci)

    def _backend(self):
        tr = self.tr
        be = getattr(getattr(getattr(tr.Rasterizer, "_fn", None), "_impl", None), "backend", None)
        return be

    def camera_entry(self, ci):
        """The backend's per-camera state (tile order, depth limits) of camera `ci` - shared with the eager step."""
        tr = self.tr
        cam = tr.cameras[ci]
        return self._backend().camera_entry(int(cam.image_width), int(cam.image_height), camera_key=("trainer", tr.uid, ci),
                                            device_index=tr.model.flat.device.index)

    def _shared_init(self, dev):
        be = self._backend()
        self.coef = torch.zeros((15,), dtype=torch.float32, device=dev)
        self.coef_host = torch.zeros((15,), dtype=torch.float32).pin_memory()
        be._pinned_by_device.setdefault((dev.index, "static"), torch.zeros((16,), dtype=torch.int32).pin_memory())
        # the replay's identity: uploaded before every replay, copied out with the status words by the captured
        # gs_forward_status (GsScratch.step_tag) - the host polls the pinned block for it instead of draining the stream
        self.s_tag = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.tag_host = torch.zeros((1,), dtype=torch.int32).pin_memory()
        self._tag = 0
        self._tag_event = None
        self._coef_event = None
        crit = self.tr.criterion
        if getattr(crit, "dwt_running_mean", None) is None and hasattr(crit, "dwt_running_mean"):
            crit.dwt_running_mean = torch.ones((1,), dtype=torch.float32, device=dev)

    def _one_step(self, ci):
        """One step of camera `ci` in the capture-safe form (fixed capacity, no host wait, constants from `coef`)."""
        tr, be = self.tr, self._backend()
        be.static_capacity = self.capacity
        be.static_step_tag = self.s_tag
        be.depth_limit_request = "graph" if tr.depth_limit else None
        tr._coef_dev = self.coef
        try:
            return tr._step_camera(ci, True, ())
        finally:
            be.static_capacity = None
            be.static_step_tag = None
            be.depth_limit_request = None
            tr._coef_dev = None

    def _capture(self, ci):
        import gc
        tr, be = self.tr, self._backend()
        m = tr.model
        dev = m.flat.device
        if self.coef is None:
            self._shared_init(dev)
        first = not self.graphs
        if first or self.capacity is None:
            # binning capacity: the largest view seen so far with head-room (one number for every camera's graph)
            self.capacity = int(self.fixed_capacity) if self.fixed_capacity is not None else \
                int(max(be._capacity_hint, 4096) * self.capacity_margin)
        crit = tr.criterion
        opt = m.optimizer
        counters = (opt.t, dict(opt.seg_steps))
        rm0 = None if getattr(crit, "dwt_running_mean", None) is None else crit.dwt_running_mean.clone()
        # warm-up on a side stream, then capture; every one of these runs is a real train step of camera `ci`: counters
        # and parameters advance as in eager mode.  The first capture settles allocator and library state; a later camera
        # takes one un-captured step, which creates what a step creates lazily per camera (the backend's entry, the
        # criterion's accumulator - nothing may come from host memory inside a capture) and gives the capture a tile-order
        # hint to bake in.
        n_warm = max(1, self.warmup) if first else 1
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(n_warm):
                self._coef_for_next()  # (_step_camera's fused_request advances the counters itself)
                self._one_step(ci)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        fits = self._view_ok(be)
        if fits:   # the warm-up steps happened: what a failed first replay has to put back is the state after them
            counters = (opt.t, dict(opt.seg_steps))
            rm0 = None if rm0 is None else crit.dwt_running_mean.clone()
        if fits:
            # A cyclic-garbage sweep INSIDE the capture may free tensors of an earlier capture's pool (autograd contexts are
            # cycles) - a device free while the stream is capturing aborts the process.  So: this camera's old graph goes
            # first, garbage is collected now, and the collector stays off until the capture has ended.
            self._drop_graph(ci)
            gc.collect()
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            self._coef_for_next()
            if opt.USE_DORMANT:
                opt.dormant_flags()   # (derived from the moments by torch kernels: now, not inside the capture)
            gc_was_on = gc.isenabled()
            gc.disable()
            try:
                with torch.cuda.graph(graph, pool=self.pool):
                    loss = self._one_step(ci)
            finally:
                if gc_was_on:
                    gc.enable()
            if self.pool is None:
                self.pool = graph.pool()
            rm_before = (tr.last.get("parts") or {}).get("running_mean_before")  # (a view of the captured output)
            # the capture itself launched nothing: replay once so that this call ends with a step
            self._upload_tag()
            graph.replay()
            torch.cuda.synchronize(dev)
            fits = self._view_ok(be)
        if not fits:
            # this view needs more instances than the capacity (or its depth limits were stale): none of the steps above
            # changed anything on the device (gs_backward_step skips on either flag).  Put the counters back and take ONE
            # eager step instead.
            self._invalidate(ci, be.last_status())
            opt.t, opt.seg_steps = counters[0], dict(counters[1])
            if rm0 is not None:
                crit.dwt_running_mean.copy_(rm0)
            self.eager_steps += 1
            self.s_loss = tr._step_camera(ci, True, ())
            return self.s_loss
        # the graph holds the addresses of this camera's entry: the entry is pinned (never evicted from the backend's cache)
        # and referenced from here for as long as the graph lives
        ent = self.camera_entry(ci)
        if ent is not None:
            ent["pinned"] = ent.get("pinned", 0) + 1
        self.graphs[ci] = dict(graph=graph, key=self._key(ci), loss=loss, rm_before=rm_before, entry=ent)
        self.captures += 1
        self.s_loss = loss
        return loss

    def _invalidate(self, ci, status):
        """After a step that did nothing on the device.  Limits that failed: the camera forgets them (its graph stays - the
        eager step that follows measures new ones into the same buffers).  A view that did not fit the capacity: every
        graph was captured with that capacity - all go, the next capture sizes it from the backend's grown hint."""
        num_rendered, overflow, trunc_failed = status
        self.fail_kinds["limits" if trunc_failed else "capacity"] += 1
        be = self._backend()
        ent = self.camera_entry(ci)
        if trunc_failed and ent is not None:
            ent["limit_ok"] = False
            ent["limit"].fill_(float("inf"))
            be.depth_limit_stats["failed"] += 1
            be.limits_failed(ent)     # (this camera's exported bounds get more slack: RasterBackend.SLACK)
        if overflow or num_rendered > self.capacity:
            self._drop_all_graphs()
            self.capacity = None   # (the eager step that follows grows the backend's hint; the next capture reads it)

    def _view_ok(self, be):
        num_rendered, overflow, trunc_failed = be.last_status()
        return num_rendered <= self.capacity and not overflow and not trunc_failed

    def _coef_for_next(self):
        """Constants of the step the NEXT launch performs: the launch (eager inside _one_step, or a replay) is step t + 1."""
        opt = self.tr.model.optimizer
        saved_t, saved_seg = opt.t, dict(opt.seg_steps)
        opt.begin_step(())
        coefs = torch.from_numpy(opt.step_coefficients(()))
        opt.t, opt.seg_steps = saved_t, saved_seg
        if self._coef_event is not None:
            self._coef_event.synchronize()  # the previous upload has read the pinned staging buffer
        self.coef_host.copy_(coefs)
        self.coef.copy_(self.coef_host, non_blocking=True)
        self._coef_event = torch.cuda.Event()
        self._coef_event.record(torch.cuda.current_stream(self.coef.device))

    def _upload_tag(self):
        self._tag = (self._tag + 1) & 0x7FFFFFFF
        if self._tag_event is not None:
            self._tag_event.synchronize()  # the previous upload has read the pinned word
        self.tag_host[0] = self._tag
        self.s_tag.copy_(self.tag_host, non_blocking=True)
        self._tag_event = torch.cuda.Event()
        self._tag_event.record(torch.cuda.current_stream(self.s_tag.device))

    def settle(self):
        """The verdict of the last replay (did the view fit the captured capacity, did its depth limits hold?), read from
        the status words the replay's forward copied out.  The host does not drain the stream for it: it polls the pinned
        block for the replay's tag, which arrives when the FORWARD of that replay has finished - the rest of the step is
        still queued behind it, so the GPU never waits for the host.  A failed replay changed nothing on the device
        (gs_backward_step is a no-op on either flag): counters and the criterion's running mean are put back and the step
        is taken eagerly."""
        p, self._replay_pending = getattr(self, "_replay_pending", None), None
        if p is None:
            return
        tr, be = self.tr, self._backend()
        st = be._pinned_by_device[(self.s_tag.device.index, "static")]
        spins = 0
        while int(st[8]) != p["tag"]:
            spins += 1
            if spins > 2000:  # (not there after ~2 ms of polling: wait for the stream the ordinary way)
                torch.cuda.current_stream(self.s_tag.device).synchronize()
                if int(st[8]) != p["tag"]:
                    raise RuntimeError("replay %d never delivered its status (block holds tag %d)" % (p["tag"], int(st[8])))
                break
        # The tag and the status words arrive with ONE 48-byte device-to-host copy, but nothing promises a host poller that
        # the bytes of a copy land in address order: the block is accepted when its check word (written by the same kernel
        # that wrote the tag, include/gsplat.h GS_STATUS_CHECK) matches the words read
        def read():
            w = (int(st[0]), int(st[1]), int(st[2]), int(st[8]), int(st[9]) & 0xFFFFFFFF)
            return w, w[3] == p["tag"] and w[4] == _status_check(w[3], w[0], w[1], w[2])
        seen, ok = read()
        spins = 0
        while not ok:
            spins += 1
            if spins == 2000:
                torch.cuda.current_stream(self.s_tag.device).synchronize()
            if spins > 2001:
                raise RuntimeError("replay %d delivered an inconsistent status block %r" % (p["tag"], seen))
            seen, ok = read()
        num_rendered, overflow, trunc_failed = seen[0], seen[1], seen[2]
        if num_rendered <= p["capacity"] and not overflow and not trunc_failed:
            if tr.depth_limit:
                ent = self.camera_entry(p["ci"])
                if ent is not None and ent["limit_ok"]:
                    be.limits_held(ent)
            return
        ci = p["ci"]
        self._invalidate(ci, (num_rendered, overflow, trunc_failed))
        opt = tr.model.optimizer
        opt.t -= 1
        for name in opt.seg_steps:
            opt.seg_steps[name] -= 1
        crit = tr.criterion
        if p["rm_before"] is not None:  # the criterion's running mean saw the loss of an un-rendered image
            crit.dwt_running_mean.copy_(p["rm_before"])
        self.eager_steps += 1
        self.s_loss = tr._step_camera(ci, True, ()).clone()

    def step(self, k):
        tr = self.tr
        tr.sync()
        self.settle()
        ci = tr.camera_index(k)
        be = self._backend()
        if tr.model.exposure is not None or not tr._fused_step_ok(be, True) or be._capacity_hint <= 0 or \
                tr._depth_prior(ci) is not None:
            # what the capture cannot hold (a torch optimizer for the exposure, N > 1, the depth term's weight - a kernel
            # argument that changes every iteration - ...), or no view has been rendered yet to size the binning capacity from
            self.eager_steps += 1
            self.s_loss = tr._step_camera(ci, True, ())
            return self.s_loss
        g = self.graphs.get(ci)
        if g is None or g["key"] != self._key(ci):
            return self._capture(ci)
        self._coef_for_next()
        self._upload_tag()
        if tr.model.optimizer.USE_DORMANT:
            tr.model.optimizer.dormant_flags()   # (in place: the graph holds the tensor's address; a no-op while they are current)
        g["graph"].replay()
        if tr.depth_limit:
            ent = self.camera_entry(ci)
            if ent is not None and ent["limit_ok"]:
                be.depth_limit_stats["used"] += 1
        tr.model.optimizer.begin_step(())
        self.replays += 1
        self._replay_pending = dict(tag=self._tag, ci=ci, capacity=self.capacity, rm_before=g["rm_before"])
        self.s_loss = g["loss"]
        return self.s_loss

    def sync(self):
        """Settle everything that is pending (an eager deferred verdict, the last replay's): call before reading the model."""
        self.tr.sync()
        self.settle()
