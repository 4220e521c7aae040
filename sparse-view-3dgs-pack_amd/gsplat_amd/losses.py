"""Autograd wrappers over the loss kernels of the C ABI (include/gsplat.h).

`LossOps(api)` mirrors, name for name, the functions of the reference's
LGDWT-GS/utils/loss_utils.py (l1_loss :40-41, ssim :56-86 / fused_ssim, get_dwt_subbands :106-153,
compute_elf_map :336-366, compute_patch_dwt_loss :368-442) plus the fused global DWT loss of
LGDWT-GS/train.py:132-164 and the loss composition of train.py:188-202.
All tensors stay on the device; no call synchronises the host.
"""
import ctypes as C

import torch

BANDS = ("LL1", "LH1", "HL1", "HH1", "LL2", "LH2", "HL2", "HH2")


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream) if t.is_cuda else None


def _c(t):
    t = t.contiguous()
    return t if t.dtype == torch.float32 else t.float()


def _p(t):
    return None if t is None else t.data_ptr()


class LossOps:
    def __init__(self, api):
        self.api = api
        self._unit = {}   # device -> the constant 1 handed to autograd as the seed gradient (unit_grad)
        ops = self

        class _L1(torch.autograd.Function):
            @staticmethod
            def forward(ctx, a, b):
                a, b = _c(a), _c(b)
                s = torch.zeros((1,), dtype=torch.float32, device=a.device)
                ops.api.call("l1_fwd", a.data_ptr(), b.data_ptr(), a.numel(), s.data_ptr(), _stream(a))
                ctx.save_for_backward(a, b)
                return (s / a.numel()).reshape(())

            @staticmethod
            def backward(ctx, g):
                a, b = ctx.saved_tensors
                # grad = sign(a - b) * (g / n), the coefficient read on the device (no host sync, one pass over the image)
                coef = (g / a.numel()).reshape(1).contiguous()
                ga = torch.empty_like(a)
                if hasattr(ops.api, "_l1_bwd_dev"):
                    ops.api.call("l1_bwd_dev", a.data_ptr(), b.data_ptr(), a.numel(), coef.data_ptr(), ga.data_ptr(), 0, _stream(a))
                else:
                    ops.api.call("l1_bwd", a.data_ptr(), b.data_ptr(), a.numel(), 1.0, ga.data_ptr(), 0, _stream(a))
                    ga = ga * coef
                return ga, (-ga if ctx.needs_input_grad[1] else None)

        class _Haar(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x):
                x = _c(x)
                N, Cc, H, W = x.shape
                h, w = (H + 1) // 2, (W + 1) // 2
                outs = [torch.empty((N, Cc, h, w), dtype=torch.float32, device=x.device) for _ in range(4)]
                ops.api.call("dwt_haar_fwd", x.data_ptr(), N * Cc, H, W, *[o.data_ptr() for o in outs], _stream(x))
                ctx.shape = (N, Cc, H, W)
                return tuple(outs)

            @staticmethod
            def backward(ctx, dll, dlh, dhl, dhh):
                N, Cc, H, W = ctx.shape
                gs = [None if g is None else _c(g) for g in (dll, dlh, dhl, dhh)]
                ref = next(g for g in gs if g is not None)
                dx = torch.empty((N, Cc, H, W), dtype=torch.float32, device=ref.device)
                ops.api.call("dwt_haar_bwd", *[_p(g) for g in gs], N * Cc, H, W, dx.data_ptr(), _stream(ref))
                return dx

        class _Dwt2L1(torch.autograd.Function):
            """sum_b w_b * mean|band_b(pred) - band_b(gt)| over the 8 sub-bands, one fused pass."""

            @staticmethod
            def forward(ctx, pred, gt, weights):
                pred, gt = _c(pred), _c(gt)
                Cc, H, W = pred.shape[-3:]
                sums = torch.zeros((8,), dtype=torch.float32, device=pred.device)
                ops.api.call("dwt2_l1_fwd", pred.data_ptr(), gt.data_ptr(), Cc, H, W, sums.data_ptr(), _stream(pred))
                h1, w1 = (H + 1) // 2, (W + 1) // 2
                h2, w2 = (h1 + 1) // 2, (w1 + 1) // 2
                counts = torch.tensor([Cc * h1 * w1] * 4 + [Cc * h2 * w2] * 4, dtype=torch.float32, device=pred.device)
                wdev = weights.to(device=pred.device, dtype=torch.float32)
                means = sums / counts
                ctx.save_for_backward(pred, gt, wdev / counts)
                ctx.mark_non_differentiable(means)
                return (means * wdev).sum(), means

            @staticmethod
            def backward(ctx, g, _gm):
                pred, gt, coef = ctx.saved_tensors
                coef = (coef * g).contiguous()
                Cc, H, W = pred.shape[-3:]
                grad = torch.empty_like(pred)
                ops.api.call("dwt2_l1_bwd", pred.data_ptr(), gt.data_ptr(), Cc, H, W, coef.data_ptr(), grad.data_ptr(),
                             0, _stream(pred))
                return grad, None, None

        class _PatchDwt(torch.autograd.Function):
            @staticmethod
            def forward(ctx, pred, gt, mask, ps, w3):
                pred, gt = _c(pred), _c(gt)
                Cc, H, W = pred.shape[-3:]
                sums = torch.zeros((3,), dtype=torch.float32, device=pred.device)
                ops.api.call("patch_dwt_fwd", pred.data_ptr(), gt.data_ptr(), Cc, H, W, ps, mask.data_ptr(),
                             sums.data_ptr(), _stream(pred))
                hp = (ps + 1) // 2
                n_sel = mask.sum().to(torch.float32)
                denom = torch.clamp_min(n_sel * (Cc * hp * hp), 1.0)
                w3 = w3.to(device=pred.device, dtype=torch.float32)
                ctx.save_for_backward(pred, gt, mask, w3 / denom)
                ctx.ps = ps
                return (sums / denom * w3).sum()

            @staticmethod
            def backward(ctx, g):
                pred, gt, mask, coef = ctx.saved_tensors
                coef = (coef * g).contiguous()
                Cc, H, W = pred.shape[-3:]
                grad = torch.empty_like(pred)
                ops.api.call("patch_dwt_bwd", pred.data_ptr(), gt.data_ptr(), Cc, H, W, ctx.ps, mask.data_ptr(),
                             coef.data_ptr(), grad.data_ptr(), 0, _stream(pred))
                return grad, None, None, None, None

        class _SSIMMap(torch.autograd.Function):
            # FusedSSIMMap, fused-ssim/fused_ssim/__init__.py:8-32
            @staticmethod
            def forward(ctx, C1, C2, img1, img2, padding="same", train=True):
                img1, img2 = _c(img1), _c(img2)
                B, Cc, H, W = img1.shape
                smap = torch.empty_like(img1)
                if train:
                    d1, d2, d3 = torch.empty_like(img1), torch.empty_like(img1), torch.empty_like(img1)
                else:
                    d1 = d2 = d3 = None
                ops.api.call("ssim_fwd", img1.data_ptr(), img2.data_ptr(), B, Cc, H, W, float(C1), float(C2),
                             smap.data_ptr(), _p(d1), _p(d2), _p(d3), _stream(img1))
                if padding == "valid":
                    smap = smap[:, :, 5:-5, 5:-5]
                if train:
                    ctx.save_for_backward(img1.detach(), img2, d1, d2, d3)
                ctx.C1, ctx.C2, ctx.padding, ctx.train = C1, C2, padding, train
                return smap

            @staticmethod
            def backward(ctx, opt_grad):
                if not ctx.train:
                    raise RuntimeError("fused_ssim(train=False) cannot be differentiated")
                img1, img2, d1, d2, d3 = ctx.saved_tensors
                dL_dmap = opt_grad
                if ctx.padding == "valid":
                    dL_dmap = torch.zeros_like(img1)
                    dL_dmap[:, :, 5:-5, 5:-5] = opt_grad
                dL_dmap = _c(dL_dmap)
                B, Cc, H, W = img1.shape
                grad = torch.empty_like(img1)
                ops.api.call("ssim_bwd", img1.data_ptr(), img2.data_ptr(), B, Cc, H, W, float(ctx.C1), float(ctx.C2),
                             dL_dmap.data_ptr(), d1.data_ptr(), d2.data_ptr(), d3.data_ptr(), grad.data_ptr(),
                             _stream(img1))
                return None, None, grad, None, None, None

        class _DepthL1(torch.autograd.Function):
            """Ll1depth_pure of LGDWT-GS/train.py:204-216: mean |(invDepth - mono_invdepth) * depth_mask|."""

            @staticmethod
            def forward(ctx, invdepth, mono, mask):
                d, m = _c(invdepth), _c(mono)
                k = None if mask is None else _c(mask)
                n = d.numel()
                part = torch.empty((int(ops.api.raw("depth_l1_partials_count")(n)),), dtype=torch.float32, device=d.device)
                ops.api.call("depth_l1", d.data_ptr(), m.data_ptr(), _p(k), n, part.data_ptr(), 0.0, None, None, _stream(d))
                ctx.save_for_backward(d, m, k)
                return (part.sum() / n).reshape(())

            @staticmethod
            def backward(ctx, g):
                d, m, k = ctx.saved_tensors
                coef = g.reshape(1).to(torch.float32).contiguous()
                grad = torch.empty_like(d)
                ops.api.call("depth_l1", d.data_ptr(), m.data_ptr(), _p(k), d.numel(), None, 1.0 / d.numel(), coef.data_ptr(),
                             grad.data_ptr(), _stream(d))
                return grad, None, None

        self._L1, self._Haar, self._Dwt2L1, self._PatchDwt, self._SSIMMap = _L1, _Haar, _Dwt2L1, _PatchDwt, _SSIMMap
        self._DepthL1 = _DepthL1

    # ---------------------------------------------------------------- loss_utils.py names
    def unit_grad(self, device):
        """A cached scalar 1 on `device`: `torch.autograd.backward(loss, ops.unit_grad(loss.device))` instead of
        `loss.backward()` saves the fill kernel autograd launches for its implicit seed, and FusedLGDWTLoss recognises
        the tensor and skips its own multiply by the upstream gradient."""
        device = torch.device(device)
        u = self._unit.get(device)
        if u is None:
            u = self._unit[device] = torch.ones((), dtype=torch.float32, device=device)
        return u

    def l1_loss(self, network_output, gt):
        return self._L1.apply(network_output, gt)

    def depth_l1(self, invdepth, mono_invdepth, depth_mask=None):
        """The pure depth-regularisation term of LGDWT-GS/train.py:204-216 on the rasterizer's inverse-depth output:
        torch.abs((invDepth - mono_invdepth) * depth_mask).mean(); the caller multiplies by depth_l1_weight(iteration)."""
        return self._DepthL1.apply(invdepth, mono_invdepth, depth_mask)

    def depth_l1_step(self, invdepth, mono_invdepth, depth_mask, weight):
        """The same term for a train step that drives the backward itself: ONE launch -> (weight * term, dL/dinvDepth of it)."""
        d, m = _c(invdepth), _c(mono_invdepth)
        k = None if depth_mask is None else _c(depth_mask)
        n = d.numel()
        part = torch.empty((int(self.api.raw("depth_l1_partials_count")(n)),), dtype=torch.float32, device=d.device)
        grad = torch.empty_like(d)
        self.api.call("depth_l1", d.data_ptr(), m.data_ptr(), _p(k), n, part.data_ptr(), float(weight) / n, None, grad.data_ptr(),
                      _stream(d))
        return part.sum() * (float(weight) / n), grad

    def fused_ssim(self, img1, img2, padding="same", train=True):
        """fused-ssim/fused_ssim/__init__.py:34-41: img [B,C,H,W] -> scalar mean SSIM."""
        assert padding in ("same", "valid")
        return self._SSIMMap.apply(0.01 ** 2, 0.03 ** 2, img1, img2, padding, train).mean()

    def ssim(self, img1, img2, window_size=11, size_average=True):
        """loss_utils.ssim signature ([C,H,W] or [B,C,H,W] images)."""
        if window_size != 11:
            raise NotImplementedError("only the 11-tap window of the reference is implemented")
        a = img1 if img1.dim() == 4 else img1.unsqueeze(0)
        b = img2 if img2.dim() == 4 else img2.unsqueeze(0)
        m = self._SSIMMap.apply(0.01 ** 2, 0.03 ** 2, a, b, "same", True)
        return m.mean() if size_average else m.mean(1).mean(1).mean(1)

    def dwt_haar(self, x):
        """One analysis level: x[N,C,H,W] -> (LL, LH, HL, HH)."""
        return self._Haar.apply(x)

    def get_dwt_subbands(self, x):
        LL1, LH1, HL1, HH1 = self.dwt_haar(x)
        LL2, LH2, HL2, HH2 = self.dwt_haar(LL1)
        return {"LL1": LL1, "LH1": LH1, "HL1": HL1, "HH1": HH1, "LL2": LL2, "LH2": LH2, "HL2": HL2, "HH2": HH2}

    def dwt_l1_loss(self, pred, gt, weights):
        """Fused train.py:132-164: weights = 8 floats in BANDS order.  Returns (loss, per-band means)."""
        w = torch.as_tensor(weights, dtype=torch.float32)
        if pred.dim() == 4:
            assert pred.shape[0] == 1, "batch of one image, as in the reference training loop"
            pred, gt = pred[0], gt[0]
        return self._Dwt2L1.apply(pred, gt, w)

    def compute_elf_map(self, image):
        image = _c(image)
        N, Cc, H, W = image.shape
        out = torch.empty((N, 1, H, W), dtype=torch.float32, device=image.device)
        low = torch.empty(((H + 1) // 2, (W + 1) // 2), dtype=torch.float32, device=image.device)
        for n in range(N):
            self.api.call("elf_map", image[n].data_ptr(), Cc, H, W, low.data_ptr(), out[n].data_ptr(), _stream(image))
        return out

    def patch_mask(self, elf_map, patch_size=128, percentile=0.2):
        """Selection rule of compute_patch_dwt_loss (loss_utils.py:391-417): uint8 mask[L] on device."""
        elf_map = _c(elf_map)
        H, W = elf_map.shape[-2:]
        L = (H // patch_size) * (W // patch_size)
        means = torch.empty((L,), dtype=torch.float32, device=elf_map.device)
        self.api.call("patch_means", elf_map.data_ptr(), H, W, patch_size, means.data_ptr(), _stream(elf_map))
        k = int(means.numel() * (1.0 - percentile))
        k = max(1, k)
        k = min(k, means.numel())
        threshold, _ = torch.kthvalue(means, k)
        return (means >= threshold).to(torch.uint8), means

    def compute_patch_dwt_loss(self, pred, gt, elf_map, patch_size=128, percentile=0.2, lh1_weight=1.0,
                               hl1_weight=1.0, mask=None):
        N, Cc, H, W = pred.shape
        if H < patch_size or W < patch_size:
            return torch.tensor(0.0, device=pred.device)
        assert N == 1, "batch of one image, as in the reference training loop"
        if mask is None:
            mask, _ = self.patch_mask(elf_map, patch_size, percentile)
        w3 = torch.tensor([lh1_weight, hl1_weight, 0.5 * (lh1_weight + hl1_weight)], dtype=torch.float32)
        return self._PatchDwt.apply(pred[0], gt[0], mask, int(patch_size), w3)


class FusedLGDWTLoss(torch.autograd.Function):
    """The whole criterion of LGDWT-GS/train.py:128-202 as ONE autograd node on the un-clamped render.

    forward : clamp(0,1) -> L1 sum, SSIM sum (+ the three derivative maps), eight DWT band sums, three patch sums
              -> gs_lgdwt_combine (loss, running-mean DWT scale and the backward coefficients, all on the device)
    backward: three kernels accumulate into ONE image-gradient buffer; the last one folds in the clamp mask.
    No torch elementwise pass over an image, no host synchronisation."""

    @staticmethod
    def forward(ctx, ops, raw, gt, mask, sums, running_mean, params):
        """sums: the camera's persistent 16-float accumulator (LGDWTCriterion.sums_for): words 0..12 are zero on entry -
        gs_lgdwt_combine_p re-zeroes them after reading (GsLgdwtParams.reset_sums) - and word 13 holds the number of
        selected patches; so neither a fill nor a copy kernel runs per step."""
        api = ops.api
        raw, gt = _c(raw), _c(gt)
        Cc, H, W = raw.shape
        st = _stream(raw)
        d1, d2, d3 = torch.empty_like(raw), torch.empty_like(raw), torch.empty_like(raw)
        # the patch term rides on the global DWT kernels when the sizes allow (its sums are level-1 band differences of the
        # same 2 x 2 blocks, restricted to the selected patches): two launches less per step
        patch_folded = bool(params.patch_enable and params.dwt_enable and H % 4 == 0 and W % 4 == 0
                            and ctx_ps(params) % 4 == 0)
        # Order-independent sums: the kernels store their workgroups' sums (rows of 12 / single floats) and the combine kernel
        # adds them in index order - with the float atomics of the plain forms the loss, the running-mean DWT scale and with it
        # every gradient depended in the last bits on which workgroup finished first
        dwt_part = l1_part = None
        fast = H % 4 == 0 and W % 4 == 0
        if patch_folded or (params.dwt_enable and fast and params.clamp):
            img = torch.empty_like(raw)
            dwt_part = torch.empty((params.n_dwt_partials * 12,), dtype=torch.float32, device=raw.device)
            api.call("l1_dwt2_patch_fwd_clamp_p", raw.data_ptr(), gt.data_ptr(), Cc, H, W, ctx_ps(params) if patch_folded else 0,
                     mask.data_ptr() if patch_folded else None, dwt_part.data_ptr(), img.data_ptr(), st)
        elif params.dwt_enable:
            img = raw.clamp(0, 1) if params.clamp else raw
            api.call("l1_dwt2_fwd", img.data_ptr(), gt.data_ptr(), Cc, H, W, sums.data_ptr(), sums[2:].data_ptr(), st)
        else:
            img = raw.clamp(0, 1) if params.clamp else raw
            if img.numel() % 4 == 0:
                l1_part = torch.empty((params.n_l1_partials,), dtype=torch.float32, device=raw.device)
                api.call("l1_fwd_p", img.data_ptr(), gt.data_ptr(), img.numel(), l1_part.data_ptr(), st)
            else:
                api.call("l1_fwd", img.data_ptr(), gt.data_ptr(), img.numel(), sums.data_ptr(), st)
        # SSIM sum as per-workgroup partials (no atomics; lgdwt_combine_p adds them up in a fixed order)
        partials = torch.empty((params.n_ssim_partials,), dtype=torch.float32, device=raw.device)
        api.call("ssim_fwd_partials", img.data_ptr(), gt.data_ptr(), 1, Cc, H, W, 0.01 ** 2, 0.03 ** 2,
                 partials.data_ptr(), d1.data_ptr(), d2.data_ptr(), d3.data_ptr(), st)
        if params.patch_enable and not patch_folded:
            api.call("patch_dwt_fwd", img.data_ptr(), gt.data_ptr(), Cc, H, W, ctx_ps(params), mask.data_ptr(),
                     sums[10:].data_ptr(), st)
        out = torch.empty((24,), dtype=torch.float32, device=raw.device)
        api.call("lgdwt_combine_pp", sums.data_ptr(), partials.data_ptr(), partials.numel(),
                 None if dwt_part is None else dwt_part.data_ptr(), 0 if dwt_part is None else dwt_part.numel() // 12,
                 None if l1_part is None else l1_part.data_ptr(), 0 if l1_part is None else l1_part.numel(),
                 running_mean.data_ptr(), C.byref(params.c), out.data_ptr(), st)
        ctx.ops, ctx.params, ctx.patch_folded = ops, params, patch_folded
        ctx.save_for_backward(raw, img, gt, mask, d1, d2, d3, out)
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)   # no zero tensor for the unused gradient of `out`
        return out[0], out

    @staticmethod
    def backward(ctx, g, _gout):
        raw, img, gt, mask, d1, d2, d3, out = ctx.saved_tensors
        if g is None:
            return (None,) * 7
        api, params = ctx.ops.api, ctx.params
        Cc, H, W = raw.shape
        st = _stream(raw)
        # [c_l1, c_ssim, c_band x8, c_patch x3, ...] x upstream gradient; a caller that seeds the backward with
        # LossOps.unit_grad (the train step does) gets the coefficients as they are: no multiply kernel
        unit = ctx.ops._unit.get(g.device)
        coef = out[8:24] if (unit is not None and g.data_ptr() == unit.data_ptr()) else (out[8:24] * g).contiguous()
        grad = torch.empty_like(raw)
        if ctx.patch_folded:
            api.call("l1_dwt2_patch_bwd", img.data_ptr(), gt.data_ptr(), Cc, H, W, ctx_ps(params), mask.data_ptr(),
                     coef.data_ptr(), coef[2:].data_ptr(), coef[10:].data_ptr(), grad.data_ptr(), 0, st)
        elif params.dwt_enable:
            api.call("l1_dwt2_bwd", img.data_ptr(), gt.data_ptr(), Cc, H, W, coef.data_ptr(), coef[2:].data_ptr(),
                     grad.data_ptr(), 0, st)
        else:
            api.call("l1_bwd_dev", img.data_ptr(), gt.data_ptr(), img.numel(), coef.data_ptr(), grad.data_ptr(), 0, st)
        if params.patch_enable and not ctx.patch_folded:
            api.call("patch_dwt_bwd", img.data_ptr(), gt.data_ptr(), Cc, H, W, ctx_ps(params), mask.data_ptr(),
                     coef[10:].data_ptr(), grad.data_ptr(), 1, st)
        hook = getattr(ctx.ops, "before_last_backward_kernel", None)
        if hook is not None:   # (the train step's side launch, RasterBackend.UNINST_AT = "ssim_backward")
            hook()
        api.call("ssim_bwd_uniform", img.data_ptr(), gt.data_ptr(), 1, Cc, H, W, coef[1:].data_ptr(), d1.data_ptr(),
                 d2.data_ptr(), d3.data_ptr(), grad.data_ptr(), 1, raw.data_ptr() if params.clamp else None, st)
        return None, grad, None, None, None, None, None


def ctx_ps(params):
    return int(params.patch_size)


class _FusedParams:
    """Host-side constants of one image size (the C struct + the patch size)."""

    def __init__(self, crit, Cc, H, W):
        from .capi import GsLgdwtParams
        h1, w1 = (H + 1) // 2, (W + 1) // 2
        h2, w2 = (h1 + 1) // 2, (w1 + 1) // 2
        c = GsLgdwtParams()
        c.lambda_dssim = crit.lambda_dssim
        c.n_pix = float(Cc * H * W)
        c.n_band1, c.n_band2 = float(Cc * h1 * w1), float(Cc * h2 * w2)
        for k in range(8):
            c.dwt_w[k] = crit.dwt_weights[k]
        c.patch_w[0], c.patch_w[1] = crit.patch_lh1_weight, crit.patch_hl1_weight
        c.patch_w[2] = 0.5 * (crit.patch_lh1_weight + crit.patch_hl1_weight)
        c.patch_weight = crit.patch_dwt_weight
        hp = (crit.patch_size + 1) // 2
        c.patch_elems_per_sel = float(Cc * hp * hp)
        self.patch_enable = bool(crit.patch_dwt_enable and H >= crit.patch_size and W >= crit.patch_size)
        self.dwt_enable = bool(crit.dwt_enable)
        c.dwt_enable, c.patch_enable = int(self.dwt_enable), int(self.patch_enable)
        c.reset_sums = 1
        cb = getattr(crit, "custom_base", None)   # (w_l1, w_ssim): the NIR term of the multispectral criterion
        if cb is not None:
            c.custom_base, c.w_l1, c.w_ssim = 1, float(cb[0]), float(cb[1])
        self.clamp = bool(getattr(crit, "clamp", True))   # False: the image is not a clamped render (NIR channel)
        self.c = c
        self.patch_size = crit.patch_size
        self.n_ssim_partials = int(crit.ops.api.raw("ssim_partials_count")(1, Cc, H, W))
        self.n_dwt_partials = int(crit.ops.api.raw("dwt_partials_count")(Cc, H, W))
        self.n_l1_partials = int(crit.ops.api.raw("l1_partials_count")(Cc * H * W))


class LGDWTCriterion:
    """Loss composition of LGDWT-GS/train.py:128-202 with the reference's defaults
    (LGDWT-GS/arguments/__init__.py:88-122).  The running-mean DWT scale lives on the DEVICE (the
    reference pulls base/dwt to the host with .item() every iteration); per-camera ELF masks (a
    function of the ground truth only) can be cached by the caller through `mask`."""

    def __init__(self, ops, lambda_dssim=0.2, dwt_enable=True, patch_dwt_enable=True,
                 dwt_weights=(1.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0), patch_size=128, patch_percentile=0.2,
                 patch_dwt_weight=0.1, patch_lh1_weight=1.0, patch_hl1_weight=1.0, fused=True):
        self.ops = ops
        self.fused = fused        # one autograd node on the UN-clamped render (see FusedLGDWTLoss)
        self._fp = {}
        self._sums_nomask, self._no_mask = {}, {}
        self.lambda_dssim = lambda_dssim
        self.dwt_enable, self.patch_dwt_enable = dwt_enable, patch_dwt_enable
        self.dwt_weights = tuple(dwt_weights)
        self.patch_size, self.patch_percentile = patch_size, patch_percentile
        self.patch_dwt_weight = patch_dwt_weight
        self.patch_lh1_weight, self.patch_hl1_weight = patch_lh1_weight, patch_hl1_weight
        self.dwt_running_mean = None  # device scalar, starts at 1.0 (train.py:75)

    def elf_mask(self, gt_image):
        elf = self.ops.compute_elf_map(gt_image.unsqueeze(0))
        mask, _ = self.ops.patch_mask(elf, self.patch_size, self.patch_percentile)
        mask._gs_n_sel = mask.sum().to(torch.float32).reshape(1)   # travels with the (per-camera, cached) mask
        return mask

    def sums_for(self, mask, device):
        """The persistent accumulator of FusedLGDWTLoss for this camera: kept with its (cached) patch mask, word 13 = number
        of selected patches; one shared buffer (word 13 = 0) when the patch term is off."""
        if mask is None:
            s = self._sums_nomask.get(device)
            if s is None:
                s = self._sums_nomask[device] = torch.zeros((16,), dtype=torch.float32, device=device)
            return s
        s = getattr(mask, "_gs_sums", None)
        if s is None:
            n_sel = getattr(mask, "_gs_n_sel", None)
            if n_sel is None:
                n_sel = mask.sum().to(torch.float32).reshape(1)
            s = torch.zeros((16,), dtype=torch.float32, device=device)
            s[13:14].copy_(n_sel)
            try:
                mask._gs_sums = s       # cached with the per-camera mask
            except Exception:
                pass
        return s

    def fused_call(self, raw_image, gt_image, mask=None, manual_ctx=None):
        """Criterion on the rasterizer's raw output (the clamp of gaussian_renderer/__init__.py:119 is applied -
        and differentiated - inside).  Returns (loss, parts) like __call__.
        manual_ctx: an object that stands in for the autograd context - the node's forward runs under no_grad and the caller
        calls FusedLGDWTLoss.backward(manual_ctx, seed, None) itself (gsplat_amd.trainer: the train step without the autograd
        engine)."""
        key = tuple(raw_image.shape)
        fp = self._fp.get(key)
        if fp is None:
            fp = self._fp[key] = _FusedParams(self, *key)
        dev = raw_image.device
        if self.dwt_running_mean is None:
            self.dwt_running_mean = torch.ones((1,), dtype=torch.float32, device=dev)
        if fp.patch_enable:
            if mask is None:
                mask = self.elf_mask(gt_image)
            sums = self.sums_for(mask, dev)
        else:
            mask = self._no_mask.get(dev)
            if mask is None:
                mask = self._no_mask[dev] = torch.zeros((1,), dtype=torch.uint8, device=dev)
            sums = self.sums_for(None, dev)
        if manual_ctx is not None:
            with torch.no_grad():
                loss, out = FusedLGDWTLoss.forward(manual_ctx, self.ops, raw_image, gt_image, mask, sums, self.dwt_running_mean, fp)
        else:
            loss, out = FusedLGDWTLoss.apply(self.ops, raw_image, gt_image, mask, sums, self.dwt_running_mean, fp)
        parts = {"l1": out[5], "ssim": out[6], "dwt": out[2], "dwt_scale": out[4], "patch": out[3], "base": out[1],
                 "running_mean_before": out[7:8]}
        return loss, parts

    def __call__(self, image, gt_image, mask=None):
        ops = self.ops
        Ll1 = ops.l1_loss(image, gt_image)
        ssim_value = ops.fused_ssim(image.unsqueeze(0), gt_image.unsqueeze(0))
        base_loss = (1.0 - self.lambda_dssim) * Ll1 + self.lambda_dssim * (1.0 - ssim_value)
        loss = base_loss
        parts = {"l1": Ll1.detach(), "ssim": ssim_value.detach()}
        if self.dwt_enable:
            dwt_loss, band_means = ops.dwt_l1_loss(image, gt_image, self.dwt_weights)
            ratio = base_loss.detach() / (dwt_loss.detach() + 1e-8)
            if self.dwt_running_mean is None:
                self.dwt_running_mean = torch.ones((1,), dtype=torch.float32, device=image.device)
            self.dwt_running_mean = 0.95 * self.dwt_running_mean + 0.05 * ratio
            dwt_scale = torch.clamp(self.dwt_running_mean, 0.1, 10.0).reshape(())
            loss = base_loss + dwt_scale * dwt_loss
            parts.update(dwt=dwt_loss.detach(), dwt_scale=dwt_scale, bands=band_means)
        if self.patch_dwt_enable and image.shape[-2] >= self.patch_size and image.shape[-1] >= self.patch_size:
            if mask is None:
                mask = self.elf_mask(gt_image)
            patch_loss = ops.compute_patch_dwt_loss(image.unsqueeze(0), gt_image.unsqueeze(0), None, self.patch_size,
                                                    self.patch_percentile, self.patch_lh1_weight,
                                                    self.patch_hl1_weight, mask=mask)
            loss = loss + self.patch_dwt_weight * patch_loss
            parts["patch"] = patch_loss.detach()
        return loss, parts
