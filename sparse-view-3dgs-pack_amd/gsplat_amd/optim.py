"""`FusedAdam`: torch.optim.Adam's update (no weight decay, no amsgrad - what LGDWT-GS/scene/gaussian_model.py:192 builds:
`torch.optim.Adam(l, lr=0.0, eps=1e-15)`) as ONE streaming kernel per parameter tensor (`gs_adam_step`, csrc/gs_adam.hip:
parameter, gradient and both moments read once, parameter and moments written once) instead of torch's multi-pass
foreach implementation.  A drop-in: same constructor arguments, same param_groups / state_dict layout
(`state[p] = {"step", "exp_avg", "exp_avg_sq"}`), so the reference's optimizer surgery (cat_tensors_to_optimizer,
_prune_optimizer, replace_tensor_to_optimizer: gaussian_model.py:316-393) works on it unchanged.  Arithmetic = torch's
single-tensor Adam (tests/test_adam.py pins gs_adam_step against torch.optim.Adam)."""
import ctypes as C

import torch


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        from ._lib import hip_api
        from .capi import GsAdamSeg
        self._api, self._Seg = hip_api(), GsAdamSeg

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue   # (torch skips such a parameter: moments and step count untouched)
                if not (p.is_cuda and p.is_contiguous() and p.dtype == torch.float32):
                    raise RuntimeError("FusedAdam steps contiguous fp32 tensors on the GPU")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                seg = (self._Seg * 1)()
                seg[0].begin, seg[0].end, seg[0].lr_a, seg[0].step = 0, p.numel(), float(group["lr"]), int(st["step"])
                self._api.call("adam_step", p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                               p.numel(), seg, 1, float(b1), float(b2), float(group["eps"]), int(st["step"]),
                               C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream))
        return loss
