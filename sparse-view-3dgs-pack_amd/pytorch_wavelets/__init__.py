"""Minimal stand-in for the third-party `pytorch_wavelets` import of the reference
(LGDWT-GS/utils/loss_utils.py:104,121,140: `DWTForward(J, mode='symmetric', wave='db1')`), so that the
reference's loss_utils.py imports and runs unmodified on the MI355X Haar kernels.
Only what the reference uses is provided: wave 'db1'/'haar', mode 'symmetric', any J >= 1."""
import torch.nn as nn

import lgdwt_loss


class DWTForward(nn.Module):
    def __init__(self, J=1, wave="db1", mode="zero"):
        super().__init__()
        if wave not in ("db1", "haar"):
            raise NotImplementedError("only the Haar ('db1') wavelet is implemented")
        if mode != "symmetric":
            raise NotImplementedError("only mode='symmetric' is implemented")
        self.J = J

    def forward(self, x):
        """x [N,C,H,W] -> (Yl, [Yh_1 .. Yh_J]) with Yh_j [N,C,3,h_j,w_j] ordered (LH, HL, HH)."""
        import torch
        yh = []
        ll = x
        for _ in range(self.J):
            ll, lh, hl, hh = lgdwt_loss.ops().dwt_haar(ll)
            yh.append(torch.stack([lh, hl, hh], dim=2))
        return ll, yh
