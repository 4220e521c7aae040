"""GPU parity of the loss kernels (through the C ABI) vs the CPU oracle, same LossOps glue."""
import pytest
import torch

from gsplat_amd.losses import LGDWTCriterion, LossOps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops_pair(hip, oracle):
    return LossOps(hip.api), LossOps(oracle.api)


def images(H, W, seed, C=3):
    g = torch.Generator().manual_seed(seed)
    gt = torch.rand((C, H, W), generator=g)
    gt[:, : H // 2] = gt[:, : H // 2] * 0.1 + 0.4
    pred = (gt + 0.1 * torch.randn((C, H, W), generator=g)).clamp(0, 1)
    return pred, gt


def close(a, b, tol, what):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = max(1e-12, float(b.abs().max()))
    e = float((a - b).abs().max()) / scale
    assert e <= tol, "%s: rel err %.3e" % (what, e)


@pytest.mark.parametrize("H,W", [(400, 400), (1080, 1920), (37, 53), (131, 260)])
def test_every_loss_term_value_and_gradient(ops_pair, H, W):
    hop, oop = ops_pair
    pred, gt = images(H, W, H + W)
    res = {}
    for name, ops, dev in (("hip", hop, "cuda"), ("oracle", oop, "cpu")):
        p = pred.to(dev).requires_grad_(True)
        g = gt.to(dev)
        out = {}
        l1 = ops.l1_loss(p, g)
        out["l1"] = l1
        (gl1,) = torch.autograd.grad(l1, p)
        out["g_l1"] = gl1
        s = ops.fused_ssim(p[None], g[None])
        out["ssim"] = s
        (out["g_ssim"],) = torch.autograd.grad(s, p)
        d, means = ops.dwt_l1_loss(p, g, (1.0, 1.0, 1.0, 0.5, 0.25, 0.7, 0.0, 1.5))
        out["dwt"], out["bands"] = d, means
        (out["g_dwt"],) = torch.autograd.grad(d, p)
        b = ops.get_dwt_subbands(p[None])
        for k in ("LL1", "HH1", "LH2", "HL2"):
            out["band_" + k] = b[k]
        elf = ops.compute_elf_map(g[None])
        out["elf"] = elf
        if H >= 128 and W >= 128:
            mask, pm = ops.patch_mask(elf, 128, 0.2)
            out["mask"], out["pmeans"] = mask.float(), pm
            pl = ops.compute_patch_dwt_loss(p[None], g[None], elf, 128, 0.2, 1.0, 0.5)
            out["patch"] = pl
            (out["g_patch"],) = torch.autograd.grad(pl, p)
        res[name] = out
    for k in res["oracle"]:
        if k == "mask":
            assert torch.equal(res["hip"][k].cpu(), res["oracle"][k]), "patch mask differs"
            continue
        # sign-type gradients (L1/DWT) are exact up to sign(0) coincidences; conv-type within fp32 rounding
        close(res["hip"][k], res["oracle"][k], 2e-5 if k.startswith("g_") or k in ("ssim",) else 1e-5, k)


def test_criterion_step_matches_oracle(ops_pair):
    hop, oop = ops_pair
    pred, gt = images(256, 384, 11)
    ch, co = LGDWTCriterion(hop), LGDWTCriterion(oop)
    for it in range(3):
        ph = pred.cuda().requires_grad_(True)
        po = pred.clone().requires_grad_(True)
        lh, _ = ch(ph, gt.cuda())
        lo, _ = co(po, gt)
        lh.backward()
        lo.backward()
        close(lh, lo, 1e-5, "loss it%d" % it)
        close(ph.grad, po.grad, 5e-5, "dL/dimage it%d" % it)


def test_reference_named_packages_run_on_gpu():
    """`fused_ssim`, `pytorch_wavelets.DWTForward` and `lgdwt_loss` drop-ins import and run."""
    import fused_ssim
    import lgdwt_loss
    from pytorch_wavelets import DWTForward
    x = torch.rand((1, 3, 66, 50), device="cuda", requires_grad=True)
    y = torch.rand((1, 3, 66, 50), device="cuda")
    v = fused_ssim.fused_ssim(x, y)
    v.backward()
    assert x.grad.shape == x.shape
    Yl, Yh = DWTForward(J=2, mode="symmetric", wave="db1").to("cuda")(x)
    assert Yl.shape == (1, 3, 17, 13) and Yh[0].shape == (1, 3, 3, 33, 25) and Yh[1].shape == (1, 3, 3, 17, 13)
    b = lgdwt_loss.get_dwt_subbands(x)
    assert torch.equal(b["LL2"], Yl) and torch.equal(b["HH1"], Yh[0][:, :, 2])


@pytest.mark.parametrize("H,W", [(1080, 1920), (131, 260)])
def test_fused_criterion_hip_vs_oracle(ops_pair, H, W):
    hop, oop = ops_pair
    g = torch.Generator().manual_seed(H)
    gt = torch.rand((3, H, W), generator=g)
    raw = gt + 0.2 * torch.randn((3, H, W), generator=g)
    ch, co = LGDWTCriterion(hop), LGDWTCriterion(oop)
    for it in range(2):
        rh = raw.cuda().requires_grad_(True)
        ro = raw.clone().requires_grad_(True)
        lh, ph = ch.fused_call(rh, gt.cuda())
        lo, po = co.fused_call(ro, gt)
        lh.backward()
        lo.backward()
        close(lh, lo, 1e-5, "fused loss")
        close(ph["dwt_scale"], po["dwt_scale"], 1e-5, "dwt scale")
        close(rh.grad, ro.grad, 5e-5, "fused dL/draw")
