"""GPU parity of the loss kernels (through the C ABI) vs the CPU oracle, same LossOps glue."""
import pytest
import torch

from gsplat_amd.losses import LGDWTCriterion, LossOps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops_pair(hip, oracle):
    return LossOps(hip.api), LossOps(oracle.api)


@pytest.mark.parametrize("tag", ("a", "b", "c", "d"))
def test_hip_loss_kernels_vs_the_reference_module_fixture(hip, tag):
    """D1-D4 of the HIP kernels against what the reference's own LGDWT-GS/utils/loss_utils.py returned in the build
    container (tests/golden/lgdwt_loss.npz, tests/lgdwt_fixture.py)."""
    import lgdwt_fixture
    lgdwt_fixture.check_case(LossOps(hip.api), torch.device("cuda"), lgdwt_fixture.load(), tag)


def test_hip_criterion_running_mean_vs_the_reference_fixture(hip):
    import lgdwt_fixture
    z = lgdwt_fixture.load()
    lgdwt_fixture.check_criterion_defaults(LGDWTCriterion, LossOps(hip.api), z)
    lgdwt_fixture.check_running_mean(LGDWTCriterion, LossOps(hip.api), z, torch.device("cuda"))


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


@pytest.mark.parametrize("H,W,acc", [(64, 96, 0), (64, 96, 1), (37, 53, 1), (131, 260, 0), (1080, 1920, 0)])
def test_l1_dwt2_one_pass_equals_the_two_kernels_and_the_oracle(hip, oracle, H, W, acc):
    """gs_l1_dwt2_fwd / _bwd (one read of the images) against gs_l1_* + gs_dwt2_l1_* of the oracle, on the float4
    path (sizes divisible by 4) and on the padded generic path."""
    pred, gt = images(H, W, 3 * H + W)
    g = torch.Generator().manual_seed(5)
    coef = torch.rand((9,), generator=g) / (H * W)          # [c_l1, c_band x8]
    g0 = torch.randn((3, H, W), generator=g) * 1e-6
    out = {}
    for name, api, dev in (("hip", hip.api, "cuda"), ("oracle", oracle.api, "cpu")):
        p, t, c, gr = pred.to(dev), gt.to(dev), coef.to(dev), g0.to(dev).clone()
        sums = torch.zeros((9,), device=dev)
        st = torch.cuda.current_stream().cuda_stream if dev == "cuda" else None
        api.call("l1_dwt2_fwd", p.data_ptr(), t.data_ptr(), 3, H, W, sums.data_ptr(), sums[1:].data_ptr(), st)
        api.call("l1_dwt2_bwd", p.data_ptr(), t.data_ptr(), 3, H, W, c.data_ptr(), c[1:].data_ptr(), gr.data_ptr(), acc, st)
        out[name] = (sums.cpu(), gr.cpu())
    close(out["hip"][0], out["oracle"][0], 1e-5, "sums")
    for k in range(9):
        assert abs(float(out["hip"][0][k]) - float(out["oracle"][0][k])) <= 1e-5 * float(out["oracle"][0][k]), k
    close(out["hip"][1], out["oracle"][1], 2e-5, "grad")


@pytest.mark.parametrize("H,W,ps,acc", [(256, 384, 64, 0), (260, 392, 128, 1), (1080, 1920, 128, 0)])
def test_patch_term_folded_into_the_dwt_kernels(hip, oracle, H, W, ps, acc):
    """gs_l1_dwt2_patch_fwd_clamp / gs_l1_dwt2_patch_bwd (L1 + global 2-level DWT + patch DWT + clamp from one pass each
    way) against the separate HIP kernels and against the oracle: the clamped image bit for bit, the twelve sums and the
    gradient to rounding (the folded form adds the patch coefficients to the band coefficients BEFORE the Haar adjoint)."""
    g = torch.Generator().manual_seed(H + ps)
    gt = torch.rand((3, H, W), generator=g)
    raw = gt + 0.2 * torch.randn((3, H, W), generator=g)          # un-clamped render: values outside [0, 1]
    mask = (torch.rand(((H // ps) * (W // ps),), generator=g) < 0.4).to(torch.uint8)
    coef = torch.rand((13,), generator=g) / (H * W)                # [c_l1, c_band x8, -, c_patch x3]
    g0 = torch.randn((3, H, W), generator=g) * 1e-6
    out = {}
    for name, api, dev, folded in (("hip", hip.api, "cuda", True), ("hip_separate", hip.api, "cuda", False),
                                   ("oracle", oracle.api, "cpu", True)):
        r, t, m, c = raw.to(dev), gt.to(dev), mask.to(dev), coef.to(dev)
        gr = g0.to(dev).clone()
        sums = torch.zeros((13,), device=dev)                       # [l1, band x8, -, patch x3]
        st = torch.cuda.current_stream().cuda_stream if dev == "cuda" else None
        if folded:
            img = torch.empty_like(r)
            api.call("l1_dwt2_patch_fwd_clamp", r.data_ptr(), t.data_ptr(), 3, H, W, ps, m.data_ptr(), sums.data_ptr(),
                     sums[1:].data_ptr(), sums[10:].data_ptr(), img.data_ptr(), st)
            api.call("l1_dwt2_patch_bwd", img.data_ptr(), t.data_ptr(), 3, H, W, ps, m.data_ptr(), c.data_ptr(),
                     c[1:].data_ptr(), c[10:].data_ptr(), gr.data_ptr(), acc, st)
        else:
            img = r.clamp(0, 1)
            api.call("l1_dwt2_fwd", img.data_ptr(), t.data_ptr(), 3, H, W, sums.data_ptr(), sums[1:].data_ptr(), st)
            api.call("patch_dwt_fwd", img.data_ptr(), t.data_ptr(), 3, H, W, ps, m.data_ptr(), sums[10:].data_ptr(), st)
            api.call("l1_dwt2_bwd", img.data_ptr(), t.data_ptr(), 3, H, W, c.data_ptr(), c[1:].data_ptr(), gr.data_ptr(), acc, st)
            api.call("patch_dwt_bwd", img.data_ptr(), t.data_ptr(), 3, H, W, ps, m.data_ptr(), c[10:].data_ptr(), gr.data_ptr(), 1, st)
        out[name] = (img.cpu(), sums.cpu(), gr.cpu())
    assert torch.equal(out["hip"][0], out["hip_separate"][0]) and torch.equal(out["hip"][0], out["oracle"][0])
    assert float(out["hip"][1][10:].min()) > 0, "no selected patch contributed"
    for other in ("hip_separate", "oracle"):
        for k in list(range(9)) + [10, 11, 12]:
            a, b = float(out["hip"][1][k]), float(out[other][1][k])
            assert abs(a - b) <= 1e-5 * abs(b), (other, k, a, b)
        close(out["hip"][2], out[other][2], 2e-5, "grad vs " + other)


@pytest.mark.parametrize("H,W", [(1080, 1920), (131, 260), (33, 40)])
def test_ssim_partials_equal_the_atomic_sum_and_the_oracle(hip, oracle, H, W):
    """gs_ssim_fwd_partials (one plain store per workgroup) + gs_lgdwt_combine_p against gs_ssim_fwd_sum +
    gs_lgdwt_combine, on the device and in the oracle."""
    import ctypes as C
    from gsplat_amd.losses import _FusedParams
    pred, gt = images(H, W, 7 * H + W)
    res = {}
    for name, api, dev in (("hip", hip.api, "cuda"), ("oracle", oracle.api, "cpu")):
        p, t = pred.to(dev), gt.to(dev)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream) if dev == "cuda" else None
        n = int(api.raw("ssim_partials_count")(1, 3, H, W))
        assert n == ((W + 31) // 32) * ((H + 31) // 32) * 3
        part = torch.full((n,), float("nan"), device=dev)
        d = [torch.empty_like(p) for _ in range(6)]
        s_atomic = torch.zeros((16,), device=dev)
        api.call("ssim_fwd_partials", p.data_ptr(), t.data_ptr(), 1, 3, H, W, 1e-4, 9e-4, part.data_ptr(), d[0].data_ptr(),
                 d[1].data_ptr(), d[2].data_ptr(), st)
        api.call("ssim_fwd_sum", p.data_ptr(), t.data_ptr(), 1, 3, H, W, 1e-4, 9e-4, s_atomic[1:].data_ptr(), d[3].data_ptr(),
                 d[4].data_ptr(), d[5].data_ptr(), st)
        assert bool(torch.isfinite(part).all())
        for k in range(3):
            assert torch.equal(d[k], d[k + 3])
        fp = _FusedParams(LGDWTCriterion(LossOps(api)), 3, H, W)
        s_atomic[0] = 0.1 * 3 * H * W
        s_part = s_atomic.clone()
        s_part[1] = 0.0
        rm1, rm2 = torch.ones(1, device=dev), torch.ones(1, device=dev)
        o1, o2 = torch.empty(24, device=dev), torch.empty(24, device=dev)
        atomic_sum = float(s_atomic[1])
        assert fp.c.reset_sums == 1   # the fused criterion's form: the call re-zeroes the accumulators it has read
        api.call("lgdwt_combine", s_atomic.data_ptr(), rm1.data_ptr(), C.byref(fp.c), o1.data_ptr(), st)
        api.call("lgdwt_combine_p", s_part.data_ptr(), part.data_ptr(), n, rm2.data_ptr(), C.byref(fp.c), o2.data_ptr(), st)
        assert float(s_atomic[:13].abs().max()) == 0.0 and float(s_part[:13].abs().max()) == 0.0
        assert float(o1[7]) == 1.0 and float(o2[7]) == 1.0   # the running mean the call started from
        res[name] = (float(part.double().sum()), atomic_sum, o1.cpu(), o2.cpu())
    for name in res:
        ps, at, o1, o2 = res[name]
        assert abs(ps - at) <= 1e-5 * abs(at), (name, ps, at)
        close(o2, o1, 1e-5, name + " combine_p vs combine")
    assert abs(res["hip"][0] - res["oracle"][0]) <= 1e-5 * abs(res["oracle"][0])
    close(res["hip"][3], res["oracle"][3], 1e-5, "combine_p hip vs oracle")
