"""Depth regularisation of the step (LGDWT-GS/train.py:69, 204-216): weight schedule, the term
`torch.abs((invDepth - mono_invdepth) * depth_mask).mean()` and its gradient into the rasterizer's inverse-depth output.

The fixture tests/golden/depth_reg.npz was computed by the REFERENCE's own statements (tests/golden/make_golden.py::gen_depth_reg
reads them from train.py when it runs and executes them on CPU stand-ins).  CPU: the oracle's gso_depth_l1 and the train
loop's composition against it; GPU: gs_depth_l1 against it, and the fused train step with the term against the autograd form."""
import os

import numpy as np
import pytest
import torch

from gsplat_amd.losses import LossOps
from gsplat_amd.trainer import TrainOptions, expon_lr

FIX = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "depth_reg.npz"))
CASES = ("a", "b", "c")


def test_weight_schedule_is_the_references():
    o = TrainOptions()
    assert o.depth_l1_weight_init == float(FIX["weight_init"]) and o.depth_l1_weight_final == float(FIX["weight_final"])
    assert o.iterations == int(FIX["iterations"])
    for it, w in zip(FIX["weight_at"], FIX["weight"]):
        got = expon_lr(int(it), o.depth_l1_weight_init, o.depth_l1_weight_final, max_steps=o.iterations)
        assert abs(got - float(w)) <= 1e-12 * max(1.0, abs(float(w))), (it, got, w)


def check_term(ops, device):
    for tag in CASES:
        inv = torch.from_numpy(FIX["invdepth_" + tag]).to(device).requires_grad_(True)
        mono = torch.from_numpy(FIX["mono_" + tag]).to(device)
        mask = torch.from_numpy(FIX["mask_" + tag]).to(device)
        it, reliable = int(FIX["iteration_" + tag]), bool(FIX["reliable_" + tag])
        w = expon_lr(it, 1.0, 0.01, max_steps=30000)
        pure = ops.depth_l1(inv, mono, mask)
        (w * pure).backward()
        want_pure = float(FIX["pure_" + tag]) if reliable else None
        if reliable:   # (an unreliable camera never reaches the term: train.py:206 - the fixture then holds zeros)
            assert abs(float(pure) - want_pure) <= 2e-6 * want_pure, (tag, float(pure), want_pure)
            assert abs(w * float(pure) - float(FIX["loss_" + tag])) <= 2e-6 * float(FIX["loss_" + tag])
            g, want = inv.grad.cpu().numpy(), FIX["grad_" + tag]
            assert np.abs(g - want).max() <= 1e-6 * np.abs(want).max(), (tag, np.abs(g - want).max())
            assert (g[:, 3:9, 5:20] == 0).all()      # exact ties: sign(0) = 0
            # the one-launch form of the train step: the same numbers
            dl, grad = ops.depth_l1_step(inv.detach(), mono, mask, w)
            assert abs(float(dl) - float(FIX["loss_" + tag])) <= 2e-6 * float(FIX["loss_" + tag])
            assert np.abs(grad.cpu().numpy() - want).max() <= 1e-6 * np.abs(want).max()
        # no mask = a mask of ones
        a = ops.depth_l1(inv.detach(), mono, None)
        b = ops.depth_l1(inv.detach(), mono, torch.ones_like(mono))
        assert float(a) == float(b)


def test_oracle_depth_term_vs_the_reference_fixture(oracle):
    check_term(LossOps(oracle.api), torch.device("cpu"))


def depth_priors(tr, seed=3):
    g = torch.Generator().manual_seed(seed)
    dev = tr.gts[0].device
    H, W = tr.gts[0].shape[-2:]
    pri = []
    for k in range(len(tr.cameras)):
        mono = (torch.rand((1, H, W), generator=g) * 0.5).to(dev)
        mask = (torch.rand((1, H, W), generator=g) > 0.2).float().to(dev)
        pri.append(None if k == 2 else (mono, mask if k != 1 else None))   # camera 2: not depth_reliable; camera 1: no mask
    return pri


def test_train_loop_composes_the_term_like_the_reference(oracle):
    from test_trainer_cpu import make_trainer
    from gsplat_amd.trainer import render
    a, b = make_trainer(oracle, P=300, W=96, H=64), make_trainer(oracle, P=300, W=96, H=64)
    a.optimizer_step = b.optimizer_step = False
    b.depth_priors = depth_priors(b)
    for it, ci in ((1, 0), (400, 1), (700, 2), (1000, 3)):
        w = expon_lr(it, 1.0, 0.01, max_steps=1000)
        b.depth_l1_weight = w
        la, lb = a._step_camera(ci, False, ()), b._step_camera(ci, False, ())
        oa, ob = dict(loss=la), dict(loss=lb)
        prior = b.depth_priors[ci]
        if prior is None:
            assert float(ob["loss"]) == float(oa["loss"]) and torch.equal(a.model.flat_grad, b.model.flat_grad)
            continue
        with torch.no_grad():
            depth = render(b.cameras[ci], b.model, b.Rasterizer, b.Settings, b.bg)["depth"]
        k = prior[1] if prior[1] is not None else torch.ones_like(prior[0])
        want = w * float(torch.abs((depth - prior[0]) * k).mean())      # train.py:211-213
        assert abs(float(ob["loss"]) - float(oa["loss"]) - want) <= 1e-5 * max(want, 1e-3), (it, float(ob["loss"]), float(oa["loss"]), want)
        assert abs(float(b.last["parts"]["depth_l1"]) - want) <= 1e-5 * want
        assert not torch.equal(a.model.flat_grad, b.model.flat_grad)   # the inverse-depth gradient reached the parameters
    # the schedule loop sets the weight of train.py:69 itself - only for a trainer that has priors
    opt = TrainOptions(iterations=1000, densify_from_iter=10 ** 9)
    a.train_iteration(5, opt)
    b.train_iteration(5, opt)
    assert b.depth_l1_weight == expon_lr(5, 1.0, 0.01, max_steps=1000) and a.depth_l1_weight == 0.0


@pytest.mark.gpu
def test_hip_depth_term_vs_the_reference_fixture(hip):
    import lgdwt_loss
    check_term(lgdwt_loss.ops(), torch.device("cuda"))
    # a 1080p plane: partial sums in a fixed order - two runs, the same bits; against float64
    g = torch.Generator().manual_seed(1)
    d, m = torch.rand((1, 1080, 1920), generator=g).cuda(), torch.rand((1, 1080, 1920), generator=g).cuda()
    k = (torch.rand((1, 1080, 1920), generator=g) > 0.3).float().cuda()
    ops = lgdwt_loss.ops()
    x, y = ops.depth_l1(d, m, k), ops.depth_l1(d, m, k)
    assert float(x) == float(y)
    want = float(((d.double() - m.double()) * k.double()).abs().mean())
    assert abs(float(x) - want) <= 1e-6 * want


@pytest.mark.gpu
def test_fused_step_with_the_depth_term_equals_the_autograd_step(hip):
    """The hand-driven fused step (one gs_depth_l1 launch, its gradient handed to gs_backward_step as dL_dinvdepth) against
    the same step through autograd (ops.depth_l1 node + rasterizer node), and against the step without the term."""
    from test_gpu_fused_step import make, state
    a, b, c = make(hip, True, P=20000), make(hip, True, P=20000), make(hip, True, P=20000)
    for t in (a, b):
        t.depth_priors = depth_priors(t)
        t.depth_l1_weight = 0.37
    b.MANUAL_BACKWARD = False
    la, lb, lc = [], [], []
    for k in range(4):
        la.append(float(a.step(k)))
        lb.append(float(b.step(k)))
        lc.append(float(c.step(k)))
        for t in (a, b):   # camera 2 has no reliable depth; the others pay the term
            assert ("depth_l1" in t.last["parts"]) == (k != 2), k
    torch.cuda.synchronize()
    assert all(abs(x - y) <= 2e-6 * abs(y) for x, y in zip(la, lb)), (la, lb)
    assert la[0] > lc[0]
    sa, sb, sc = state(a), state(b), state(c)
    for k in sa:   # (same kernels, same inputs up to the last bit of weight / n: the models stay together)
        d = (sa[k] - sb[k]).double()
        assert float(d.pow(2).mean().sqrt()) <= 1e-5 * max(1e-12, float(sb[k].double().pow(2).mean().sqrt())), k
    assert not torch.equal(sa["flat"], sc["flat"])
