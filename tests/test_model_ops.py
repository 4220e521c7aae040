"""gs_activations_fwd / gs_activations_bwd / gs_densify_stats: the oracle against torch itself
(exp, F.normalize, sigmoid and their autograd; the reference's boolean-indexed statistics), the HIP kernels against
the oracle, and the trainer's fused path against its torch path."""
import pytest
import torch

from gsplat_amd.trainer import _stream_of


def raw(P, seed):
    g = torch.Generator().manual_seed(seed)
    scaling = torch.randn((P, 3), generator=g) * 1.5 - 3.0
    rotation = torch.randn((P, 4), generator=g)
    rotation[::17] *= 1e-3
    opacity = torch.randn((P, 1), generator=g) * 3.0
    grads = (torch.randn((P, 3), generator=g), torch.randn((P, 4), generator=g), torch.randn((P, 1), generator=g))
    return scaling, rotation, opacity, grads


def run_api(api, scaling, rotation, opacity, grads):
    dev = scaling.device
    P = scaling.shape[0]
    outs = [torch.empty_like(t) for t in (scaling, rotation, opacity)]
    api.call("activations_fwd", scaling.data_ptr(), rotation.data_ptr(), opacity.data_ptr(), P, *[o.data_ptr() for o in outs],
             _stream_of(scaling))
    d = [torch.empty_like(t) for t in (scaling, rotation, opacity)]
    g = [x.to(dev).contiguous() for x in grads]
    api.call("activations_bwd", scaling.data_ptr(), rotation.data_ptr(), opacity.data_ptr(), P, *[x.data_ptr() for x in g],
             *[x.data_ptr() for x in d], _stream_of(scaling))
    return outs, d


def test_oracle_activations_match_torch_autograd(oracle):
    scaling, rotation, opacity, grads = raw(3000, 0)
    outs, d = run_api(oracle.api, scaling, rotation, opacity, grads)
    s, r, o = (t.clone().requires_grad_(True) for t in (scaling, rotation, opacity))
    want = (torch.exp(s), torch.nn.functional.normalize(r), torch.sigmoid(o))
    torch.autograd.backward(want, grads)
    for a, b in zip(outs, want):
        assert torch.allclose(a, b.detach(), rtol=2e-7, atol=1e-30)
    for a, b in zip(d, (s.grad, r.grad, o.grad)):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max()))


def stats_reference(radii, grad, max_radii2D, accum, denom):
    """train.py:266-268 + gaussian_model.py:471-473 verbatim (boolean indexing)."""
    vis = radii > 0
    max_radii2D[vis] = torch.max(max_radii2D[vis], radii[vis].float())
    accum[vis] += torch.norm(grad[vis, :2], dim=-1, keepdim=True)
    denom[vis] += 1
    return max_radii2D, accum, denom


def stats_inputs(P, seed):
    g = torch.Generator().manual_seed(seed)
    radii = torch.randint(-1, 40, (P,), generator=g, dtype=torch.int32).clamp_min(0)
    grad = torch.randn((P, 3), generator=g) * 1e-3
    return radii, grad, torch.rand((P,), generator=g) * 30, torch.rand((P, 1), generator=g), torch.randint(0, 5, (P, 1), generator=g).float()


def test_oracle_densify_stats_match_reference_indexing(oracle):
    radii, grad, mr, acc, den = stats_inputs(5000, 1)
    want = stats_reference(radii, grad, mr.clone(), acc.clone(), den.clone())
    oracle.api.call("densify_stats", radii.data_ptr(), grad.data_ptr(), 5000, mr.data_ptr(), acc.data_ptr(), den.data_ptr(), None)
    assert torch.equal(mr, want[0]) and torch.equal(den, want[2])
    assert torch.allclose(acc, want[1], rtol=1e-6, atol=0)


def test_trainer_fused_path_equals_torch_path(oracle):
    from test_trainer_cpu import make_trainer
    from gsplat_amd.trainer import render
    a, b = make_trainer(oracle, P=300, W=96, H=64, dwt=False), make_trainer(oracle, P=300, W=96, H=64, dwt=False)
    a.optimizer_step = b.optimizer_step = False
    a.step(0)
    # torch path: the get_* activations and the reference's statistics ops
    m = b.model
    m.zero_grad()
    pkg = render(b.cameras[0], m, b.Rasterizer, b.Settings, b.bg, filter_as_indices=False, clamp=False, fused=False)
    loss, _ = b.criterion.fused_call(pkg["render"], b.gts[0], mask=None)
    loss.backward()
    m.collect_grads()
    assert float((a.model.flat_grad - m.flat_grad).abs().max()) <= 2e-6 * float(m.flat_grad.abs().max())
    ref = stats_reference(pkg["radii"], pkg["viewspace_points"].grad, torch.zeros(300), torch.zeros((300, 1)), torch.zeros((300, 1)))
    assert torch.equal(a.model.max_radii2D, ref[0]) and torch.equal(a.model.denom, ref[2])
    # (atol: a gradient that is ~1e-11 on one path and exactly 0 on the other - the oracle's OpenMP partition depends on the host's
    #  thread count; seen with 128 threads)
    got, want = a.model.xyz_gradient_accum, ref[1]
    off = ~torch.isclose(got, want, rtol=1e-5, atol=1e-12)
    assert bool(((torch.minimum(got.abs(), want.abs()) == 0) & (torch.maximum(got.abs(), want.abs()) <= 1e-9))[off].all())


@pytest.mark.gpu
def test_hip_model_ops_match_oracle(hip, oracle):
    scaling, rotation, opacity, grads = raw(100003, 2)
    o_out, o_d = run_api(oracle.api, scaling, rotation, opacity, grads)
    h_out, h_d = run_api(hip.api, scaling.cuda(), rotation.cuda(), opacity.cuda(), grads)
    for a, b in zip(h_out + h_d, o_out + o_d):
        assert float((a.cpu() - b).abs().max()) <= 1e-6 * max(1.0, float(b.abs().max()))
    radii, grad, mr, acc, den = stats_inputs(100003, 3)
    want = [t.clone() for t in (mr, acc, den)]
    oracle.api.call("densify_stats", radii.data_ptr(), grad.data_ptr(), 100003, *[t.data_ptr() for t in want], None)
    dev = [t.cuda() for t in (radii, grad, mr, acc, den)]
    hip.api.call("densify_stats", dev[0].data_ptr(), dev[1].data_ptr(), 100003, dev[2].data_ptr(), dev[3].data_ptr(),
                 dev[4].data_ptr(), _stream_of(dev[0]))
    for a, b in zip(dev[2:], want):
        assert torch.equal(a.cpu(), b)
