"""Behaviour of the reference-named Python API (argument rules, error text, empty inputs, return
shapes), exercised on the CPU through the oracle-bound subclass of the same classes."""
import pytest
import torch

import diff_gaussian_rasterization as dgr
from gsplat_amd import synthetic
from helpers import settings_for


def mk(oracle, W=48, H=32, bg=(0.1, 0.2, 0.3)):
    cam = synthetic.look_at_camera((3.0, 0.5, 0.5), W, H)
    rs = settings_for(oracle.Settings, cam, torch.tensor(bg), 0, torch.device("cpu"))
    return oracle.Rasterizer(rs), cam


def test_settings_is_the_reference_namedtuple():
    f = dgr.GaussianRasterizationSettings._fields
    assert f == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
                 "projmatrix", "sh_degree", "campos", "prefiltered", "debug", "antialiasing")
    assert not hasattr(dgr, "SparseGaussianAdam")          # LGDWT-GS/train.py:42-46 probes for it
    assert not hasattr(dgr._C, "fusedssim")                # LGDWT-GS/utils/loss_utils.py:16-19
    import dgr_3dgs
    assert dgr_3dgs.GaussianRasterizer is dgr.GaussianRasterizer


def test_exclusive_argument_rules_raise_like_the_reference(oracle):
    rast, _ = mk(oracle)
    m = torch.zeros((3, 3))
    o = torch.ones((3, 1))
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        rast(means3D=m, means2D=m, opacities=o, scales=m, rotations=torch.ones((3, 4)))
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        rast(means3D=m, means2D=m, opacities=o, shs=torch.zeros((3, 16, 3)), colors_precomp=m, scales=m,
             rotations=torch.ones((3, 4)))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        rast(means3D=m, means2D=m, opacities=o, colors_precomp=m, scales=m)
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        rast(means3D=m, means2D=m, opacities=o, colors_precomp=m, scales=m, rotations=torch.ones((3, 4)),
             cov3D_precomp=torch.zeros((3, 6)))
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):
        rast(means3D=torch.zeros((3, 4)), means2D=m, opacities=o, colors_precomp=m, scales=m, rotations=torch.ones((3, 4)))


def test_empty_input_returns_zero_outputs(oracle):
    rast, _ = mk(oracle)
    z = torch.zeros((0, 3))
    color, radii, invd = rast(means3D=z, means2D=z, opacities=torch.zeros((0, 1)), colors_precomp=z, scales=z,
                              rotations=torch.zeros((0, 4)))
    assert color.shape == (3, 32, 48) and invd.shape == (1, 32, 48) and radii.shape == (0,)
    assert radii.dtype == torch.int32 and float(color.abs().max()) == 0.0


def test_single_isotropic_gaussian_known_answer(oracle):
    """One Gaussian centred on a pixel: alpha = min(.99, o) there, colour = alpha c + (1-alpha) bg."""
    W, H = 33, 33
    cam = synthetic.look_at_camera((4.0, 0.0, 0.0), W, H)
    bg = torch.tensor([0.1, 0.2, 0.3])
    rs = settings_for(oracle.Settings, cam, bg, 0, torch.device("cpu"))
    rast = oracle.Rasterizer(rs)
    m = torch.zeros((1, 3))  # projects to the image centre: ndc 0 -> pixel (W-1)/2 = 16
    c = torch.tensor([[0.9, 0.5, 0.2]])
    for o in (0.6, 1.0):
        color, radii, invd = rast(means3D=m, means2D=m, opacities=torch.tensor([[o]]), colors_precomp=c,
                                  scales=torch.full((1, 3), 0.2), rotations=torch.tensor([[1.0, 0, 0, 0]]))
        a = min(0.99, o)
        expect = a * c[0] + (1 - a) * bg
        assert torch.allclose(color[:, 16, 16], expect, atol=1e-6)
        assert abs(float(invd[0, 16, 16]) - a / 4.0) < 1e-6     # inverse depth 1/4 weighted by alpha
        assert int(radii[0]) > 0
        assert torch.allclose(color[:, 0, 0], bg, atol=1e-6)    # far corner: background only


def test_two_gaussians_blend_front_to_back(oracle):
    W, H = 33, 33
    cam = synthetic.look_at_camera((4.0, 0.0, 0.0), W, H)
    rs = settings_for(oracle.Settings, cam, torch.zeros(3), 0, torch.device("cpu"))
    rast = oracle.Rasterizer(rs)
    m = torch.tensor([[-1.0, 0.0, 0.0], [1.0, 0.0, 0.0]])  # second one is nearer to the camera at x=4
    c = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
    color, _, _ = rast(means3D=m, means2D=torch.zeros_like(m), opacities=torch.tensor([[0.5], [0.5]]), colors_precomp=c,
                       scales=torch.full((2, 3), 0.3), rotations=torch.tensor([[1.0, 0, 0, 0]] * 2))
    # near (green) first: 0.5 green, then 0.5 * 0.5 red
    assert torch.allclose(color[:, 16, 16], torch.tensor([0.25, 0.5, 0.0]), atol=1e-6)


def test_mark_visible(oracle):
    rast, _ = mk(oracle)
    vis = rast.markVisible(torch.tensor([[0.0, 0, 0], [10.0, 0.5, 0.5], [2.9, 0.5, 0.5]]))
    assert vis.dtype == torch.bool and vis.tolist() == [True, False, False]
