"""Pins the oracle (and the host-side camera code) against golden vectors produced by the
reference's own importable Python (tests/golden/make_golden.py)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from gsplat_amd import synthetic

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_colour_forward_backward_vs_reference_eval_sh(oracle, deg):
    z = np.load(os.path.join(G, "sh_colour.npz"))
    means, campos, sh, w = (np.ascontiguousarray(z[k], dtype=np.float32) for k in ("means", "campos", "sh", "w"))
    P = means.shape[0]
    rgb = np.zeros((P, 3), np.float32)
    clamped = np.zeros((P, 3), np.uint8)
    assert oracle.lib.gso_test_sh_fwd(P, deg, 16, _p(means), _p(campos), _p(sh), _p(rgb), _p(clamped)) == 0
    ref = z["colors_deg%d" % deg]
    assert np.abs(rgb - ref).max() < 2e-6
    raw = z["raw_deg%d" % deg]
    safe = np.abs(raw) > 1e-5  # away from the clamp boundary the flag is unambiguous
    assert np.array_equal((raw < 0)[safe], clamped.astype(bool)[safe])
    assert clamped.sum() > 10  # the fixture does exercise clamping

    dmeans = np.zeros((P, 3), np.float32)
    dsh = np.zeros((P, 16, 3), np.float32)
    assert oracle.lib.gso_test_sh_bwd(P, deg, 16, _p(means), _p(campos), _p(sh), _p(clamped), _p(w), _p(dmeans),
                                      _p(dsh)) == 0
    ref_dsh = z["dsh_deg%d" % deg]
    assert np.abs(dsh - ref_dsh).max() < 1e-5 * max(1.0, np.abs(ref_dsh).max())
    ref_dm = z["dmeans_deg%d" % deg]
    assert np.abs(dmeans - ref_dm).max() < 2e-5 * max(1.0, np.abs(ref_dm).max())
    assert np.all(dsh[:, (deg + 1) ** 2:, :] == 0)


def test_camera_matrices_vs_reference_graphics_utils():
    z = np.load(os.path.join(G, "cameras.npz"))
    for i in range(z["R"].shape[0]):
        W, H = int(z["WH"][i][0]), int(z["WH"][i][1])
        fovx = float(z["FoVx"][i])
        fovy = synthetic.focal2fov(synthetic.fov2focal(fovx, W), H)
        assert abs(fovy - float(z["FoVy"][i])) < 1e-12
        cam = synthetic.make_camera(z["R"][i], z["t"][i], fovx, fovy, W, H)
        assert np.array_equal(cam.world_view_transform.numpy(), z["world_view_transform"][i])
        assert np.allclose(cam.full_proj_transform.numpy(), z["full_proj_transform"][i], rtol=0, atol=1e-6)
        assert np.allclose(cam.camera_center.numpy(), z["camera_center"][i], rtol=0, atol=1e-6)


def test_rgb2sh_inverse_sigmoid_vs_reference():
    z = np.load(os.path.join(G, "schedule.npz"))
    assert np.allclose(synthetic.rgb2sh(torch.tensor(z["rgb"])).numpy(), z["rgb2sh"], atol=1e-7)
    assert np.allclose(synthetic.inverse_sigmoid(torch.tensor(z["inv_sig_x"])).numpy(), z["inv_sig"], atol=1e-6)
