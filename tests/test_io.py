"""On-disk formats (gsplat_amd/io.py): COLMAP readers against what the REFERENCE's own loader
returned for the committed fixture (tests/golden/colmap + colmap_expected.npz, generator make_golden.py),
PLY byte layout / round trips, camera conventions, the sparse-view split."""
import json
import math
import os
import struct

import numpy as np
import pytest
import torch

from gsplat_amd import io as gio
from gsplat_amd import synthetic

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CM = os.path.join(GOLD, "colmap")


@pytest.fixture(scope="module")
def exp():
    return np.load(os.path.join(GOLD, "colmap_expected.npz"))


def test_colmap_cameras_bin_and_txt_match_reference_loader(exp):
    cams = gio.read_intrinsics_binary(os.path.join(CM, "cameras.bin"))
    assert sorted(cams) == exp["cam_ids"].tolist()
    for k in cams:
        assert cams[k].model == str(exp["cam%d_model" % k])
        assert [cams[k].width, cams[k].height] == exp["cam%d_wh" % k].tolist()
        assert np.array_equal(cams[k].params, exp["cam%d_params" % k])
    txt = gio.read_intrinsics_text(os.path.join(CM, "cameras.txt"))
    assert txt[2].model == "PINHOLE" and np.array_equal(txt[2].params, exp["txtcam_params"])


@pytest.mark.parametrize("tag,fn,name", [("bin", gio.read_extrinsics_binary, "images.bin"),
                                        ("txt", gio.read_extrinsics_text, "images.txt")])
def test_colmap_images_match_reference_loader(exp, tag, fn, name):
    ims = fn(os.path.join(CM, name))
    assert sorted(ims) == exp["img_ids_" + tag].tolist()
    for k, im in ims.items():
        assert np.array_equal(np.concatenate((im.qvec, im.tvec)), exp["img%d_%s_qt" % (k, tag)])
        assert im.camera_id == int(exp["img%d_%s_cam" % (k, tag)]) and im.name == str(exp["img%d_%s_name" % (k, tag)])
        assert np.array_equal(im.xys.reshape(-1, 2), exp["img%d_%s_xys" % (k, tag)])   # incl. images without points
        assert np.array_equal(im.point3D_ids, exp["img%d_%s_p3d" % (k, tag)])
        R = gio.qvec2rotmat(im.qvec)
        assert np.allclose(R, exp["img%d_%s_R" % (k, tag)], rtol=0, atol=1e-15)
        assert np.allclose(gio.rotmat2qvec(R), exp["img%d_%s_q_back" % (k, tag)], rtol=0, atol=1e-12)


def test_colmap_points_match_reference_loader(exp):
    for tag, fn, name in (("bin", gio.read_points3D_binary, "points3D.bin"), ("txt", gio.read_points3D_text, "points3D.txt")):
        xyz, rgb, err = fn(os.path.join(CM, name))
        assert np.array_equal(xyz, exp["pts_xyz_" + tag]) and np.array_equal(rgb, exp["pts_rgb_" + tag])
        assert np.array_equal(err, exp["pts_err_" + tag]) and err.shape == exp["pts_err_" + tag].shape
        assert rgb.dtype == exp["pts_rgb_" + tag].dtype


def test_truncated_colmap_file_raises(tmp_path):
    raw = open(os.path.join(CM, "images.bin"), "rb").read()
    p = tmp_path / "images.bin"
    p.write_bytes(raw[:100])
    with pytest.raises(ValueError):
        gio.read_extrinsics_binary(str(p))


def test_colmap_cameras_to_camera_infos_and_view_matrices():
    """readColmapCameras conventions: R = qvec2rotmat(q)^T, T = tvec, per-model focal -> FoV; fed through the
    camera builder that the golden cameras.npz pins (getWorld2View2 etc.), world points land where COLMAP says."""
    cams = gio.read_intrinsics_binary(os.path.join(CM, "cameras.bin"))
    ims = gio.read_extrinsics_binary(os.path.join(CM, "images.bin"))
    infos = gio.read_colmap_cameras(ims, cams, images_folder="/data/images", test_cam_names=("img_03.png",))
    assert [c.image_name for c in infos] == sorted(c.image_name for c in infos)
    assert sum(c.is_test for c in infos) == 1
    by_name = {im.name: im for im in ims.values()}
    for c in infos:
        im = by_name[c.image_name]
        intr = cams[im.camera_id]
        fx = intr.params[0]
        fy = intr.params[1] if intr.model in ("PINHOLE", "OPENCV") else fx
        assert abs(c.FovX - 2 * math.atan(intr.width / (2 * fx))) < 1e-12
        assert abs(c.FovY - 2 * math.atan(intr.height / (2 * fy))) < 1e-12
        assert c.image_path == "/data/images/" + c.image_name and (c.width, c.height) == (intr.width, intr.height)
        cam = synthetic.make_camera(c.R, c.T, c.FovX, c.FovY, c.width, c.height)
        X = np.array([0.3, -0.2, 0.9])
        want = gio.qvec2rotmat(im.qvec) @ X + im.tvec            # COLMAP: x_cam = R x_world + t
        got = (torch.tensor(np.append(X, 1.0), dtype=torch.float32) @ cam.world_view_transform)[:3]
        assert np.allclose(got.numpy(), want, atol=1e-5)


def test_sparse_view_split():
    infos = [gio.CameraInfo(i, np.eye(3), np.zeros(3), 1.0, 1.0, "", "im%03d.jpg" % i, 8, 8, False) for i in range(50)]
    train, test = gio.split_train_test(infos[::-1], eval_mode=True, llffhold=8, n_views=3)
    assert [c.image_name for c in test] == ["im%03d.jpg" % i for i in range(0, 50, 8)]
    full = [i for i in range(50) if i % 8 != 0]
    idx = np.linspace(0, len(full) - 1, 3, dtype=int)
    assert [c.image_name for c in train] == ["im%03d.jpg" % full[i] for i in idx]
    train, test = gio.split_train_test(infos, eval_mode=False, n_views=0)
    assert len(train) == 50 and test == []


def test_gaussian_ply_layout_and_round_trip(tmp_path):
    rng = np.random.RandomState(3)
    P = 7
    xyz, feat = rng.randn(P, 3).astype(np.float32), rng.randn(P, 16, 3).astype(np.float32)
    op, sc, rot = rng.randn(P, 1).astype(np.float32), rng.randn(P, 3).astype(np.float32), rng.randn(P, 4).astype(np.float32)
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    gio.save_gaussians_ply(path, xyz, feat, op, sc, rot)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().strip().split("\n")
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 7"]
    names = [ln.split()[2] for ln in lines[3:]]
    assert all(ln.startswith("property float ") for ln in lines[3:]) and len(names) == 62
    assert names[:9] == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    assert names[9] == "f_rest_0" and names[53] == "f_rest_44" and names[54:] == \
        ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    assert len(body) == P * 62 * 4
    row = np.frombuffer(body, dtype="<f4").reshape(P, 62)
    assert np.array_equal(row[:, 0:3], xyz) and not row[:, 3:6].any()
    assert np.array_equal(row[:, 6:9], feat[:, 0, :])
    # channel-major rest block: f_rest_{c*15 + j} = coefficient 1+j of channel c  (transpose(1, 2), :246)
    for c in range(3):
        for j in (0, 7, 14):
            assert np.array_equal(row[:, 9 + c * 15 + j], feat[:, 1 + j, c])
    assert np.array_equal(row[:, 54], op[:, 0]) and np.array_equal(row[:, 55:58], sc) and np.array_equal(row[:, 58:], rot)
    back = gio.load_gaussians_ply(path)
    for k, v in (("xyz", xyz), ("features", feat), ("opacity", op), ("scaling", sc), ("rotation", rot)):
        assert back[k].dtype == np.float32 and np.array_equal(back[k], v), k
    with pytest.raises(AssertionError):
        gio.load_gaussians_ply(path, max_sh_degree=2)


def test_point_cloud_ply_ascii_big_endian_and_extra_elements(tmp_path):
    xyz = np.array([[0.5, -1.25, 2.0], [3.0, 4.5, -6.0]], dtype=np.float32)
    rgb = np.array([[255, 0, 17], [1, 2, 3]], dtype=np.uint8)
    p = str(tmp_path / "points3D.ply")
    gio.store_ply(p, xyz, rgb)
    pc = gio.fetch_ply(p)
    assert np.array_equal(pc.points, xyz) and np.allclose(pc.colors, rgb / 255.0) and not pc.normals.any()
    head = ("ply\nformat %s 1.0\ncomment made by hand\nelement vertex 2\nproperty float x\nproperty float y\n"
            "property float z\nproperty float nx\nproperty float ny\nproperty float nz\nproperty uchar red\n"
            "property uchar green\nproperty uchar blue\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n")
    a = str(tmp_path / "ascii.ply")
    with open(a, "w") as f:
        f.write(head % "ascii")
        for i in range(2):
            f.write("%r %r %r 0 0 0 %d %d %d\n" % (*[float(v) for v in xyz[i]], *[int(v) for v in rgb[i]]))
        f.write("3 0 1 1\n")
    b = str(tmp_path / "be.ply")
    with open(b, "wb") as f:
        f.write((head % "binary_big_endian").encode())
        for i in range(2):
            f.write(struct.pack(">ffffffBBB", *[float(v) for v in xyz[i]], 0, 0, 0, *[int(v) for v in rgb[i]]))
        f.write(struct.pack(">Biii", 3, 0, 1, 1))
    for path in (a, b):
        pc2 = gio.fetch_ply(path)
        assert np.array_equal(pc2.points, xyz) and np.allclose(pc2.colors, rgb / 255.0)
    with pytest.raises(ValueError):
        gio.read_ply(a, element="edge")
    with pytest.raises(ValueError):
        (tmp_path / "bad.ply").write_bytes(b"plx\n")
        gio.read_ply(str(tmp_path / "bad.ply"))


def test_transforms_json_cameras(tmp_path):
    ang = 0.6911112070083618
    c2w = np.array([[0.0, -0.6, 0.8, 3.2], [1.0, 0.0, 0.0, 0.0], [0.0, 0.8, 0.6, 2.4], [0, 0, 0, 1.0]])
    with open(tmp_path / "transforms_train.json", "w") as f:
        json.dump({"camera_angle_x": ang, "frames": [{"file_path": "./train/r_0", "transform_matrix": c2w.tolist()}]}, f)
    cams = gio.read_cameras_from_transforms(str(tmp_path), "transforms_train.json", image_size=(400, 300))
    c = cams[0]
    assert c.image_name == "r_0" and c.image_path.endswith("train/r_0.png") and (c.width, c.height) == (400, 300)
    assert c.FovX == ang and abs(c.FovY - 2 * math.atan(300 / (2 * (400 / (2 * math.tan(ang / 2)))))) < 1e-15
    flipped = c2w.copy()
    flipped[:3, 1:3] *= -1
    w2c = np.linalg.inv(flipped)
    assert np.allclose(c.R, w2c[:3, :3].T) and np.allclose(c.T, w2c[:3, 3])
    # the camera centre recovered from (R, T) is the c2w translation, and it looks along its +z (COLMAP axes)
    n = gio.nerfpp_norm(cams)
    assert np.allclose(-n["translate"], c2w[:3, 3]) and n["radius"] == 0.0
    j = gio.camera_to_json(0, c)
    assert np.allclose(j["position"], c2w[:3, 3]) and np.allclose(j["rotation"], flipped[:3, :3])
    assert abs(j["fx"] - 400 / (2 * math.tan(ang / 2))) < 1e-9 and j["img_name"] == "r_0"
    gio.write_cameras_json(str(tmp_path / "cameras.json"), cams)
    assert json.load(open(tmp_path / "cameras.json"))[0]["width"] == 400


def test_rgba_composite_and_extent():
    rgba = np.array([[[255, 128, 0, 255], [255, 128, 0, 0]], [[10, 20, 30, 128], [200, 100, 50, 64]]], dtype=np.uint8)
    w = gio.composite_rgba(rgba, True)
    k = gio.composite_rgba(rgba, False)
    assert w.shape == (3, 2, 2) and w.dtype == np.float32
    assert np.allclose(w[:, 0, 0], [1.0, 128 / 255, 0.0]) and np.allclose(w[:, 0, 1], [1, 1, 1]) and np.allclose(k[:, 0, 1], 0)
    a = 128 / 255.0
    assert np.allclose(k[:, 1, 0] * 255, [int(10 * a), int(20 * a), int(30 * a)])      # truncation, not rounding
    assert np.allclose(w[:, 1, 0] * 255, [int(10 * a + 255 * (1 - a)), int(20 * a + 255 * (1 - a)), int(30 * a + 255 * (1 - a))])
    from gsplat_amd.trainer import cameras_extent
    cams = synthetic.orbit_cameras(64, 64)
    infos = []
    for c in cams:
        w2c = c.world_view_transform.numpy().T.astype(np.float64)
        infos.append(gio.CameraInfo(0, w2c[:3, :3].T, w2c[:3, 3], c.FoVy, c.FoVx, "", "x", 64, 64, False))
    assert abs(gio.nerfpp_norm(infos)["radius"] - cameras_extent([c.camera_center for c in cams])) < 1e-4


def test_model_ply_and_checkpoint_round_trip(tmp_path, oracle):
    from gsplat_amd.trainer import GaussianModelLite
    rng = np.random.RandomState(0)
    pts, col = rng.uniform(-1, 1, (50, 3)), rng.uniform(0, 1, (50, 3))
    m = GaussianModelLite.create_from_pcd(pts, col, torch.device("cpu"), api=oracle.api)
    assert m.P == 50 and m.active_sh_degree == 0
    assert torch.allclose(torch.sigmoid(m.params["opacity"]), torch.full((50, 1), 0.1), atol=1e-6)
    assert torch.allclose(m.params["features"][:, 0, :], (torch.tensor(col, dtype=torch.float32) - 0.5) / 0.28209479177387814)
    assert float(m.params["features"][:, 1:, :].abs().max()) == 0
    d2 = synthetic.brute_force_knn_dist2(torch.tensor(pts, dtype=torch.float32))
    assert torch.allclose(torch.exp(m.params["scaling"]), torch.sqrt(d2)[:, None].repeat(1, 3), rtol=1e-5)
    g = torch.Generator().manual_seed(0)
    for _ in range(3):
        m.flat_grad.copy_(torch.randn(m.flat.numel(), generator=g) * 1e-2)
        m.optimizer.step()
    path = str(tmp_path / "pc.ply")
    m.save_ply(path)
    m2 = GaussianModelLite.create_from_pcd(pts[:10], col[:10], torch.device("cpu"), api=oracle.api)
    m2.load_ply(path)
    assert m2.P == 50 and m2.active_sh_degree == 3 and torch.equal(m2.flat, m.flat)
    assert float(m2.optimizer.exp_avg.abs().max()) == 0 and m2.optimizer.exp_avg.numel() == m.flat.numel()
    ck = str(tmp_path / "chkpnt3.pth")
    torch.save((m.capture(), 3), ck)
    state, it = torch.load(ck)
    m3 = GaussianModelLite.create_from_pcd(pts[:5], col[:5], torch.device("cpu"), api=oracle.api)
    m3.restore(state)
    assert it == 3 and torch.equal(m3.flat, m.flat) and torch.equal(m3.optimizer.exp_avg_sq, m.optimizer.exp_avg_sq)
    grad = torch.randn(m.flat.numel(), generator=g) * 1e-2
    for mm in (m, m3):
        mm.flat_grad.copy_(grad)
        mm.optimizer.step()
    assert torch.equal(m3.flat, m.flat) and m3.optimizer.seg_steps == m.optimizer.seg_steps


def test_nir_model_ply_round_trip(tmp_path, oracle):
    from gsplat_amd.trainer import GaussianModelLite
    sc = synthetic.trained_like(30, seed=2)
    m = GaussianModelLite(sc, torch.device("cpu"), api=oracle.api, with_nir=True)
    with torch.no_grad():
        m.params["nir_albedo"].copy_(torch.linspace(-2, 2, 30)[:, None])
    p = str(tmp_path / "nir.ply")
    m.save_ply(p)
    v = gio.read_ply(p)
    assert v.dtype.names[-1] == "nir_albedo" and len(v.dtype.names) == 63
    m2 = GaussianModelLite(synthetic.trained_like(5, seed=3), torch.device("cpu"), api=oracle.api, with_nir=True)
    m2.load_ply(p)
    assert m2.P == 30 and torch.equal(m2.flat, m.flat)
    plain = str(tmp_path / "plain.ply")
    GaussianModelLite(sc, torch.device("cpu"), api=oracle.api).save_ply(plain)
    m2.load_ply(plain)  # no NIR property: albedo starts from the DC coefficient
    assert torch.equal(m2.params["nir_albedo"].detach()[:, 0], m2.params["features"].detach()[:, 0, 0])
