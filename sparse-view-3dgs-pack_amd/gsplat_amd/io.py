"""On-disk formats either side of the hot path - pure host code, numpy only.

  Gaussian model PLY     LGDWT-GS/scene/gaussian_model.py:225-256 (save_ply), :263-314 (load_ply)
  point-cloud PLY        LGDWT-GS/scene/dataset_readers.py:163-186 (fetchPly / storePly)
  COLMAP model           LGDWT-GS/scene/colmap_loader.py:43-258 (cameras / images / points3D, .bin and .txt)
  camera records         LGDWT-GS/scene/dataset_readers.py:71-160 (readColmapCameras), :331-374 (transforms_*.json)
  sparse-view split      LGDWT-GS/scene/dataset_readers.py:223-256 (llffhold, n_views linspace)
  cameras.json           LGDWT-GS/utils/camera_utils.py:77-96

`plyfile` is not installed in this image, so PLY is read and written here directly: header parsing for
ascii / binary_little_endian / binary_big_endian, scalar properties of any PLY type; list properties are
understood in elements that FOLLOW the requested one only as far as needed to skip them (ascii) or rejected
(binary) - the reference's files hold a single `vertex` element of scalar properties.
"""
import collections
import json
import math
import os
import struct

import numpy as np

# ------------------------------------------------------------------------------------------------ PLY
_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
              "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
              "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}
_PLY_NAMES = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float",
              "f8": "double"}


def read_ply(path, element="vertex"):
    """Returns a numpy structured array with one field per scalar property of `element`."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s: not a PLY file" % path)
        fmt = None
        elements = []  # [name, count, [(prop, dtype) | (prop, None) for lists]]
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: truncated PLY header" % path)
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append([tok[1], int(tok[2]), []])
            elif tok[0] == "property":
                if tok[1] == "list":
                    elements[-1][2].append((tok[4], None))
                else:
                    if tok[1] not in _PLY_TYPES:
                        raise ValueError("%s: unknown PLY type %s" % (path, tok[1]))
                    elements[-1][2].append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
            raise ValueError("%s: unsupported PLY format %r" % (path, fmt))
        for name, count, props in elements:
            has_list = any(t is None for _, t in props)
            if name == element:
                if has_list:
                    raise ValueError("%s: list properties in element %s are not supported" % (path, element))
                if fmt == "ascii":
                    dt = np.dtype([(p, t) for p, t in props])
                    out = np.empty(count, dtype=dt)
                    for i in range(count):
                        vals = f.readline().split()
                        out[i] = tuple(np.array(v, dtype=np.float64).astype(t) for v, (p, t) in zip(vals, props))
                    return out
                end = "<" if fmt == "binary_little_endian" else ">"
                dt = np.dtype([(p, end + t) for p, t in props])
                raw = f.read(count * dt.itemsize)
                if len(raw) != count * dt.itemsize:
                    raise ValueError("%s: truncated PLY body" % path)
                return np.frombuffer(raw, dtype=dt).astype(np.dtype([(p, t) for p, t in props]))
            # skip an element that precedes the requested one
            if fmt == "ascii":
                for _ in range(count):
                    f.readline()
            else:
                if has_list:
                    raise ValueError("%s: cannot skip binary element %s with list properties" % (path, name))
                f.seek(count * sum(np.dtype(t).itemsize for _, t in props), os.SEEK_CUR)
    raise ValueError("%s: no element %s" % (path, element))


def write_ply(path, data, element="vertex", text=False):
    """data: numpy structured array of scalar fields -> `format binary_little_endian 1.0` (what plyfile writes on
    the reference's machines) or ascii."""
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)  # mkdir_p(os.path.dirname(path)), gaussian_model.py:241
    names = data.dtype.names
    kinds = [data.dtype[n].str.lstrip("<>=|") for n in names]
    head = ["ply", "format %s 1.0" % ("ascii" if text else "binary_little_endian"),
            "element %s %d" % (element, data.shape[0])]
    head += ["property %s %s" % (_PLY_NAMES[k], n) for n, k in zip(names, kinds)]
    head.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\n").encode("ascii"))
        if text:
            for row in data:
                f.write((" ".join(repr(v.item()) for v in row) + "\n").encode("ascii"))
        else:
            f.write(data.astype(np.dtype([(n, "<" + k) for n, k in zip(names, kinds)])).tobytes())


def gaussian_attribute_names(n_rest=45):
    """construct_list_of_attributes (gaussian_model.py:225-238): 62 float properties at SH degree 3."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += ["f_dc_%d" % i for i in range(3)]
    names += ["f_rest_%d" % i for i in range(n_rest)]
    names.append("opacity")
    names += ["scale_%d" % i for i in range(3)]
    names += ["rot_%d" % i for i in range(4)]
    return names


def save_gaussians_ply(path, xyz, features, opacity, scaling, rotation, extra=None):
    """save_ply (gaussian_model.py:240-256).  All arguments are the RAW (pre-activation) parameters as arrays:
    xyz [P,3], features [P,K,3] (coefficient-major: DC first, as the rasterizer reads them), opacity [P,1],
    scaling [P,3], rotation [P,4].  The file stores the SH block channel-major (`transpose(1, 2)`), DC and rest
    separately, and zero normals.  extra: dict name -> [P] / [P,1] array appended as further float properties in
    the given order (the multispectral variant appends `nir_albedo`, mult-dwtgs/scene/gaussian_model.py:317-338)."""
    xyz, features, opacity, scaling, rotation = (np.asarray(a, dtype=np.float32) for a in
                                                 (xyz, features, opacity, scaling, rotation))
    P, K = xyz.shape[0], features.shape[1]
    f_dc = features[:, 0, :]                                               # [P,3] = [P,1,3]^T flattened
    f_rest = np.transpose(features[:, 1:, :], (0, 2, 1)).reshape(P, 3 * (K - 1))
    cols = np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, opacity.reshape(P, 1), scaling, rotation), axis=1)
    names = gaussian_attribute_names(3 * (K - 1))
    for k, v in (extra or {}).items():
        names.append(k)
        cols = np.concatenate((cols, np.asarray(v, dtype=np.float32).reshape(P, 1)), axis=1)
    out = np.empty(P, dtype=[(n, "f4") for n in names])
    for i, n in enumerate(names):
        out[n] = cols[:, i]
    write_ply(path, out)


def load_gaussians_ply(path, max_sh_degree=3):
    """load_ply (gaussian_model.py:263-314) -> dict(xyz, features [P,(D+1)^2,3], opacity [P,1], scaling, rotation),
    float32, raw parameters; `f_rest_*`, `scale_*`, `rot_*` are ordered by their numeric suffix like the reference
    and the number of rest coefficients must match max_sh_degree (the reference asserts)."""
    v = read_ply(path)
    names = v.dtype.names
    P = v.shape[0]

    def numbered(prefix):
        cols = sorted((n for n in names if n.startswith(prefix)), key=lambda n: int(n.split("_")[-1]))
        return cols
    xyz = np.stack((v["x"], v["y"], v["z"]), axis=1).astype(np.float32)
    rest_names = numbered("f_rest_")
    K = (max_sh_degree + 1) ** 2
    if len(rest_names) != 3 * K - 3:
        raise AssertionError("%s holds %d f_rest properties, SH degree %d needs %d"
                             % (path, len(rest_names), max_sh_degree, 3 * K - 3))
    features = np.zeros((P, K, 3), dtype=np.float32)
    for c in range(3):
        features[:, 0, c] = v["f_dc_%d" % c]
    if rest_names:
        rest = np.stack([v[n] for n in rest_names], axis=1).reshape(P, 3, K - 1)  # channel-major on disk
        features[:, 1:, :] = np.transpose(rest, (0, 2, 1))
    scaling = np.stack([v[n] for n in numbered("scale_")], axis=1).astype(np.float32)
    rotation = np.stack([v[n] for n in numbered("rot")], axis=1).astype(np.float32)
    opacity = np.asarray(v["opacity"], dtype=np.float32)[:, None]
    out = dict(xyz=xyz, features=features, opacity=opacity, scaling=scaling, rotation=rotation)
    if "nir_albedo" in names:  # multispectral models (mult-dwtgs/scene/gaussian_model.py:411-416)
        out["nir_albedo"] = np.asarray(v["nir_albedo"], dtype=np.float32)[:, None]
    return out


BasicPointCloud = collections.namedtuple("BasicPointCloud", ["points", "colors", "normals"])


def fetch_ply(path):
    """fetchPly (dataset_readers.py:163-169): colours scaled to [0, 1]."""
    v = read_ply(path)
    pts = np.vstack([v["x"], v["y"], v["z"]]).T
    col = np.vstack([v["red"], v["green"], v["blue"]]).T / 255.0
    nrm = np.vstack([v["nx"], v["ny"], v["nz"]]).T
    return BasicPointCloud(points=pts, colors=col, normals=nrm)


def store_ply(path, xyz, rgb):
    """storePly (dataset_readers.py:171-186): float32 positions, zero normals, uchar colours."""
    xyz = np.asarray(xyz)
    out = np.empty(xyz.shape[0], dtype=[("x", "f4"), ("y", "f4"), ("z", "f4"), ("nx", "f4"), ("ny", "f4"),
                                        ("nz", "f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    for i, n in enumerate(("x", "y", "z")):
        out[n] = xyz[:, i]
    out["nx"] = out["ny"] = out["nz"] = 0
    rgb = np.asarray(rgb)
    for i, n in enumerate(("red", "green", "blue")):
        out[n] = rgb[:, i]
    write_ply(path, out)


# ------------------------------------------------------------------------------------------------ COLMAP
CameraModel = collections.namedtuple("CameraModel", ["model_id", "model_name", "num_params"])
ColmapCamera = collections.namedtuple("Camera", ["id", "model", "width", "height", "params"])
ColmapImage = collections.namedtuple("Image", ["id", "qvec", "tvec", "camera_id", "name", "xys", "point3D_ids"])
CAMERA_MODELS = [CameraModel(0, "SIMPLE_PINHOLE", 3), CameraModel(1, "PINHOLE", 4), CameraModel(2, "SIMPLE_RADIAL", 4),
                 CameraModel(3, "RADIAL", 5), CameraModel(4, "OPENCV", 8), CameraModel(5, "OPENCV_FISHEYE", 8),
                 CameraModel(6, "FULL_OPENCV", 12), CameraModel(7, "FOV", 5), CameraModel(8, "SIMPLE_RADIAL_FISHEYE", 4),
                 CameraModel(9, "RADIAL_FISHEYE", 5), CameraModel(10, "THIN_PRISM_FISHEYE", 12)]
CAMERA_MODEL_IDS = {m.model_id: m for m in CAMERA_MODELS}
CAMERA_MODEL_NAMES = {m.model_name: m for m in CAMERA_MODELS}


def qvec2rotmat(q):
    """colmap_loader.py:43-53 (w, x, y, z)."""
    w, x, y, z = (float(v) for v in q)
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                     [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                     [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


def rotmat2qvec(R):
    """colmap_loader.py:55-66: eigenvector of the largest eigenvalue of the symmetric 4x4 K matrix, w >= 0."""
    R = np.asarray(R, dtype=np.float64)
    Rxx, Ryx, Rzx, Rxy, Ryy, Rzy, Rxz, Ryz, Rzz = R.flat
    K = np.array([[Rxx - Ryy - Rzz, 0, 0, 0], [Ryx + Rxy, Ryy - Rxx - Rzz, 0, 0],
                  [Rzx + Rxz, Rzy + Ryz, Rzz - Rxx - Ryy, 0],
                  [Ryz - Rzy, Rzx - Rxz, Rxy - Ryx, Rxx + Ryy + Rzz]]) / 3.0
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    return -q if q[0] < 0 else q


def _unpack(f, fmt):
    n = struct.calcsize("<" + fmt)
    b = f.read(n)
    if len(b) != n:
        raise ValueError("truncated COLMAP file")
    return struct.unpack("<" + fmt, b)


def read_intrinsics_binary(path):
    """cameras.bin (colmap_loader.py:202-228): u64 count, then per camera i32 id, i32 model, u64 w, u64 h, f64 params."""
    cams = {}
    with open(path, "rb") as f:
        for _ in range(_unpack(f, "Q")[0]):
            cid, model, w, h = _unpack(f, "iiQQ")
            m = CAMERA_MODEL_IDS[model]
            cams[cid] = ColmapCamera(cid, m.model_name, w, h, np.array(_unpack(f, "d" * m.num_params)))
    return cams


def read_extrinsics_binary(path):
    """images.bin (colmap_loader.py:167-199): per image i32 id, f64 qvec[4], f64 tvec[3], i32 camera id, zero-terminated
    name, u64 n, n x (f64 x, f64 y, i64 point3D id)."""
    images = {}
    with open(path, "rb") as f:
        for _ in range(_unpack(f, "Q")[0]):
            p = _unpack(f, "idddddddi")
            name = b""
            while True:
                c = f.read(1)
                if c == b"\x00" or not c:
                    break
                name += c
            n = _unpack(f, "Q")[0]
            rec = np.frombuffer(f.read(24 * n), dtype=np.dtype([("x", "<f8"), ("y", "<f8"), ("id", "<i8")]))
            if rec.shape[0] != n:
                raise ValueError("truncated COLMAP file")
            images[p[0]] = ColmapImage(p[0], np.array(p[1:5]), np.array(p[5:8]), p[8], name.decode("utf-8"),
                                       np.column_stack((rec["x"], rec["y"])), rec["id"].copy())
    return images


def read_points3D_binary(path):
    """points3D.bin (colmap_loader.py:113-141) -> (xyz [N,3] f64, rgb [N,3], error [N,1]); tracks are skipped."""
    with open(path, "rb") as f:
        n = _unpack(f, "Q")[0]
        xyz, rgb, err = np.empty((n, 3)), np.empty((n, 3)), np.empty((n, 1))
        for i in range(n):
            p = _unpack(f, "QdddBBBd")
            xyz[i], rgb[i], err[i] = p[1:4], p[4:7], p[7]
            f.seek(8 * _unpack(f, "Q")[0], os.SEEK_CUR)
    return xyz, rgb, err


def _data_lines(path):
    with open(path, "r") as f:
        for line in f:
            line = line.strip()
            if line and not line.startswith("#"):
                yield line


def read_intrinsics_text(path):
    """cameras.txt: `id MODEL width height params...`.  (The reference's text reader asserts PINHOLE,
    colmap_loader.py:158-160; every model of the table is accepted here.)"""
    cams = {}
    for line in _data_lines(path):
        e = line.split()
        cams[int(e[0])] = ColmapCamera(int(e[0]), e[1], int(e[2]), int(e[3]), np.array([float(x) for x in e[4:]]))
    return cams


def read_extrinsics_text(path):
    """images.txt (colmap_loader.py:231-258): two lines per image (pose + name, then the 2D points)."""
    images = {}
    it = _data_lines_keep_empty(path)
    for line in it:
        e = line.split()
        pts = next(it, "").split()
        xys = np.column_stack((np.array(pts[0::3], dtype=np.float64), np.array(pts[1::3], dtype=np.float64))) \
            if pts else np.zeros((0, 2))
        ids = np.array(pts[2::3], dtype=np.int64)
        images[int(e[0])] = ColmapImage(int(e[0]), np.array([float(x) for x in e[1:5]]),
                                        np.array([float(x) for x in e[5:8]]), int(e[8]), e[9], xys, ids)
    return images


def _data_lines_keep_empty(path):
    # the 2D-point line of an image without observations is empty and must still be consumed
    with open(path, "r") as f:
        pose_next = True
        for line in f:
            s = line.strip()
            if s.startswith("#"):
                continue
            if pose_next and not s:
                continue
            yield s
            pose_next = not pose_next


def read_points3D_text(path):
    """points3D.txt (colmap_loader.py:83-110): `id x y z r g b error track...`.  Shapes / dtypes as the reference's
    text reader returns them (rgb int64 [N,3], error [N] - its binary reader gives float64 [N,3] and [N,1])."""
    rows = [line.split() for line in _data_lines(path)]
    xyz = np.array([[float(v) for v in r[1:4]] for r in rows]).reshape(-1, 3)
    rgb = np.array([[int(v) for v in r[4:7]] for r in rows], dtype=np.int64).reshape(-1, 3)
    err = np.array([float(r[7]) for r in rows])
    return xyz, rgb, err


def write_colmap_binary(dirpath, cameras, images, points=None):
    """Writer for the three .bin files (used to build test fixtures and to export synthetic scenes)."""
    os.makedirs(dirpath, exist_ok=True)
    with open(os.path.join(dirpath, "cameras.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(cameras)))
        for c in cameras.values():
            m = CAMERA_MODEL_NAMES[c.model]
            f.write(struct.pack("<iiQQ", c.id, m.model_id, c.width, c.height))
            f.write(struct.pack("<" + "d" * m.num_params, *[float(x) for x in c.params]))
    with open(os.path.join(dirpath, "images.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(images)))
        for im in images.values():
            f.write(struct.pack("<idddddddi", im.id, *[float(x) for x in im.qvec], *[float(x) for x in im.tvec],
                                im.camera_id))
            f.write(im.name.encode("utf-8") + b"\x00")
            f.write(struct.pack("<Q", len(im.point3D_ids)))
            for (x, y), pid in zip(im.xys, im.point3D_ids):
                f.write(struct.pack("<ddq", float(x), float(y), int(pid)))
    if points is not None:
        xyz, rgb, err = points
        with open(os.path.join(dirpath, "points3D.bin"), "wb") as f:
            f.write(struct.pack("<Q", len(xyz)))
            for i in range(len(xyz)):
                f.write(struct.pack("<QdddBBBd", i + 1, *[float(v) for v in xyz[i]], *[int(v) for v in rgb[i]],
                                    float(err[i][0])))
                f.write(struct.pack("<Q", 0))


# ------------------------------------------------------------------------------------------------ cameras
CameraInfo = collections.namedtuple("CameraInfo", ["uid", "R", "T", "FovY", "FovX", "image_path", "image_name", "width",
                                                   "height", "is_test"])


from .synthetic import focal2fov, fov2focal  # noqa: E402  (pinhole field of view <-> focal length in pixels)


# models whose first TWO parameters are (fx, fy); the others start with a single focal length
_TWO_FOCALS = ("PINHOLE", "OPENCV", "OPENCV_FISHEYE", "FULL_OPENCV", "THIN_PRISM_FISHEYE")


def read_colmap_cameras(cam_extrinsics, cam_intrinsics, images_folder="", test_cam_names=()):
    """readColmapCameras (dataset_readers.py:71-160): R = qvec2rotmat(q)^T (stored transposed 'due to glm'),
    T = tvec, FoV from the focal length(s) of the camera model; sorted by image name like readColmapSceneInfo."""
    out = []
    for key in cam_extrinsics:
        ext = cam_extrinsics[key]
        intr = cam_intrinsics[ext.camera_id]
        fx = float(intr.params[0])
        fy = float(intr.params[1]) if intr.model in _TWO_FOCALS else fx
        # (a model outside the table falls back to "first parameter is the focal length", dataset_readers.py:135-139)
        out.append(CameraInfo(uid=intr.id, R=np.transpose(qvec2rotmat(ext.qvec)), T=np.array(ext.tvec),
                              FovY=focal2fov(fy, intr.height), FovX=focal2fov(fx, intr.width),
                              image_path=os.path.join(images_folder, ext.name), image_name=ext.name,
                              width=intr.width, height=intr.height, is_test=ext.name in test_cam_names))
    return sorted(out, key=lambda c: c.image_name)


def split_train_test(cam_infos, eval_mode, llffhold=8, n_views=0, train_test_exp=False):
    """dataset_readers.py:223-256: every llffhold-th camera (by sorted name) is a test view; with n_views > 0 the
    training set is thinned to n_views evenly spaced cameras (`np.linspace(0, n-1, n_views, dtype=int)`)."""
    names = sorted(c.image_name for c in cam_infos)
    test_names = set(n for i, n in enumerate(names) if eval_mode and llffhold and i % llffhold == 0)
    cams = [c._replace(is_test=c.image_name in test_names) for c in sorted(cam_infos, key=lambda c: c.image_name)]
    train = [c for c in cams if train_test_exp or not c.is_test]
    test = [c for c in cams if c.is_test]
    if n_views > 0 and len(train) > n_views:
        train = [train[i] for i in np.linspace(0, len(train) - 1, n_views, dtype=int)]
    return train, test


def read_cameras_from_transforms(path, transformsfile, is_test=False, extension=".png", image_size=None):
    """readCamerasFromTransforms (dataset_readers.py:331-374) without opening the images: NeRF c2w (OpenGL axes) ->
    flip y, z -> invert -> R = w2c[:3,:3]^T, T = w2c[:3,3]; FovY from FovX and the image size (`image_size` =
    (width, height); the reference reads it from the image file, 800x800 for NeRF-synthetic)."""
    with open(os.path.join(path, transformsfile)) as f:
        contents = json.load(f)
    fovx = contents["camera_angle_x"]
    out = []
    for idx, frame in enumerate(contents["frames"]):
        cam_name = os.path.join(path, frame["file_path"] + extension)
        c2w = np.array(frame["transform_matrix"], dtype=np.float64)
        c2w[:3, 1:3] *= -1
        w2c = np.linalg.inv(c2w)
        w, h = image_size if image_size is not None else (800, 800)
        out.append(CameraInfo(uid=idx, R=np.transpose(w2c[:3, :3]), T=w2c[:3, 3],
                              FovY=focal2fov(fov2focal(fovx, w), h), FovX=fovx, image_path=cam_name,
                              image_name=os.path.splitext(os.path.basename(cam_name))[0], width=w, height=h,
                              is_test=is_test))
    return out


def composite_rgba(rgba_u8, white_background):
    """dataset_readers.py:353-359 + general_utils.py:21-27: alpha-composite onto the background, requantise through
    a signed byte cast exactly as `np.array(arr * 255.0, dtype=np.byte)` does, return float [3,H,W] in [0,1]."""
    norm = np.asarray(rgba_u8, dtype=np.float64) / 255.0
    bg = np.array([1.0, 1.0, 1.0]) if white_background else np.array([0.0, 0.0, 0.0])
    arr = norm[:, :, :3] * norm[:, :, 3:4] + bg * (1 - norm[:, :, 3:4])
    q = (arr * 255.0).astype(np.int64).astype(np.uint8)  # truncation toward zero, then the byte's bit pattern
    return np.transpose(q.astype(np.float32) / 255.0, (2, 0, 1))


def nerfpp_norm(cam_infos):
    """getNerfppNorm (dataset_readers.py:48-69) -> dict(translate, radius)."""
    centers = []
    for c in cam_infos:
        Rt = np.zeros((4, 4))
        Rt[:3, :3] = c.R.transpose()
        Rt[:3, 3] = c.T
        Rt[3, 3] = 1.0
        centers.append(np.linalg.inv(Rt)[:3, 3])
    centers = np.stack(centers, axis=1)
    center = centers.mean(axis=1, keepdims=True)
    radius = float(np.linalg.norm(centers - center, axis=0).max() * 1.1)
    return {"translate": -center.flatten(), "radius": radius}


def camera_to_json(cid, cam):
    """camera_to_JSON (camera_utils.py:77-96)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = cam.R.transpose()
    Rt[:3, 3] = cam.T
    Rt[3, 3] = 1.0
    c2w = np.linalg.inv(Rt)
    return {"id": cid, "img_name": cam.image_name, "width": cam.width, "height": cam.height,
            "position": c2w[:3, 3].tolist(), "rotation": [r.tolist() for r in c2w[:3, :3]],
            "fy": fov2focal(cam.FovY, cam.height), "fx": fov2focal(cam.FovX, cam.width)}


def write_cameras_json(path, cam_infos):
    """scene/__init__.py:62-70: cameras.json next to the model."""
    with open(path, "w") as f:
        json.dump([camera_to_json(i, c) for i, c in enumerate(cam_infos)], f)
