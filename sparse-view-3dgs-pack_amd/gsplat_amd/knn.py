"""distCUDA2 host glue (simple-knn/spatial.cu:15-26) over gs_knn_mean_dist2."""
import ctypes as C

import torch


def dist2(api, points):
    """points: float32 [P,3] on the backend's device -> float32 [P] mean squared 3-NN distance."""
    if points.ndim != 2 or points.shape[1] != 3:
        raise RuntimeError("points must have dimensions (num_points, 3)")
    P = int(points.shape[0])
    pts = points.contiguous().float()
    means = torch.zeros((P,), dtype=torch.float32, device=pts.device)
    if P == 0:
        return means
    nbytes = int(api.raw("knn_tmp_bytes")(P))
    tmp = torch.empty((nbytes,), dtype=torch.uint8, device=pts.device)
    stream = C.c_void_p(torch.cuda.current_stream(pts.device).cuda_stream) if pts.is_cuda else None
    api.call("knn_mean_dist2", pts.data_ptr(), P, means.data_ptr(), tmp.data_ptr(), nbytes, stream)
    return means


def dist2_with_indices(api, points):
    """FSGS's distCUDA2 (FSGS/submodules/simple-knn/spatial.cu): -> (float32 [P], int32 [P,3] nearest-first)."""
    if points.ndim != 2 or points.shape[1] != 3:
        raise RuntimeError("points must have dimensions (num_points, 3)")
    P = int(points.shape[0])
    pts = points.contiguous().float()
    means = torch.zeros((P,), dtype=torch.float32, device=pts.device)
    nearest = torch.zeros((P, 3), dtype=torch.int32, device=pts.device)
    if P == 0:
        return means, nearest
    nbytes = int(api.raw("knn_tmp_bytes")(P))
    tmp = torch.empty((nbytes,), dtype=torch.uint8, device=pts.device)
    stream = C.c_void_p(torch.cuda.current_stream(pts.device).cuda_stream) if pts.is_cuda else None
    api.call("knn_mean_dist2_idx", pts.data_ptr(), P, means.data_ptr(), nearest.data_ptr(), tmp.data_ptr(), nbytes, stream)
    return means, nearest
