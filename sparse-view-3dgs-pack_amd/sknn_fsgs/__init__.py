"""`sknn_fsgs` stand-in (FSGS/scene/gaussian_model.py:20,144,406: `from sknn_fsgs import distCUDA2`):
distCUDA2(points[P,3] float32 cuda) -> (float32 [P] mean squared 3-NN distance, int32 [P,3] neighbour indices),
FSGS/submodules/simple-knn/spatial.cu + simple_knn.cu:132-189, served by libgsplat_hip.so.  No CPU fallback."""
from gsplat_amd._lib import hip_api
from gsplat_amd.knn import dist2_with_indices


def distCUDA2(points):
    if not points.is_cuda:
        raise RuntimeError("distCUDA2 expects a CUDA(HIP) tensor - there is no CPU path")
    return dist2_with_indices(hip_api(), points)
