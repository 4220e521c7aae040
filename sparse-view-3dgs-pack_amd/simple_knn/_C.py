"""`simple_knn._C` stand-in: distCUDA2(points[P,3] float32 cuda) -> float32[P]
(simple-knn/spatial.cu:15-26, ext.cpp), served by libgsplat_hip.so.  No CPU fallback."""
from gsplat_amd._lib import hip_api
from gsplat_amd.knn import dist2


def distCUDA2(points):
    if not points.is_cuda:
        raise RuntimeError("distCUDA2 expects a CUDA(HIP) tensor - there is no CPU path")
    return dist2(hip_api(), points)
