"""Alias package: the pack's renamed rasterizer module (`dgr_<model>` convention,
/root/reference/fs3dgs_benchmark/readme.md:130-190) -> same objects as diff_gaussian_rasterization."""
from diff_gaussian_rasterization import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                         _RasterizeGaussians, rasterize_gaussians, _C)
