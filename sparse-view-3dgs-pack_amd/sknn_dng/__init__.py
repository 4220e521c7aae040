"""`sknn_dng` stand-in (DNGaussian/scene/gaussian_model.py:20: `from sknn_dng._C import distCUDA2`): the plain
simple-knn (mean squared 3-NN distance, no indices)."""
from ._C import distCUDA2  # noqa: F401

__all__ = ["distCUDA2"]
