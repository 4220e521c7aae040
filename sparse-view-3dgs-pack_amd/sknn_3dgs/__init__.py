"""Alias package (`sknn_<model>` naming of the pack, simple-knn/sknn_3dgs/__init__.py)."""
from simple_knn._C import distCUDA2  # noqa: F401

__all__ = ["distCUDA2"]
