from simple_knn._C import distCUDA2  # noqa: F401
