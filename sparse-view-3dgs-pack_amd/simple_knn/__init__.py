"""Drop-in for the reference's `simple_knn` package (simple-knn/, imported as
`from simple_knn._C import distCUDA2` at LGDWT-GS/scene/gaussian_model.py:21)."""
