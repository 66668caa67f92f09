"""Import shim for `from simple_knn._C import distCUDA2` (/root/reference/core/gaussians/gs.py:6)."""
