"""MI355X-native Gaussian-splatting rasterizer (drop-in for diff_gaussian_rasterization / simple_knn.distCUDA2).

Importing the package has no side effects on the process (no environment variables, no HIP initialisation); the HIP library is
loaded on first use (sigman_release_amd/_cabi.py)."""
