"""Import shim: lets `/root/reference/core/gaussians/gs.py:8-11` run unchanged on MI355X.

    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

resolves to the gfx950 HIP implementation in sigman_release_amd (no CUDA package involved)."""
from sigman_release_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, _RasterizeGaussians,  # noqa: F401
                                           mark_visible, rasterize_gaussians)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "mark_visible"]
