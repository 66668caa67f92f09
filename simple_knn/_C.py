"""`distCUDA2` with upstream's signature, backed by the gfx950 grid-hash 3-NN kernel (sgr_knn_dist2)."""
from sigman_release_amd.renderer import dist_cuda2 as distCUDA2  # noqa: F401
