"""MI355X-native Gaussian-splatting rasterizer (drop-in for diff_gaussian_rasterization / simple_knn.distCUDA2)."""
import os as _os
import sys as _sys

# hipGraph replay of the forward chain needs ROCm's graph "packet capture" OFF (see csrc/rasterize.hip: graphs_allowed);
# the HIP runtime reads the variable once when it initialises, so it is only set here while torch -- and with it the HIP
# runtime -- is certainly not loaded yet.  Import this package (or export the variable) before torch to get graph replay;
# otherwise the library silently uses plain launches.
if "torch" not in _sys.modules:
    _os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
