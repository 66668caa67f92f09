"""Dev tool: per-kernel HIP-event timings of forward / forward+backward on the C2 inputs (or --views N batched).
Usage: python tools/time_kernels.py [--views 1] [--iters 30] [--bwd]   (SIGMAN_GSPLAT_LIB=/path/to/variant.so to A/B builds)"""
import argparse, ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigman_release_amd import _cabi, cameras, synthetic
from sigman_release_amd import rasterizer as R

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=1); ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--bwd", action="store_true"); ap.add_argument("--P", type=int, default=100000); ap.add_argument("--size", type=int, default=512)
ap.add_argument("--layers", type=int, default=0); ap.add_argument("--subjects", type=int, default=1); ap.add_argument("--fwd-mode", type=int, default=0); ap.add_argument("--sort-mode", type=int, default=-1)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = synthetic.humanoid(a.P, 1) if not a.layers else synthetic.humanoid_layers(a.P, 4, a.layers)
cov = synthetic.covariance_from_gaussians(g)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
V = [(30, 37, 45, 53, 65, 85, 0, 8)[i % 8] for i in range(a.views)] * a.subjects
cv, cvp, cp = cameras.make_cameras(V)
st = R.BatchedRasterizationSettings(a.size, a.size, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, torch.ones(3, device=dev), 1.0, t(cv), t(cvp), 0, t(cp), a.views)
m, c, o, rgb = [t(x)[None].repeat(a.subjects, *([1] * x.ndim)).contiguous().requires_grad_(a.bwd) for x in (g["position"], cov, g["opacity"], g["rgb"])]
L = _cabi.lib()
L.sgr_set_forward_mode(a.fwd_mode)
if a.sort_mode >= 0: L.sgr_set_sort_mode(a.sort_mode)
names = {0: "pre_fwd", 1: "scan", 2: "dup", 3: "sort", 4: "ranges", 5: "render_fwd", 6: "render_bwd", 7: "pre_bwd"}
gc = torch.randn(len(V), 3, a.size, a.size, device=dev) / (a.size * a.size)
def step():
    with torch.set_grad_enabled(a.bwd):
        color, radii, depth, alpha = R.rasterize_gaussians_batched(m, None, None, rgb, o, None, None, c, st)
        if a.bwd:
            color.backward(gc)
for _ in range(5): step()
torch.cuda.synchronize(); L.sgr_prof_configure(0xFFFF)
for _ in range(a.iters): step()
torch.cuda.synchronize()
ms = (C.c_double * 16)(); cnt = (C.c_uint32 * 16)(); L.sgr_prof_collect(ms, cnt)
print(os.environ.get("SIGMAN_GSPLAT_LIB", "default"), {names[k]: round(ms[k] / a.iters * 1000, 1) for k in names if cnt[k]}, "us/step")
