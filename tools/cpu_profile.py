"""Dev tool: where does the HOST time of one fwd+bwd step go? (cProfile over N steps, sync-free mode)"""
import cProfile, pstats, os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import numpy as np
if os.environ.get("AFF"):
    os.sched_setaffinity(0, set(int(x) for x in os.environ["AFF"].split(",")))
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigman_release_amd import cameras, synthetic
from sigman_release_amd import rasterizer as R
from sigman_release_amd.losses import clamped_l1_loss
dev = torch.device("cuda:0")
P, H = 100000, 512
g = synthetic.humanoid(P, 1); cov = synthetic.covariance_from_gaussians(g)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
cv, cvp, cp = cameras.make_cameras([30])
st = R.BatchedRasterizationSettings(H, H, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, torch.ones(3, device=dev), 1.0, t(cv), t(cvp), 0, t(cp), 1, False, int(os.environ.get("CAP", "260000")))
m, c, o, rgb = [t(x)[None].requires_grad_(True) for x in (g["position"], cov, g["opacity"], g["rgb"])]
gt = torch.rand(1, 3, H, H, device=dev)
from sigman_release_amd import _cabi as _c
_c.lib().sgr_set_graphs(int(os.environ.get("GRAPHS", "1")))
FUSED = os.environ.get("FUSED", "0") == "1"
def step():
    for v in (m, c, o, rgb): v.grad = None
    if FUSED:
        R.rasterize_l1_loss_batched(m, None, None, rgb, o, None, None, c, st, gt, None, 1e-6)[0].backward()
        return
    color, radii, depth, alpha = R.rasterize_gaussians_batched(m, None, None, rgb, o, None, None, c, st)
    clamped_l1_loss(color, gt, None, 1e-6).backward()
for _ in range(20): step()
torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host-side {1e6*(t1-t0)/N:.0f} us/step issue time; {1e6*(t2-t0)/N:.0f} us/step incl. drain")
import ctypes as C
h, m_ = C.c_uint64(0), C.c_uint64(0)
_c.lib().sgr_graph_stats(C.byref(h), C.byref(m_)); print("graph hits", h.value, "misses", m_.value, flush=True)
if os.environ.get("NOPROF"): sys.exit(0)
pr = cProfile.Profile(); pr.enable()
for _ in range(N): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats(os.environ.get("SORT", "cumulative")).print_stats(int(os.environ.get("LINES", "28")))
import ctypes as C
from sigman_release_amd import _cabi
h, m = C.c_uint64(0), C.c_uint64(0)
_cabi.lib().sgr_graph_stats(C.byref(h), C.byref(m)); print("graph hits", h.value, "misses", m.value)
