"""Dev tool (GPU): the host's share of a batched fused step -- the same call sequence as bench.py's C2 step on a scene so small (2 000 Gaussians, 64x64)
that the kernels take less than the host needs to issue them: the step time IS the host time per step (forward on the Python thread + backward on
the autograd thread).   usage: python tools/host_bound.py [cores, e.g. 0-3]"""
import os, sys, time
if len(sys.argv) > 1:
    a, b = sys.argv[1].split("-"); os.sched_setaffinity(0, set(range(int(a), int(b) + 1)))
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigman_release_amd import cameras, synthetic, rasterizer as R
dev = torch.device("cuda:0")
P, H = 2000, 64
g = synthetic.humanoid(P, 1); cov = synthetic.covariance_from_gaussians(g)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
cv, cvp, cp = cameras.make_cameras([30])
st = R.BatchedRasterizationSettings(H, H, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, torch.ones(3, device=dev), 1.0, t(cv), t(cvp), 0, t(cp), 1, False, 60000)
m, c, o, rgb = [t(x)[None].requires_grad_(True) for x in (g["position"], cov, g["opacity"], g["rgb"])]
gt = torch.rand(1, 3, H, H, device=dev); one = torch.ones((), device=dev)
def step():
    for v in (m, c, o, rgb): v.grad = None
    R.rasterize_l1_loss_batched(m, None, None, rgb, o, None, None, c, st, gt, None, 1e-6)[0].backward(one)
res = []
for rep in range(5):
    for _ in range(200): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2000): step()
    torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 2000 * 1e6)
print("host-bound step, us:", [round(x, 1) for x in res], "affinity", sorted(os.sched_getaffinity(0))[:6])
