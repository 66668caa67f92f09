"""Dev tool (GPU): what the HOST needs per step of bench.py's headline call (rasterize_l1_loss_batched through the C++ node + loss.backward()): the
same call on a scene so small that the GPU is never the limit (200 splats, 32 x 32), wall time per step over 2 000 steps -- the level below which the C2
step (0.131 ms of kernels) stays GPU-bound.   usage: python tools/host_floor.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigman_release_amd import cameras, synthetic, rasterizer as R
dev = torch.device("cuda:0")
P, H = 200, 32
g = synthetic.humanoid(P, 1); cov = synthetic.covariance_from_gaussians(g)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
cv, cvp, cp = cameras.make_cameras([30])
st = R.BatchedRasterizationSettings(H, H, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, torch.ones(3, device=dev), 1.0, t(cv), t(cvp), 0, t(cp), 1, False, 20000)
m, c, o, rgb = [t(x)[None].requires_grad_(True) for x in (g["position"], cov, g["opacity"].reshape(P, 1), g["rgb"])]
gt = torch.rand(1, 3, H, H, device=dev); mask = (torch.rand(1, 1, H, H, device=dev) > 0.5).float()
def step():
    for x in (m, c, o, rgb): x.grad = None
    R.rasterize_l1_loss_batched(m, None, None, rgb, o, None, None, c, st, gt, mask, 1e-3)[0].backward()
for fused in (1, 0, 1, 0):
    R._cabi.lib().sgr_set_fused_step(fused)
    for _ in range(200): step()
    torch.cuda.synchronize()
    res = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(2000): step()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 2000 * 1e6)
    print(f"fused step {fused}: {min(res):.1f} .. {max(res):.1f} us per step (tiny scene: host / launch bound)")
