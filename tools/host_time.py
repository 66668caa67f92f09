import os, sys, time
sys.path.insert(0, "/root/repo")
os.sched_setaffinity(0, {0,1,2,3})
import numpy as np, torch
from sigman_release_amd import cameras, synthetic, rasterizer as R
dev = torch.device("cuda:0")
P, H = 100000, 512
g = synthetic.humanoid(P, 1); cov = synthetic.covariance_from_gaussians(g)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
cv, cvp, cp = cameras.make_cameras([30])
st = R.BatchedRasterizationSettings(H, H, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, torch.ones(3, device=dev), 1.0, t(cv), t(cvp), 0, t(cp), 1, False, 260000)
m, c, o, rgb = [t(x)[None].requires_grad_(True) for x in (g["position"], cov, g["opacity"], g["rgb"])]
gt = torch.rand(1, 3, H, H, device=dev)
one = torch.ones((), device=dev)
acc = {"fwd": 0.0, "bwd_total": 0.0, "bwd_impl": 0.0, "fwd_impl": 0.0, "ccall_f": 0.0, "ccall_b": 0.0}
orig_b = R._backward_impl
def timed_b(*a, **k):
    t0 = time.perf_counter(); r = orig_b(*a, **k); acc["bwd_impl"] += time.perf_counter() - t0; return r
R._backward_impl = timed_b
orig_f = R._forward_impl
def timed_f(*a, **k):
    t0 = time.perf_counter(); r = orig_f(*a, **k); acc["fwd_impl"] += time.perf_counter() - t0; return r
R._forward_impl = timed_f
L = R._cabi.lib()
for name, key in (("sgr_rasterize_forward_l1", "ccall_f"), ("sgr_rasterize_backward", "ccall_b")):
    f = getattr(L, name)
    def mk(f, key):
        def w(*a):
            t0 = time.perf_counter(); r = f(*a); acc[key] += time.perf_counter() - t0; return r
        return w
    setattr(L, name, mk(f, key))
def step():
    for v in (m, c, o, rgb): v.grad = None
    t0 = time.perf_counter()
    out = R.rasterize_l1_loss_batched(m, None, None, rgb, o, None, None, c, st, gt, None, 1e-6)[0]
    t1 = time.perf_counter()
    out.backward(one)
    t2 = time.perf_counter()
    acc["fwd"] += t1 - t0; acc["bwd_total"] += t2 - t1
for _ in range(30): step()
torch.cuda.synchronize()
for k in acc: acc[k] = 0.0
N = 300
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter(); torch.cuda.synchronize()
print("issue us/step %.1f" % ((t1 - t0) / N * 1e6), {k: round(v / N * 1e6, 1) for k, v in acc.items()})
