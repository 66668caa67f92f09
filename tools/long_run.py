"""Dev tool: the C2 step (fused rasterizer + clamp/L1 node, sync-free mode) for many thousand steps, in blocks: wall time per step with the GPU
drained at block ends, the host's time to ISSUE a step (no sync inside a block), the GPU's own time per step (events around a block), plus
the process's RSS and the collector's counters -- tells a host that falls behind from a GPU that slows down.
usage: python tools/long_run.py [blocks] [steps per block] [pin: 1|0]"""
import gc, os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sigman_release_amd import cameras, synthetic, rasterizer as R
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 20
per = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
if len(sys.argv) > 3 and sys.argv[3] == "1": os.sched_setaffinity(0, {0, 1, 2, 3})
dev = torch.device("cuda:0")
P, H = int(os.environ.get('LR_P', 100000)), int(os.environ.get('LR_H', 512))
g = synthetic.humanoid(P, 1); cov = synthetic.covariance_from_gaussians(g)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
cv, cvp, cp = cameras.make_cameras([30])
st = R.BatchedRasterizationSettings(H, H, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, torch.ones(3, device=dev), 1.0, t(cv), t(cvp), 0, t(cp), 1, False, int(os.environ.get('LR_CAP', 260000)))
m, c, o, rgb = [t(x)[None].requires_grad_(True) for x in (g["position"], cov, g["opacity"], g["rgb"])]
gt = torch.rand(1, 3, H, H, device=dev)
one = torch.ones((), device=dev)
def step():
    for v in (m, c, o, rgb): v.grad = None
    R.rasterize_l1_loss_batched(m, None, None, rgb, o, None, None, c, st, gt, None, 1e-6)[0].backward(one)
for _ in range(50): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
A = torch.randn(4096, 4096, device=dev)
def probes():
    """(GPU clock probe: one 4096^3 fp32 matmul, event-timed; CPU clock probe: a fixed pure-Python loop)"""
    torch.cuda.synchronize(); e0.record(); B = A @ A; e1.record(); torch.cuda.synchronize(); g = e0.elapsed_time(e1)
    t0 = time.perf_counter(); x = 0
    for i in range(200000): x += i & 3
    return g, (time.perf_counter() - t0) * 1e3
for b in range(blocks):
    pg, pc = probes() if os.environ.get("LR_PROBES", "1") == "1" else (0.0, 0.0)
    t0 = time.perf_counter(); e0.record()
    for _ in range(per): step()
    t1 = time.perf_counter(); e1.record(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("block %3d: wall %.1f us/step, host issue %.1f us/step, gpu %.1f us/step, rss %d MB, gc %s, probes: matmul %.3f ms, python loop %.2f ms" % (
        b, (t2 - t0) / per * 1e6, (t1 - t0) / per * 1e6, e0.elapsed_time(e1) / per * 1e3, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024, gc.get_count(), pg, pc), flush=True)
