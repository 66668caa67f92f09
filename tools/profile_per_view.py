"""Dev tool: host vs GPU time of the reference's per-view call pattern (gs.py:62-109) -- V views of one 100k-Gaussian subject through the
upstream-signature GaussianRasterizer, clamp, stack, L1, backward.  Prints issue time (host only) and drained time per phase.
env: GRAPHS=0|1|2, V (views, default 8), PROF=1 (cProfile of the forward loop)."""
import cProfile, pstats, os, sys, time
import numpy as np
if os.environ.get("AFF"):
    os.sched_setaffinity(0, set(int(x) for x in os.environ["AFF"].split(",")))
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigman_release_amd import _cabi, cameras, synthetic
from sigman_release_amd import rasterizer as R
from sigman_release_amd.losses import clamped_l1_loss
dev = torch.device("cuda:0")
P, H, V = 100000, 512, int(os.environ.get("V", "8"))
VIEWS = (30, 37, 45, 53, 65, 85, 0, 8)
g = synthetic.humanoid(P, 100); cov = synthetic.covariance_from_gaussians(g)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
cv, cvp, cp = [t(x) for x in cameras.make_cameras([VIEWS[i % 8] for i in range(V)])]
m, c, o, rgb = [t(x).requires_grad_(True) for x in (g["position"], cov, g["opacity"], g["rgb"])]
gt = torch.rand(V, 3, H, H, device=dev)
bg = torch.ones(3, device=dev)
NS = int(os.environ.get("STREAMS", "0"))
streams = [torch.cuda.Stream() for _ in range(NS)]

def forward():
    imgs = []
    cur = torch.cuda.current_stream()
    for s in streams: s.wait_stream(cur)
    for v in range(V):
      with torch.cuda.stream(streams[v % NS]) if NS else torch.cuda.stream(cur):
        rs = R.GaussianRasterizationSettings(image_height=H, image_width=H, tanfovx=cameras.TAN_HALF_FOV, tanfovy=cameras.TAN_HALF_FOV, bg=bg,
                                             scale_modifier=0.5, viewmatrix=cv[v], projmatrix=cvp[v], sh_degree=0, campos=cp[v], prefiltered=False, debug=False)
        rast = R.GaussianRasterizer(raster_settings=rs)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=True):
            img, radii, depth, alpha = rast(means3D=m, means2D=torch.zeros_like(m, dtype=torch.float32, device=dev), shs=None, colors_precomp=rgb, opacities=o,
                                            cov3D_precomp=c)
        imgs.append(img.clamp(0, 1))
    for s in streams: cur.wait_stream(s)
    return clamped_l1_loss(torch.stack(imgs, 0), gt, None, 1e-6)

def step(timing=None):
    for x in (m, c, o, rgb): x.grad = None
    t0 = time.perf_counter(); loss = forward(); t1 = time.perf_counter()
    if timing is not None: torch.cuda.synchronize()
    t2 = time.perf_counter(); loss.backward(); t3 = time.perf_counter()
    if timing is not None:
        torch.cuda.synchronize(); t4 = time.perf_counter()
        timing.append((t1 - t0, t2 - t0, t3 - t2, t4 - t2))

for _ in range(10): step()
torch.cuda.synchronize()
N = 40
t0 = time.perf_counter()
for _ in range(N): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
tm = []
for _ in range(N): step(tm)
a = np.median(np.array(tm), 0) * 1e6 / V
print(f"STREAMS={NS} V={V} GRAPHS={os.environ.get('GRAPHS','default')}: {dt*1e6/V:.0f} us/view pipelined ({V/dt:.0f} views/s) | per view: fwd issue {a[0]:.0f} us, fwd drained {a[1]:.0f} us, bwd issue {a[2]:.0f} us, bwd drained {a[3]:.0f} us", flush=True)
if os.environ.get("PROF"):
    pr = cProfile.Profile(); pr.enable()
    for _ in range(N): forward()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
