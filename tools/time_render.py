"""Dev tool (GPU): what GaussianRenderer.render forward + backward (+ the fused loss) costs, by stage: wall time per step and per-stage HIP-event
times (3-NN, covariance, preprocess, emission, sort, compositing fwd/bwd, gather, loss).   usage: python tools/time_render.py [--subjects 1 --views 1]"""
import argparse, ctypes as C, os, sys, time
from types import SimpleNamespace
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigman_release_amd import _cabi, cameras, synthetic
from sigman_release_amd.losses import clamped_l1_loss
from sigman_release_amd.renderer import GaussianRenderer

ap = argparse.ArgumentParser()
ap.add_argument("--subjects", type=int, default=1); ap.add_argument("--views", type=int, default=1); ap.add_argument("--P", type=int, default=100000)
ap.add_argument("--size", type=int, default=512); ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
dev = torch.device("cuda:0")
B, V, P, H = a.subjects, a.views, a.P, a.size
subj = [synthetic.humanoid(P, 100 + b) for b in range(B)]
gd = {k: torch.from_numpy(np.stack([s[k] for s in subj])).to(dev).requires_grad_(True) for k in ("position", "opacity", "scale", "cov3d", "rgb")}
VIEWS = (30, 37, 45, 53, 65, 85, 0, 8)
cams = [cameras.make_cameras([VIEWS[i % 8] for i in range(V)]) for _ in range(B)]
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
cv, cvp, cp = (t(np.stack([c[i] for c in cams])) for i in range(3))
rend = GaussianRenderer(SimpleNamespace(FoVy=cameras.FOVY, output_size_h=H, output_size_w=H))
gt = torch.rand(B * V, 3, H, H, device=dev)
one = torch.ones((), device=dev)
norm = 1.0 / (B * V * 3 * H * H)
def step():
    for v in gd.values(): v.grad = None
    img = rend.render(gd, cv, cvp, cp)["image"].reshape(B * V, 3, H, H)
    clamped_l1_loss(img, gt, None, norm).backward(one)
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.iters): step()
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / a.iters
L = _cabi.lib()
L.sgr_prof_configure(0xFFFF)
for _ in range(a.iters): step()
torch.cuda.synchronize()
ms = (C.c_double * 16)(); cnt = (C.c_uint32 * 16)(); L.sgr_prof_collect(ms, cnt)
names = {0: "pre_fwd", 1: "scan", 2: "dup", 3: "sort", 4: "ranges", 5: "render_fwd", 6: "render_bwd", 7: "pre_bwd", 8: "knn", 9: "cov3d", 10: "loss"}
st = {names[k]: round(ms[k] / a.iters * 1000, 1) for k in names if cnt[k]}
print(f"render() B={B} V={V} P={P} {H}^2: wall {wall * 1e3:.4f} ms/step; stage us/step (event pairs, each includes its launch gaps): {st}  sum {sum(st.values()):.0f}")
