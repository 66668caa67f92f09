"""All five BASELINE.json configs on one MI355X (evidence for DESIGN.md / README; bench.py stays the driver's contract = C2).
Prints one JSON line per config."""
import json, os, sys, time
from types import SimpleNamespace
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sigman_release_amd import cameras, synthetic
from sigman_release_amd import rasterizer as R
from sigman_release_amd.losses import clamped_l1_loss
from sigman_release_amd.renderer import GaussianRenderer

dev = torch.device("cuda:0")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
VIEWS = (30, 37, 45, 53, 65, 85, 0, 8)


def timeit(fn, steps, warmup):
    for _ in range(warmup): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def raster_case(g, cov, views, H, bwd, da_grads=False):
    cv, cvp, cp = cameras.make_cameras(views)
    st = R.BatchedRasterizationSettings(H, H, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, torch.ones(3, device=dev), 1.0, t(cv), t(cvp), 0, t(cp), len(views))
    m, c, o, rgb = [t(x)[None].requires_grad_(bwd) for x in (g["position"], cov, g["opacity"], g["rgb"])]
    with torch.no_grad():
        probe = R.forward_debug(m.detach(), o.detach(), colors_precomp=rgb.detach(), cov3D_precomp=c.detach(), settings=st)
    Rn, S = probe["num_rendered"], int(probe["n_contrib"].to(torch.int64).sum())
    st = st._replace(max_rendered=int(Rn * 1.25) + 4096)
    gt = torch.rand(len(views), 3, H, H, device=dev)
    def step():
        with torch.set_grad_enabled(bwd):
            for v in (m, c, o, rgb): v.grad = None
            if bwd:
                loss, _, color, radii, depth, alpha = R.rasterize_l1_loss_batched(m, None, None, rgb, o, None, None, c, st, gt, None, 1e-6)
                if da_grads: loss = loss + 1e-6 * (depth.sum() + alpha.sum())
                loss.backward()
            else:
                R.rasterize_gaussians_batched(m, None, None, rgb, o, None, None, c, st)
    return step, Rn, S


out = []
# C1: 10k random Gaussians, 1 view 256^2 -- the CPU-runnable case: CPU oracle next to the GPU
g = synthetic.random_cloud(10_000, 0); cov = synthetic.covariance_from_gaussians(g)
from oracle import ref as oracle
cv, cvp, cp = cameras.make_cameras([30])
kw = dict(viewmatrix=cv[0], projmatrix=cvp[0], campos=cp[0], bg=np.ones(3, np.float32), tanfovx=cameras.TAN_HALF_FOV, tanfovy=cameras.TAN_HALF_FOV, image_height=256, image_width=256)
ts = []
for i in range(4):
    t0 = time.perf_counter(); r = oracle.forward(g["position"], g["opacity"].reshape(-1), colors_precomp=g["rgb"], cov3D_precomp=cov, **kw)
    oracle.backward(r, np.ones((3, 256, 256), np.float32) / 65536); ts.append(time.perf_counter() - t0)
os.sched_setaffinity(0, {0, 1, 2, 3})      # like bench.py: the GPU steps are driven by two host threads; keep them on one CCX
step, Rn, S = raster_case(g, cov, [30], 256, True)
dt = timeit(step, 50, 10)
out.append(dict(config="C1 10k random, 1 view 256x256, fwd+bwd", cpu_oracle_views_per_s=round(1 / np.median(ts[1:]), 2), cpu_cores=os.cpu_count(), gpu_views_per_s=round(1 / dt, 1), num_rendered=Rn, visits=S))
# C2
g = synthetic.humanoid(100_000, 1); cov = synthetic.covariance_from_gaussians(g)
step, Rn, S = raster_case(g, cov, [30], 512, True)
dt = timeit(step, 50, 10)
out.append(dict(config="C2 100k humanoid, 1 view 512x512, fwd+bwd+L1", ms_per_step=round(dt * 1e3, 4), views_per_s=round(1 / dt, 1), visits_per_s=round(S / dt, 0), num_rendered=Rn))
# C3: the VAE render-loss step on ONE GPU: 8 subjects x 8 views through GaussianRenderer.render (3-NN + covariance + raster) + fused loss
B, V, P, H = 8, 8, 100_000, 512
subj = [synthetic.humanoid(P, 100 + b) for b in range(B)]
gauss = {k: torch.from_numpy(np.stack([s[k] for s in subj])).to(dev).requires_grad_(True) for k in ("position", "opacity", "scale", "cov3d", "rgb")}
cams = [cameras.make_cameras(VIEWS) for _ in range(B)]
cam_view = t(np.stack([c[0] for c in cams])); cam_view_proj = t(np.stack([c[1] for c in cams])); cam_pos = t(np.stack([c[2] for c in cams]))
rend = GaussianRenderer(SimpleNamespace(FoVy=cameras.FOVY, output_size_h=H, output_size_w=H))
gt = torch.rand(B * V, 3, H, H, device=dev)
def c3():
    for v in gauss.values(): v.grad = None
    o = rend.render(gauss, cam_view, cam_view_proj, cam_pos)
    clamped_l1_loss(o["image"].view(B * V, 3, H, H), gt, None, 1.0 / (B * V * 3 * H * H)).backward()
dt = timeit(c3, 10, 3)
out.append(dict(config="C3 VAE render-loss step on one GPU: 8 subjects x 8 views 512x512, GaussianRenderer.render (3-NN, covariance, raster) + loss, fwd+bwd", ms_per_step=round(dt * 1e3, 3), views_per_s=round(B * V / dt, 1)))
# C3 again, but driven exactly like the reference's gs.py:62-109: a Python double loop over subjects and views through the
# upstream-signature GaussianRasterizer (one launch chain per view), same 3-NN / covariance / loss ops around it
from sigman_release_amd.renderer import dist_cuda2, covariance_from_scale_rotation
def c3_per_view():
    for v in gauss.values(): v.grad = None
    imgs = []
    for b in range(B):
        with torch.no_grad():
            d2 = dist_cuda2(gauss["position"][b])
        cov_b = covariance_from_scale_rotation(gauss["scale"][b:b + 1], gauss["cov3d"][b:b + 1], d2[None])[0]
        for v in range(V):
            rs = R.GaussianRasterizationSettings(H, H, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, torch.ones(3, device=dev), 0.5,
                                                 cam_view[b, v], cam_view_proj[b, v], 0, cam_pos[b, v], False, False)
            img, radii, depth, alpha = R.GaussianRasterizer(rs)(means3D=gauss["position"][b], means2D=torch.zeros_like(gauss["position"][b]),
                                                                opacities=gauss["opacity"][b], colors_precomp=gauss["rgb"][b], cov3D_precomp=cov_b)
            imgs.append(img.clamp(0, 1))
    clamped_l1_loss(torch.stack(imgs), gt, None, 1.0 / (B * V * 3 * H * H)).backward()
dt = timeit(c3_per_view, 6, 2)
out.append(dict(config="C3 through the reference's own call pattern (gs.py:62-109): Python loop over 8 subjects x 8 views, upstream-signature GaussianRasterizer per view, fwd+bwd", ms_per_step=round(dt * 1e3, 3), views_per_s=round(B * V / dt, 1)))
# C4: decode path, 200k Gaussians, 90-view orbit at 1024^2, forward only
g = synthetic.humanoid(200_000, 3); cov = synthetic.covariance_from_gaussians(g)
step, Rn, S = raster_case(g, cov, list(range(90)), 1024, False)
dt = timeit(step, 5, 2)
out.append(dict(config="C4 200k humanoid, 90 views 1024x1024, forward only", ms_per_step=round(dt * 1e3, 3), views_per_s=round(90 / dt, 1), visits_per_s=round(S / dt, 0), num_rendered=Rn))
# C5: 1M stress, depth + alpha gradients on
g = synthetic.humanoid_layers(1_000_000, 4, layers=10); cov = synthetic.covariance_from_gaussians(g)
step, Rn, S = raster_case(g, cov, [30], 512, True, da_grads=True)
dt = timeit(step, 20, 5)
out.append(dict(config="C5 1M Gaussians (10 jittered layers), 1 view 512x512, depth+alpha grads on, fwd+bwd", ms_per_step=round(dt * 1e3, 4), views_per_s=round(1 / dt, 1), visits_per_s=round(S / dt, 0), num_rendered=Rn,
                algorithmic_MB_per_view=round((212 * 1e6 + 176 * Rn + 52 * 512 * 512 + 8 * 1024) / 1e6, 1),
                hbm_roofline_pct=round(100 * ((212 * 1e6 + 176 * Rn + 52 * 512 * 512 + 8 * 1024) / dt) / 8e12, 3)))
for o in out: print(json.dumps(o))
