#!/usr/bin/env python3
"""Randomised check of the upstream-signature single-view op on the GPU box (dev) -- the reference's own call pattern (gs.py:62-109): a long random
SEQUENCE of GaussianRasterizer calls that revisits a handful of shapes with subjects of very different instance counts, so that the automatic
capacity is learned, overflows and is re-learned (the inline check re-renders an overflowing forward exactly before anything is returned); several
forwards before one backward (one count slot per pending forward); colours + covariances (C++ node) or SH + scales / rotations (Python node).
Every view is compared, bit for bit, with the same view rendered by the batched op in exact mode: image, radii, depth, alpha, all gradients.
usage: python tools/fuzz_per_view.py [seconds] [seed]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigman_release_amd import cameras, synthetic
from sigman_release_amd import rasterizer as R


def run(seconds=60.0, seed=1):
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(seed)
    t_end = time.time() + float(seconds)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    shapes = [(int(rng.choice([50, 1000, 6000])), int(rng.integers(16, 260)), int(rng.integers(16, 260))) for _ in range(4)]
    n = 0
    while time.time() < t_end:
        P, H, W = shapes[int(rng.integers(len(shapes)))]
        sh = rng.random() < 0.25
        g = synthetic.humanoid(P, int(rng.integers(1, 1 << 30))) if rng.random() < 0.5 else synthetic.random_cloud(P, int(rng.integers(1, 1 << 30)))
        scale = float(rng.choice([0.05, 1.0, 1.0, 4.0, 40.0]))               # instance counts over three orders of magnitude for ONE shape
        means, op = t(g["position"]), t(g["opacity"].reshape(P, 1))
        if sh:
            deg = int(rng.integers(0, 4)); M = (deg + 1) ** 2
            q = rng.normal(size=(P, 4)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
            base = dict(means3D=means, opacities=op, shs=t((rng.normal(size=(P, M, 3)) * 0.3).astype(np.float32)),
                        scales=t((rng.uniform(0.005, 0.05, (P, 3)) * np.sqrt(scale)).astype(np.float32)), rotations=t(q))
        else:
            deg = 0
            base = dict(means3D=means, opacities=op, colors_precomp=t(g["rgb"]), cov3D_precomp=t((synthetic.covariance_from_gaussians(g) * scale).astype(np.float32)))
        V = int(rng.choice([1, 2, 4]))
        views = [int(v) for v in rng.choice(90, V, replace=False)]
        cv, cvp, cp = cameras.make_cameras(views)
        bg = torch.tensor(rng.uniform(0, 1, 3).astype(np.float32), device=dev)
        smod = float(rng.uniform(0.5, 1.5))
        gC = torch.randn(V, 3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30))))
        # ---- the reference's loop: V forwards, then one backward
        d = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        outs = []
        for i in range(V):
            rs = R.GaussianRasterizationSettings(H, W, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, bg, smod, t(cv[i]), t(cvp[i]), deg, t(cp[i]), False, False)
            outs.append(R.GaussianRasterizer(rs)(means3D=d["means3D"], means2D=torch.zeros_like(d["means3D"]), opacities=d["opacities"], shs=d.get("shs"),
                                                 colors_precomp=d.get("colors_precomp"), scales=d.get("scales"), rotations=d.get("rotations"),
                                                 cov3D_precomp=d.get("cov3D_precomp")))
        sum((o[0] * gC[i]).sum() for i, o in enumerate(outs)).backward()
        torch.cuda.synchronize()
        R.check_pending_overflows(True)
        # ---- the same views through the batched op, exact mode, one view at a time (gradients summed in view order like autograd does: reversed)
        for i in range(V):
            bst = R.BatchedRasterizationSettings(H, W, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, bg, smod, t(cv[i:i + 1]), t(cvp[i:i + 1]), deg, t(cp[i:i + 1]), 1, False, 0)
            e = {k: v.clone()[None].requires_grad_(True) for k, v in base.items()}
            ref = R.rasterize_gaussians_batched(e["means3D"], None, e.get("shs"), e.get("colors_precomp"), e["opacities"], e.get("scales"), e.get("rotations"),
                                                e.get("cov3D_precomp"), bst)
            for nm, a, b in zip(("color", "radii", "depth", "alpha"), outs[i], ref):
                a, b = a.detach().cpu().numpy(), b.detach()[0].cpu().numpy()
                if not np.array_equal(a.view(np.uint8), b.reshape(a.shape).view(np.uint8)):
                    raise AssertionError(("per-view vs batched", nm, dict(P=P, H=H, W=W, V=V, view=i, sh=sh, scale=scale, scene=n, seed=seed), int((a != b.reshape(a.shape)).sum())))
        n += 1
    return n


if __name__ == "__main__":
    print("fuzz ok:", run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1), "sequences of per-view calls")
