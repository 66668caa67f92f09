#!/usr/bin/env python3
"""Randomised check of the front end and of GaussianRenderer.render on the GPU box (dev):
  * distCUDA2 (exact 3-NN mean squared distance) of random point sets -- 1 to 60 000 points, blobs / shells / lines / planes / exact duplicates /
    far outliers / huge offsets, batches of 1-4 sets -- against scipy's cKDTree, and twice (bit for bit);
  * render() of 1-3 subjects x 1-4 views forward + backward twice: image, alpha and every gradient identical bit for bit; a fifth of the scenes with
    NaN / Inf coordinates or attributes sprinkled in: nothing may fault.
usage: python tools/fuzz_render.py [seconds] [seed]"""
import os, sys, time
from types import SimpleNamespace
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigman_release_amd import cameras, synthetic
from sigman_release_amd.renderer import GaussianRenderer, dist_cuda2


def point_set(rng, P):
    kind = rng.choice(["blob", "shell", "line", "plane", "dups", "outliers", "offset", "humanoid", "grid"])
    if kind == "humanoid":
        return synthetic.humanoid(P, int(rng.integers(1, 1 << 30)))["position"], kind
    x = rng.normal(size=(P, 3)) * rng.uniform(0.01, 2.0)
    if kind == "shell":
        x = x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-9) * rng.uniform(0.1, 3.0)
    elif kind == "line":
        x[:, 1:] = 0.0
    elif kind == "plane":
        x[:, 2] = rng.uniform(-1, 1)
    elif kind == "dups":
        x = np.repeat(x[: max(1, P // 4)], 4, 0)[:P]
        if len(x) < P: x = np.concatenate([x, x[: P - len(x)]])
    elif kind == "outliers":
        x[:: max(1, P // 7)] += rng.uniform(5, 50, size=(len(x[:: max(1, P // 7)]), 3))
    elif kind == "offset":
        x = x * 0.3 + rng.uniform(-100, 100, size=(1, 3))
    elif kind == "grid":
        n = max(1, int(round(P ** (1 / 3))))
        g = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float64) * 0.1      # massive exact distance ties
        x = np.concatenate([g, x])[:P]
    return x.astype(np.float32), kind


def run(seconds=60.0, seed=1):
    from scipy.spatial import cKDTree
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(seed)
    t_end = time.time() + float(seconds)
    n_knn = n_render = 0
    while time.time() < t_end:
        # ---- 3-NN
        B = int(rng.choice([1, 1, 2, 4])); P = int(rng.choice([1, 2, 3, 4, 5, 17, 64, 65, 300, 4000, 20000, 60000]))
        sets = [point_set(rng, P) for _ in range(B)]
        pts = np.stack([s[0] for s in sets])
        a = dist_cuda2(torch.from_numpy(pts).to(dev)).cpu().numpy()
        b = dist_cuda2(torch.from_numpy(pts).to(dev)).cpu().numpy()
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), ("knn run to run", P, [s[1] for s in sets])
        for k in range(B):
            p64 = pts[k].astype(np.float64)
            if P >= 4:
                d, _ = cKDTree(p64).query(p64, k=4)
                want = (d[:, 1:4] ** 2).mean(1)
                ok = np.allclose(a[k], want, rtol=3e-5, atol=1e-12 + 1e-6 * float(np.abs(p64).max()) ** 2 * 1e-6)
                assert ok, ("knn vs cKDTree", P, sets[k][1], float(np.abs(a[k] - want).max()), float(want.max()))
            else:
                assert a[k].shape == (P,)          # fewer than three other points: whatever upstream's convention, nothing may fault
        n_knn += 1
        # ---- render()
        S = int(rng.choice([1, 1, 2, 3])); V = int(rng.choice([1, 2, 4])); P = int(rng.choice([1, 2, 3, 4, 50, 2000, 12000]))
        H = int(rng.integers(16, 200)); W = int(rng.integers(16, 200))
        gs = [synthetic.humanoid(P, int(rng.integers(1, 1 << 30))) if rng.random() < 0.6 else synthetic.random_cloud(P, int(rng.integers(1, 1 << 30))) for _ in range(S)]
        base = {k: torch.from_numpy(np.stack([g[k] for g in gs])).to(dev) for k in ("position", "opacity", "scale", "cov3d", "rgb")}
        poisoned = []
        if rng.random() < 0.2:
            for k in base:
                if rng.random() < 0.5:
                    flat = base[k].reshape(-1)
                    idx = torch.from_numpy(rng.integers(0, flat.numel(), size=min(flat.numel(), int(rng.integers(1, 12))))).to(dev)
                    flat[idx] = float(rng.choice([np.nan, np.inf, -np.inf, 1e30]))
                    poisoned.append((k, float(flat[idx[0]])))
        views = [int(v) for v in rng.choice(90, V, replace=False)]
        cams = [cameras.make_cameras(views) for _ in range(S)]
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        cv, cvp, cp = (t(np.stack([c[i] for c in cams])) for i in range(3))
        rend = GaussianRenderer(SimpleNamespace(FoVy=cameras.FOVY, output_size_h=H, output_size_w=W))
        gI = torch.randn(S, V, 3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30))))
        res = []
        for _rep in range(2):
            gd = {k: v.clone().requires_grad_(True) for k, v in base.items()}
            out = rend.render(gd, cv, cvp, cp)
            (torch.nan_to_num(out["image"]) * gI).sum().backward()
            torch.cuda.synchronize()
            res.append([out["image"].detach().cpu().numpy(), out["alpha"].detach().cpu().numpy()] +
                       [(gd[k].grad if gd[k].grad is not None else torch.zeros_like(gd[k])).cpu().numpy() for k in sorted(gd)])
        for nm, x, y in zip(["image", "alpha"] + ["d_" + k for k in sorted(base)], *res):
            if not np.array_equal(x.view(np.uint8), y.view(np.uint8)):
                bad = np.argwhere(~((x == y) | (np.isnan(x) & np.isnan(y))))
                d1 = dist_cuda2(base["position"]).cpu().numpy(); d2 = dist_cuda2(base["position"]).cpu().numpy()
                raise AssertionError(("render run to run", nm, dict(S=S, V=V, P=P, H=H, W=W, seed=seed, scene=n_render, poisoned=poisoned), len(bad), bad[:3].tolist(),
                                      "dist2 differs in", int((d1.view(np.uint32) != d2.view(np.uint32)).sum())))
        n_render += 1
    return n_knn, n_render


if __name__ == "__main__":
    print("fuzz ok: %d point-set batches, %d render scenes" % run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
