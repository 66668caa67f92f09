#!/usr/bin/env python3
"""For configurations the parity sweep flagged: how many pixels are off, and is there a Gaussian whose alpha sits on the 1/255 threshold
(or a pixel whose transmittance sits on 1e-4) at those pixels?    usage: python tools/fuzz_parity_explain.py seed [seed ...]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from oracle import ref as oracle
from sigman_release_amd import _cabi, cameras
from sigman_release_amd import rasterizer as R
import test_gpu_parity as T
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for seed in map(int, sys.argv[1:]):
    rng = np.random.default_rng(7000 + seed)
    inp, st = T._random_config(seed)
    V = int(rng.choice([1, 1, 2, 3]))
    views = [int(v) for v in rng.choice(90, V, replace=False)]
    st["viewmatrix"], st["projmatrix"], st["campos"] = cameras.make_cameras(views)
    d = {k: t(v)[None] for k, v in inp.items()}
    bst = T._batched_settings(st, dev, V)
    with torch.no_grad():
        color, radii, depth, alpha = R.rasterize_gaussians_batched(d["means3D"], None, d.get("shs"), d.get("colors_precomp"), d["opacities"][..., None],
                                                                   d.get("scales"), d.get("rotations"), d.get("cov3D_precomp"), bst)
    for v in range(V):
        r = oracle.forward(**inp, **cases.single_view(st, v))
        e = np.abs(color[v].cpu().numpy() - r.color).max(0)
        ed = np.abs(depth[v, 0].cpu().numpy() - r.depth[0]); ea = np.abs(alpha[v, 0].cpu().numpy() - r.alpha[0])
        print(f"seed {seed} view {v}: P {r.P} {st['image_height']}x{st['image_width']}: pixels with colour error > 1e-5: {(e > 1e-5).sum()}, > 1e-4: {(e > 1e-4).sum()} "
              f"(max {e.max():.2e}); depth > 1e-4: {(ed > 1e-4).sum()} (max {ed.max():.2e}); alpha > 1e-4: {(ea > 1e-4).sum()} (max {ea.max():.2e}); "
              f"alpha error at the worst pixel / (1/255): {ea.flat[np.argmax(np.maximum(e, ed).ravel())] * 255:.3f}")
