#!/usr/bin/env python3
"""For a configuration whose GRADIENT the parity sweep flagged (tools/fuzz_parity.py: 'gradient (seed, tensor, ...)'): who is right?  The fp32 C
oracle, the HIP path (if a GPU is there) and the independent fp64 dense autograd oracle on the same inputs, per tensor: largest difference
relative to the tensor's largest entry, and the Gaussian it belongs to.      usage: python tools/fuzz_grad_explain.py seed [seed ...]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from oracle import ref as oracle, dense_oracle
from sigman_release_amd import cameras
import test_gpu_parity as T

names = {"means3D": "means3D", "opacities": "opacities", "colors_precomp": "colors_precomp", "shs": "sh", "cov3D_precomp": "cov3D_precomp",
         "scales": "scales", "rotations": "rotations"}
for seed in map(int, sys.argv[1:]):
    rng = np.random.default_rng(7000 + seed)
    inp, st = T._random_config(seed)
    V = int(rng.choice([1, 1, 2, 3]))
    views = [int(v) for v in rng.choice(90, V, replace=False)]
    st["viewmatrix"], st["projmatrix"], st["campos"] = cameras.make_cameras(views)
    H, W = st["image_height"], st["image_width"]
    g = [cases.grads_for(H, W, seed=seed * 7 + v) for v in range(V)]
    acc32 = acc64 = None
    margin = 1e9
    for v in range(V):
        sv = cases.single_view(st, v)
        r = oracle.forward(**inp, **sv)
        gr = oracle.backward(r, *g[v])
        acc32 = gr if acc32 is None else {k: acc32[k] + gr[k] for k in acc32}
        tin = {k: torch.tensor(x, dtype=torch.float64, requires_grad=True) for k, x in inp.items()}
        out = dense_oracle.render(**tin, **sv)
        margin = min(margin, float(out["margin"]))
        ((out["color"] * torch.tensor(g[v][0], dtype=torch.float64)).sum() + (out["depth"] * torch.tensor(g[v][1], dtype=torch.float64)).sum()
         + (out["alpha"] * torch.tensor(g[v][2], dtype=torch.float64)).sum()).backward()
        g64 = {k: tin[k].grad.numpy() for k in tin}
        acc64 = g64 if acc64 is None else {k: acc64[k] + g64[k] for k in acc64}
    hip = None
    if torch.cuda.is_available():
        from sigman_release_amd import rasterizer as R
        dev = torch.device("cuda", 0)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        d = {k: t(x)[None].requires_grad_(True) for k, x in inp.items()}
        bst = T._batched_settings(st, dev, V)
        color, radii, depth, alpha = R.rasterize_gaussians_batched(d["means3D"], None, d.get("shs"), d.get("colors_precomp"), d["opacities"][..., None],
                                                                   d.get("scales"), d.get("rotations"), d.get("cov3D_precomp"), bst)
        sum((color[v] * t(g[v][0])).sum() + (depth[v] * t(g[v][1])).sum() + (alpha[v] * t(g[v][2])).sum() for v in range(V)).backward()
        hip = {k: d[k].grad[0].cpu().numpy() for k in d}
    print(f"seed {seed}: P {inp['means3D'].shape[0]} {H}x{W}, {V} view(s); smallest distance of a discrete decision from its threshold (fp64): {margin:.2e}")
    for k in inp:
        want = acc64[k].reshape(-1)
        c32 = acc32[names[k]].reshape(-1).astype(np.float64)
        sc = max(np.abs(want).max(), 1e-30)
        line = f"  {k:14s} max|g| {sc:.3e}:  C oracle (fp32) vs fp64 autograd {np.abs(c32 - want).max() / sc:.3e}"
        if hip is not None:
            h = hip[k].reshape(-1).astype(np.float64)
            i = int(np.argmax(np.abs(h - c32)))
            line += f";  HIP vs fp64 {np.abs(h - want).max() / sc:.3e};  HIP vs C oracle {np.abs(h - c32).max() / sc:.3e} (entry {i}: HIP {h[i]:.6e}, C {c32[i]:.6e}, fp64 {want[i]:.6e})"
        print(line)
