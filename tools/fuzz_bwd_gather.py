#!/usr/bin/env python3
"""Randomised cross-check of the two backward gathers on the GPU box (dev): random multi-subject, multi-view batches, forward + backward with
the lanes kernel and with the view-loop kernel (sgr_set_backward_gather 0 / 1), twice each: all gradients identical bit for bit, run to run
and kernel to kernel.     usage: python tools/fuzz_bwd_gather.py [seconds]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from sigman_release_amd import _cabi, cameras, synthetic
from sigman_release_amd import rasterizer as R

dev = torch.device("cuda", 0)
L = _cabi.lib()
rng = np.random.default_rng(77)
t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
n = 0
try:
    while time.time() < t_end:
        S = int(rng.choice([1, 2, 3])); V = int(rng.choice([2, 4, 8])); P = int(rng.choice([3000, 12000, 40000]))
        H = int(rng.choice([128, 256, 304])); W = int(rng.choice([128, 272, 256]))
        views = [int(v) for v in rng.choice(90, V, replace=False)]
        hosts = [synthetic.humanoid(P, int(rng.integers(1, 1 << 30))) for _ in range(S)]
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        base = [torch.stack([t(g["position"]) for g in hosts]), torch.stack([t(g["opacity"].reshape(P, 1)) for g in hosts]),
                torch.stack([t(g["rgb"]) for g in hosts]), torch.stack([t(synthetic.covariance_from_gaussians(g)) for g in hosts])]
        cv, cvp, cp = cameras.make_cameras(views * S)
        st = R.BatchedRasterizationSettings(H, W, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, torch.ones(3, device=dev), 1.0, t(cv), t(cvp), 0, t(cp), V)
        gC = torch.randn(S * V, 3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30))))
        ref = None
        for mode in (0, 1, 0, 1):
            L.sgr_set_backward_gather(mode)
            leaves = [x.clone().requires_grad_(True) for x in base]
            color = R.rasterize_gaussians_batched(leaves[0], None, None, leaves[2], leaves[1], None, None, leaves[3], st)[0]
            (color * gC).sum().backward()
            torch.cuda.synchronize()
            got = [x.grad.detach().cpu().numpy().copy() for x in leaves]
            if ref is None: ref = got; continue
            for a, b, nm in zip(got, ref, ("means3D", "opacity", "rgb", "cov3D")):
                assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), (nm, S, V, P, H, W, mode)
        n += 1
finally:
    L.sgr_set_backward_gather(0)
print("fuzz ok:", n, "batches")
