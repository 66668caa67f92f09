#!/usr/bin/env python3
"""Randomised cross-check of the sort flavours on the GPU box (dev): random humanoid scenes (1-2 views, 256..512 px, 2k..130k Gaussians, exact and
sync-free mode) through forward_debug with the automatic flavour and with the three-kernel whole-key passes: sorted keys, point list, tile ranges
and images must be identical bit for bit.     usage: [FUZZ_VIEWS=3,5,8] python tools/fuzz_bin_modes.py [seconds]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from sigman_release_amd import _cabi, cameras, synthetic
from sigman_release_amd import rasterizer as R

dev = torch.device("cuda", 0)
L = _cabi.lib()
rng = np.random.default_rng(20260929)
t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
n = 0
while time.time() < t_end:
    P = int(rng.choice([2000, 9000, 33000, 70001, 100000, 130000]))
    H = int(rng.choice([256, 272, 400, 512])); W = int(rng.choice([256, 304, 512]))
    nv = int(rng.choice([int(x) for x in os.environ.get("FUZZ_VIEWS", "1,1,2").split(",")]))
    views = [int(v) for v in rng.choice(90, nv, replace=False)]
    g = synthetic.humanoid(P, int(rng.integers(1, 1 << 30)))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    m, o, c, cov = t(g["position"])[None], t(g["opacity"].reshape(P))[None], t(g["rgb"])[None], t(synthetic.covariance_from_gaussians(g))[None]
    cv, cvp, cp = cameras.make_cameras(views)
    st = R.BatchedRasterizationSettings(H, W, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, torch.ones(3, device=dev), 1.0, t(cv), t(cvp), 0, t(cp), nv)
    ref = None
    for mode in (1, 3):
        for cap in (0, None):
            L.sgr_set_sort_mode(mode)
            try:
                s2 = st
                if cap is None:
                    if ref is None: continue
                    s2 = st._replace(max_rendered=int(ref["num_rendered"] * 1.1) + 100)
                out = R.forward_debug(m, o, colors_precomp=c, cov3D_precomp=cov, settings=s2)
                torch.cuda.synchronize()
                got = {k: out[k].detach().cpu().numpy().copy() for k in ("keys", "point_list", "ranges", "color", "n_contrib")}
                got["num_rendered"] = out["num_rendered"]
                if ref is None: ref = got; continue
                for k in ("keys", "point_list", "color", "n_contrib"):
                    a, b = got[k], ref[k]
                    assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), (k, P, H, W, views, mode, cap)
                occ = ref["ranges"][..., 1] > ref["ranges"][..., 0]
                assert np.array_equal(got["ranges"][occ], ref["ranges"][occ]), ("ranges", P, H, W, views, mode, cap)
            finally:
                L.sgr_set_sort_mode(3)
    n += 1
print("fuzz ok:", n, "scenes")
