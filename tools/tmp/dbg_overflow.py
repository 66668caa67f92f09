import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import cases
from oracle import ref as oracle
from sigman_release_amd import rasterizer as R
dev = torch.device("cuda:0")
inp, st = cases.humanoid(P=5000, H=128, W=128, seed=12)
r = oracle.forward(**inp, **cases.single_view(st))
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
base = R.BatchedRasterizationSettings(128, 128, st["tanfovx"], st["tanfovy"], t(st["bg"]), 1.0, t(st["viewmatrix"]), t(st["projmatrix"]), 0, t(st["campos"]), 1)
for cap in (r.R + 1000, r.R, r.R // 2):
    d = {k: t(v)[None].requires_grad_(True) for k, v in inp.items()}
    try:
        color, radii, depth, alpha = R.rasterize_gaussians_batched(d["means3D"], None, None, d["colors_precomp"], d["opacities"][..., None],
                                                                   None, None, d["cov3D_precomp"], base._replace(max_rendered=cap, debug=True))
        torch.cuda.synchronize()
        print("fwd ok", cap, flush=True)
        color.sum().backward()
        torch.cuda.synchronize()
        print("bwd ok", cap, flush=True)
    except Exception as e:
        print("cap", cap, "->", str(e)[:300], flush=True)
print("done")
