"""Dev tool (GPU): tile-list length distribution of a bench config -> gpurun_out/tile_hist_<cfg>.json (+ printed class table).
usage: python tools/tile_hist.py c3|c4|c2"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sigman_release_amd import cameras, rasterizer as R

name = sys.argv[1]
cfg = bench.CONFIGS[name]
dev = torch.device("cuda", 0)
P, H, S = cfg["P"], cfg["size"], cfg["subjects"]
views = {"c2": [bench.VIEWS[0]], "c3": None, "c4": list(range(90)), "c5": [bench.VIEWS[0]]}[name]
if views is None:
    views = list(bench.VIEWS)
seeds = {"c2": [1], "c3": [100 + b for b in range(S)], "c4": [3], "c5": [4]}[name]
subs = [bench.build_subject(name, P, s, dev)[0] for s in seeds]
subj = {k: torch.stack([x[k] for x in subs]) for k in ("means3D", "cov3D", "opacity", "rgb")}
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
cv, cvp, cp = cameras.make_cameras(views * S)
st = R.BatchedRasterizationSettings(H, H, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, torch.ones(3, device=dev), 0.5, t(cv), t(cvp), 0, t(cp), len(views))
with torch.no_grad():
    d = R.forward_debug(subj["means3D"], subj["opacity"], colors_precomp=subj["rgb"], cov3D_precomp=subj["cov3D"], settings=st)
rg = d["ranges"].cpu().numpy().astype(np.int64)
n = (rg[..., 1] - rg[..., 0]).reshape(-1)
edges = [0, 1, 65, 129, 257, 513, 1025, 2049, 4097, 8193, 16385, 1 << 30]
print(name, "R", int(d["num_rendered"]), "tiles", n.size, "occupied", int((n > 0).sum()), "max", int(n.max()))
for a, b in zip(edges[:-1], edges[1:]):
    m = (n >= a) & (n < b)
    print(f"  [{a:6d},{b:6d})  tiles {int(m.sum()):7d}  keys {int(n[m].sum()):10d}  ({100.0 * n[m].sum() / max(1, n.sum()):5.1f} %)")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(sorted(int(x) for x in n if x > 0), open(f"gpurun_out/tile_hist_{name}.json", "w"))
