"""Dev tool: phase stamps of the single-view path's per-tile sort (deep_tile_kernel<.., FB>), one forward of C2 / C1.
usage (GPU box): tools/build_ab.sh stamps tile_sort.hip -DSGR_DEEP_TIMING && SIGMAN_GSPLAT_LIB=$PWD/tools/ab/stamps.so SIGMAN_PY_NODE=1 python tools/deep_stamps.py [c1|c2|c5]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sigman_release_amd import _cabi, cameras, rasterizer as R

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
dev = torch.device("cuda:0")
c = bench.CONFIGS[cfg]
P, H = c["P"], c["size"]
sub, _, _ = bench.build_subject(cfg, P, {"c1": 0, "c2": 1, "c5": 4}[cfg], dev, os.environ.get("ORDER", "random"))
cv, cvp, cp = cameras.make_cameras([bench.VIEWS[0]])
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
st = R.BatchedRasterizationSettings(H, H, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, torch.ones(3, device=dev), 0.5, t(cv), t(cvp), 0, t(cp), 1)
with torch.no_grad():
    probe = R.forward_debug(sub["means3D"][None], sub["opacity"][None], colors_precomp=sub["rgb"][None], cov3D_precomp=sub["cov3D"][None], settings=st)
st = st._replace(max_rendered=int(probe["num_rendered"] * 1.25) + 4096)
leaves = [sub[k][None].clone().requires_grad_(True) for k in ("means3D", "rgb", "opacity", "cov3D")]
fused = os.environ.get("FUSED", "0") == "1"
tgt = torch.rand(1, 3, H, H, device=dev)
for _ in range(5):
    if fused:
        R.FUSE_STEP_IN_PYTHON_NODE = True
        out = R.rasterize_l1_loss_batched(leaves[0], None, None, leaves[1], leaves[2], None, None, leaves[3], st, tgt, None, 1e-6)
    else:
        out = R.rasterize_gaussians_batched(leaves[0], None, None, leaves[1], leaves[2], None, None, leaves[3], st)
torch.cuda.synchronize()
L = ctypes.CDLL(_cabi.LIB_PATH)
buf = np.zeros(1024 * 16, dtype=np.uint64)
assert L.sgr_debug_deep_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
s = buf.reshape(1024, 16).astype(np.int64)
t0 = s[:, 0][s[:, 0] > 0].min()
rows = [(int(s[i, 15]), i) for i in range(1024) if s[i, 9] > 0]
rows.sort(reverse=True)
names = {0: "start", 8: "select", 9: "columns", 10: "gather", 1: "range", 2: "coarse", 3: "alloc", 4: "fine", 5: "scan", 6: "place", 7: "rank/out"}
print("num_rendered", probe["num_rendered"], "workgroups with work:", len(rows), "  units: us since the first start")
for n, i in rows[:4] + [rows[len(rows) * k // 16] for k in (1, 2, 3, 4, 6, 8, 10, 12)] + rows[-3:]:
    print(f"wg {i:4d} n={n:5d}: " + "  ".join(f"{names[k]} {(s[i, k] - t0) / 100.0:6.2f}" for k in (0, 8, 9, 10, 1, 2, 3, 4, 5, 6, 7) if s[i, k] > 0))
empt = [i for i in range(1024) if s[i, 0] > 0 and s[i, 9] == 0]
if empt:
    print("other workgroups:", len(empt), "start min/median/max", (np.min(s[empt, 0]) - t0) / 100.0, (np.median(s[empt, 0]) - t0) / 100.0, (np.max(s[empt, 0]) - t0) / 100.0,
          " select:", (np.max(s[empt, 8]) - t0) / 100.0)
ends = np.array([(s[i, 7] - t0) / 100.0 for _, i in rows])
starts = np.array([(s[i, 0] - t0) / 100.0 for _, i in rows])
print("working workgroups: start min/median/max", starts.min(), np.median(starts), starts.max(), " end min/median/max", ends.min(), np.median(ends), ends.max())
late = sorted(((s[i, 7] - t0) / 100.0, i, n) for n, i in rows)[-4:]
for e, i, n in late:
    print(f"  late wg {i:4d} n={n:5d}: " + "  ".join(f"{names[k]} {(s[i, k] - t0) / 100.0:6.2f}" for k in (0, 8, 9, 10, 1, 5, 6, 7) if s[i, k] > 0))
print("last stamp of any workgroup:", (s[:, :11].max() - t0) / 100.0, " fused" if fused else " unfused")
