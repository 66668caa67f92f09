"""Dev tool: time dist_cuda2 (3-NN) on a 100k-point humanoid and a 100k volumetric cloud."""
import sys, time, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigman_release_amd import synthetic
from sigman_release_amd.renderer import dist_cuda2
which = sys.argv[1] if len(sys.argv) > 1 else ""
for name, pts in (("humanoid surface", synthetic.humanoid(100000, 100)["position"]), ("uniform volume", np.random.default_rng(0).uniform(-1, 1, (100000, 3)).astype(np.float32))):
    if which and which not in name: continue
    p = torch.from_numpy(pts).cuda()
    for _ in range(5): d = dist_cuda2(p)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): d = dist_cuda2(p)
    torch.cuda.synchronize(); print(name, "knn us/call", round((time.perf_counter() - t) / 50 * 1e6, 1))
