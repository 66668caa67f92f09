"""Dev tool: the fused single-view step of a BASELINE configuration for many thousand steps on constant inputs -- the loss and every gradient tensor
must come out bit for bit the same at every check (the path has no float atomics and no order-dependent sums: any difference is a race).
usage (GPU box): python tools/soak.py [c1|c2|c5] [steps] [check every]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from sigman_release_amd import cameras, rasterizer as R

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
every = int(sys.argv[3]) if len(sys.argv) > 3 else 500
dev = torch.device("cuda:0")
c = bench.CONFIGS[cfg]
P, H = c["P"], c["size"]
sub, _, _ = bench.build_subject(cfg, P, {"c1": 0, "c2": 1, "c5": 4}[cfg], dev)
cv, cvp, cp = cameras.make_cameras([bench.VIEWS[0]])
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
st = R.BatchedRasterizationSettings(H, H, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, torch.ones(3, device=dev), 0.5, t(cv), t(cvp), 0, t(cp), 1)
with torch.no_grad():
    probe = R.forward_debug(sub["means3D"][None], sub["opacity"][None], colors_precomp=sub["rgb"][None], cov3D_precomp=sub["cov3D"][None], settings=st)
st = st._replace(max_rendered=int(probe["num_rendered"] * 1.25) + 4096)
leaves = [sub[k][None].clone().requires_grad_(True) for k in ("means3D", "rgb", "opacity", "cov3D")]
tgt = torch.rand(1, 3, H, H, device=dev)
mask = (torch.rand(1, 1, H, H, device=dev) > 0.3).float()
one = torch.ones((), device=dev)
R.FUSE_STEP_IN_PYTHON_NODE = True

def step():
    for v in leaves: v.grad = None
    out = R.rasterize_l1_loss_batched(leaves[0], None, None, leaves[1], leaves[2], None, None, leaves[3], st, tgt, mask, 1e-6)
    out[0].backward(one)
    return out[0]

def digest(loss):
    parts = [loss.detach().reshape(1).view(torch.int32).to(torch.int64).sum()]
    for v in leaves: parts.append(v.grad.contiguous().view(torch.int32).to(torch.int64).sum())
    return torch.stack(parts)

ref = digest(step()).cpu()
bad = 0
for k in range(1, steps + 1):
    loss = step()
    if k % every == 0:
        d = digest(loss).cpu()
        if not torch.equal(d, ref): bad += 1; print("step", k, "differs:", d.tolist(), "vs", ref.tolist())
print(f"soak {cfg}: {steps} steps, checked every {every}: {'identical bits throughout' if bad == 0 else str(bad) + ' checks DIFFER'} (loss bits + integer sums of the four gradient tensors' bit patterns)")
