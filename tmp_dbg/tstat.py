import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigman_release_amd import cameras, synthetic
from sigman_release_amd import rasterizer as R
dev = torch.device("cuda:0")
g = synthetic.humanoid(100000, 1); cov = synthetic.covariance_from_gaussians(g)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
cv, cvp, cp = cameras.make_cameras([30])
st = R.BatchedRasterizationSettings(512, 512, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, torch.ones(3, device=dev), 1.0, t(cv), t(cvp), 0, t(cp), 1)
with torch.no_grad():
    d = R.forward_debug(t(g["position"])[None], t(g["opacity"]).reshape(1,-1,1), colors_precomp=t(g["rgb"])[None], cov3D_precomp=t(cov)[None], settings=st)
r = d["ranges"][0].cpu().numpy().astype(np.int64)
n = r[:,1]-r[:,0]
print("tiles", len(n), "occupied", (n>0).sum(), "R", n.sum(), "mean(occ)", n[n>0].mean(), "p50", np.percentile(n[n>0],50), "p90", np.percentile(n[n>0],90), "p99", np.percentile(n[n>0],99), "max", n.max())
print("tiles >512:", (n>512).sum(), ">1024:", (n>1024).sum(), ">2048:", (n>2048).sum(), "sum chunks", np.ceil(n/512).sum())
nc = d["n_contrib"][0].cpu().numpy()
print("n_contrib mean over covered px", nc[nc>0].mean(), "max", nc.max(), "covered frac", (nc>0).mean())
# depth at which pixels stop: per tile max n_contrib vs n
nct = nc.reshape(32,16,32,16).transpose(0,2,1,3).reshape(1024,256).max(1)
print("per-tile walk length needed (max n_contrib): mean", nct[n>0].mean(), "max", nct.max(), " vs list len mean", n[n>0].mean())
heavy = np.argsort(-n)[:8]
print("heaviest tiles: n", n[heavy], "needed", nct[heavy])
