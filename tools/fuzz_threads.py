#!/usr/bin/env python3
"""Randomised concurrency check on the GPU box (dev): three host threads, each on its own HIP stream, render random small scenes at the same time --
the rasterizer + masked L1 node (fused single-view step), the batched rasterizer in a random capacity mode and the upstream-signature per-view op --
forward + backward; every result is compared, bit for bit, with the same call made beforehand on one thread.  What is shared between the threads:
the process-wide pools of pinned count slots, the learned capacities and blob sizes, the autograd engine's worker thread, the caching allocator.
usage: python tools/fuzz_threads.py [seconds] [seed]"""
import os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigman_release_amd import cameras, synthetic
from sigman_release_amd import rasterizer as R

dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def make_job(rng):
    kind = str(rng.choice(["l1", "batched", "per_view"]))
    P = int(rng.choice([50, 1500, 8000])); H = int(rng.integers(16, 220)); W = int(rng.integers(16, 220)); V = int(rng.choice([1, 2, 4]))
    g = synthetic.humanoid(P, int(rng.integers(1, 1 << 30))) if rng.random() < 0.6 else synthetic.random_cloud(P, int(rng.integers(1, 1 << 30)))
    scale = float(rng.choice([0.3, 1.0, 1.0, 8.0]))
    base = [t(g["position"])[None], t(g["rgb"])[None], t(g["opacity"].reshape(P, 1))[None], t((synthetic.covariance_from_gaussians(g) * scale).astype(np.float32))[None]]
    views = [int(v) for v in rng.choice(90, V, replace=False)]
    cv, cvp, cp = (t(x) for x in cameras.make_cameras(views))
    bg = torch.tensor(rng.uniform(0, 1, 3).astype(np.float32), device=dev)
    gen = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
    target = torch.rand(V, 3, H, W, device=dev, generator=gen); gC = torch.randn(V, 3, H, W, device=dev, generator=gen)
    cap = int(rng.choice([-1, 0, 400000]))
    torch.cuda.synchronize()
    return dict(kind=kind, base=base, cv=cv, cvp=cvp, cp=cp, bg=bg, H=H, W=W, V=V, target=target, gC=gC, cap=cap, P=P)


def run_job(j):
    leaves = [x.clone().requires_grad_(True) for x in j["base"]]
    H, W, V = j["H"], j["W"], j["V"]
    if j["kind"] == "l1":
        st = R.BatchedRasterizationSettings(H, W, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, j["bg"], 1.0, j["cv"], j["cvp"], 0, j["cp"], V, False, 400000)
        out = R.rasterize_l1_loss_batched(leaves[0], None, None, leaves[1], leaves[2], None, None, leaves[3], st, j["target"], None, 1.0 / (H * W))
        out[0].backward()
        outs = [out[0], out[2], out[3]]
    elif j["kind"] == "batched":
        st = R.BatchedRasterizationSettings(H, W, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, j["bg"], 1.0, j["cv"], j["cvp"], 0, j["cp"], V, False, j["cap"])
        out = R.rasterize_gaussians_batched(leaves[0], None, None, leaves[1], leaves[2], None, None, leaves[3], st)
        (out[0] * j["gC"]).sum().backward()
        outs = [out[0], out[1], out[3]]
    else:
        imgs = []
        for i in range(V):
            rs = R.GaussianRasterizationSettings(H, W, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, j["bg"], 1.0, j["cv"][i], j["cvp"][i], 0, j["cp"][i], False, False)
            imgs.append(R.GaussianRasterizer(rs)(means3D=leaves[0][0], means2D=torch.zeros_like(leaves[0][0]), opacities=leaves[2][0], colors_precomp=leaves[1][0],
                                                 cov3D_precomp=leaves[3][0])[0])
        img = torch.stack(imgs)
        (img * j["gC"]).sum().backward()
        outs = [img]
    torch.cuda.current_stream().synchronize()
    return [np.atleast_1d(x.detach().cpu().numpy()).copy() for x in outs] + [x.grad.detach().cpu().numpy().copy() for x in leaves]


def run(seconds=60.0, seed=1, n_threads=3):
    rng = np.random.default_rng(seed)
    t_end = time.time() + float(seconds)
    rounds = 0
    while time.time() < t_end:
        jobs = [[make_job(rng) for _ in range(4)] for _ in range(n_threads)]
        want = [[run_job(j) for j in js] for js in jobs]                      # serial reference (also learns capacities / blob sizes first)
        got, errs = [None] * n_threads, []
        def worker(k):
            try:
                with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                    got[k] = [run_job(j) for j in jobs[k] for _ in range(2)]
            except Exception as ex:      # noqa: BLE001
                errs.append((k, repr(ex)))
        ths = [threading.Thread(target=worker, args=(k,)) for k in range(n_threads)]
        [th.start() for th in ths]; [th.join() for th in ths]
        torch.cuda.synchronize()
        assert not errs, errs
        R.check_pending_overflows(True)
        for k in range(n_threads):
            for i, j in enumerate(jobs[k]):
                for rep in range(2):
                    for a, b in zip(want[k][i], got[k][2 * i + rep]):
                        if not np.array_equal(a.view(np.uint8), b.view(np.uint8)):
                            if a.size == 1 and np.allclose(a, b, rtol=1e-5):          # (the unfused loss kernel's float atomics: an auto-capacity re-run is not at play here)
                                continue
                            raise AssertionError(("threads vs serial", j["kind"], dict(P=j["P"], H=j["H"], W=j["W"], V=j["V"], cap=j["cap"]), int((a != b).sum()), a.shape))
        rounds += 1
    return rounds


if __name__ == "__main__":
    print("fuzz ok:", run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1), "rounds of 3 threads x 8 calls")
