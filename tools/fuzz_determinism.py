#!/usr/bin/env python3
"""Randomised run-to-run check on the GPU box (dev): random batches (a fifth of them with NaN / Inf / huge entries sprinkled over the inputs) (1-3 subjects x 1-8 views, odd image sizes, humanoids / random clouds with
random covariance scale, colours + covariances or spherical harmonics + scales / rotations, with and without gradients into depth and alpha,
every capacity mode) rendered forward + backward TWICE: images, radii, depth, alpha and all gradients identical bit for bit.  The library has no
float atomics and no order-dependent reductions on these paths, so any difference is a race.     usage: python tools/fuzz_determinism.py [seconds] [seed]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigman_release_amd import cameras, synthetic
from sigman_release_amd import rasterizer as R


def run(seconds=60.0, seed=1, max_scenes=None):
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(seed)
    t_end = time.time() + float(seconds)
    n = 0
    kinds = {}
    while time.time() < t_end and (max_scenes is None or n < max_scenes):
        S = int(rng.choice([1, 1, 2, 3])); V = int(rng.choice([1, 1, 2, 4, 8]))
        H = int(rng.integers(8, 300)); W = int(rng.integers(8, 300))
        P = int(rng.choice([1, 50, 1000, 6000, 20000]))
        if os.environ.get("FUZZ_THIN"):               # extreme aspect ratios: a few pixels by a few thousand
            a, b = int(rng.integers(1, 20)), int(rng.integers(300, 5000))
            H, W = (a, b) if rng.random() < 0.5 else (b, a)
        if os.environ.get("FUZZ_BIG"):                # full-size scenes: several rounds per tile list, deep tiles, every sort flavour's large path
            S, V = 1, int(rng.choice([1, 2, 8])); H = W = int(rng.choice([512, 496, 1024])); P = int(rng.choice([100000, 300000]))
        sh = rng.random() < 0.3
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        subs = []
        for _ in range(S):
            g = synthetic.humanoid(P, int(rng.integers(1, 1 << 30))) if rng.random() < 0.6 else synthetic.random_cloud(P, int(rng.integers(1, 1 << 30)))
            subs.append(g)
        scale = float(rng.uniform(1.0, 20.0)) if rng.random() < 0.4 else 1.0
        means = torch.stack([t(g["position"]) for g in subs]); op = torch.stack([t(g["opacity"].reshape(P, 1)) for g in subs])
        if sh:
            deg = int(rng.integers(0, 4)); M = (deg + 1) ** 2
            feat = t((rng.normal(size=(S, P, M, 3)) * 0.3).astype(np.float32))
            q = rng.normal(size=(S, P, 4)).astype(np.float32); q /= np.linalg.norm(q, axis=2, keepdims=True)
            sc = t((rng.uniform(0.005, 0.05, (S, P, 3)) * np.sqrt(scale)).astype(np.float32)); rot = t(q)
            base = dict(means3D=means, opacities=op, sh=feat, scales=sc, rotations=rot)
        else:
            deg = 0
            cov = torch.stack([t((synthetic.covariance_from_gaussians(g) * scale).astype(np.float32)) for g in subs])
            base = dict(means3D=means, opacities=op, colors_precomp=torch.stack([t(g["rgb"]) for g in subs]), cov3Ds_precomp=cov)
        if rng.random() < 0.2:                        # non-finite entries (a diverged decoder): nothing may fault, and two runs still agree bit for bit
            for k in [k for k in base if k != "rotations"]:
                if rng.random() < 0.5:
                    flat = base[k].reshape(-1)
                    idx = torch.from_numpy(rng.integers(0, flat.numel(), size=min(flat.numel(), int(rng.integers(1, 20))))).to(dev)
                    flat[idx] = float(rng.choice([np.nan, np.inf, -np.inf, 1e30, -1e30]))
        views = [int(v) for v in rng.choice(90, V, replace=False)]
        cv, cvp, cp = cameras.make_cameras(views * S)
        da = bool(rng.random() < 0.3)
        cap = int(rng.choice([0, -1, 1]))
        bg = torch.tensor(rng.uniform(0, 1, 3).astype(np.float32), device=dev)
        st = R.BatchedRasterizationSettings(H, W, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, bg, float(rng.uniform(0.5, 1.5)), t(cv), t(cvp), deg, t(cp), V, bool(os.environ.get("FUZZ_DEBUG")),
                                            cap, True if (da and rng.random() < 0.5) else None)
        only = os.environ.get("FUZZ_ONLY")
        skip = only is not None and int(only) != n
        if cap == 1 and skip:
            rng.uniform(1.0, 1.5); rng.integers(1, 3000)
        elif cap == 1:
            with torch.no_grad():
                kw = {k: v for k, v in base.items() if k not in ("means3D", "opacities")}
                kw = {("shs" if k == "sh" else ("cov3D_precomp" if k == "cov3Ds_precomp" else k)): v for k, v in kw.items()}
                cnt = int(R.forward_debug(base["means3D"], base["opacities"], settings=st._replace(max_rendered=0), **kw)["num_rendered"])
            st = st._replace(max_rendered=int(cnt * rng.uniform(1.0, 1.5)) + int(rng.integers(1, 3000)))
        gen = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
        gC = torch.randn(S * V, 3, H, W, device=dev, generator=gen); gD = torch.randn(S * V, 1, H, W, device=dev, generator=gen); gA = torch.randn(S * V, 1, H, W, device=dev, generator=gen)
        if os.environ.get("FUZZ_DUMP") and not skip:
            os.makedirs("gpurun_out", exist_ok=True)
            np.savez("gpurun_out/fuzz_scene.npz", views=np.array(views), H=H, W=W, V=V, S=S, **{k: v.cpu().numpy() for k, v in base.items()})
        if os.environ.get("FUZZ_CAP"):
            st = st._replace(max_rendered=int(os.environ["FUZZ_CAP"]))
        if os.environ.get("FUZZ_VERBOSE"):
            print("scene", n, views, dict(S=S, V=V, H=H, W=W, P=P, sh=sh, deg=deg, da=da, cap=st.max_rendered, ckpt_da=st.depth_alpha_grads, scale=round(scale, 2), seed=seed), flush=True)
        if skip:
            n += 1
            continue
        res = []
        for _rep in range(2):
            d = {k: v.clone().requires_grad_(True) for k, v in base.items()}
            try:
                color, radii, depth, alpha = R.rasterize_gaussians_batched(d["means3D"], None, d.get("sh"), d.get("colors_precomp"), d["opacities"], d.get("scales"),
                                                                           d.get("rotations"), d.get("cov3Ds_precomp"), st)
                loss = (color * gC).sum()
                if da:
                    loss = loss + (depth * gD).sum() + (alpha * gA).sum()
                loss.backward()
                torch.cuda.synchronize()
            except RuntimeError as ex:                  # a scene whose buffers do not fit the GPU must be refused with an error (forward or backward), never fault
                if "allocat" not in str(ex).lower() and "out of memory" not in str(ex).lower():
                    raise
                kinds["refused (memory)"] = kinds.get("refused (memory)", 0) + 1
                res = None
                del d
                torch.cuda.empty_cache()
                break
            res.append([x.detach().cpu().numpy().copy() for x in (color, radii, depth, alpha)] + [d[k].grad.detach().cpu().numpy().copy() for k in sorted(d)])
        R.check_pending_overflows(True)
        if res is None:
            n += 1
            continue
        names = ["color", "radii", "depth", "alpha"] + ["d_" + k for k in sorted(base)]
        cfg = dict(S=S, V=V, H=H, W=W, P=P, sh=sh, da=da, cap=st.max_rendered, scale=round(scale, 2), scene=n, seed=seed)
        for nm, a, b in zip(names, *res):
            if not np.array_equal(a.view(np.uint8), b.view(np.uint8)):
                bad = np.argwhere(a != b)
                raise AssertionError(("run to run", nm, cfg, len(bad), bad[:4].tolist(), a[tuple(bad[0])], b[tuple(bad[0])]))
        key = ("seg" if ((H + 15) // 16) * ((W + 15) // 16) * S * V <= 2048 else "wave", "sh" if sh else "rgb")
        kinds[key] = kinds.get(key, 0) + 1
        n += 1
    return n, kinds


if __name__ == "__main__":
    n, kinds = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("fuzz ok:", n, "scenes", kinds)
