#!/usr/bin/env python
"""bench.py -- rasterizer hot-path benchmark (contract in the task brief; metric from BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1], "C2"): one procedural-humanoid subject of 100 000 Gaussians
(sigman_release_amd.synthetic.humanoid, seed 1; stands in for the SMPL-X subdivided template), rendered at
512x512 from the reference camera rig, forward + backward, fp32, reference mode (colors_precomp + cov3D_precomp,
gs.py:98-106), white background, loss = mean |clamp(img,0,1) - gt| with gt = render of a perturbed copy
(stand-in for the masked L1 of core/loss/whole_loss.py:126-131).

One "step" = one pass of the hot path over one batch: at N=1 ONE view (rig view 0030) per step.  At N>1 the views
[30,37,45,53,65,85,0,8] of the same subject are sharded one-view-per-GPU (weak scaling: per-GPU work fixed) with the
exchange of sigman_release_amd/parallel.py: by default what BASELINE.json's north_star names -- replicated attributes, RCCL
all-reduce of the image-space loss, overlapped with the backward; `--exchange full` adds the attribute broadcast and the
all-reduce of the attribute gradients.

`value` = views/s of the whole job with inputs resident in HBM.  `roofline` is for the dominant kernel, timed with HIP
events recorded by the library on the launch stream inside the timed region.  `cpu_baseline` is the CPU oracle
(oracle/gsplat_ref.c, OpenMP) on the same inputs, rank 0 at N=1 only.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before torch/HIP initialise: see sigman_release_amd/__init__.py


def _pin_host_threads():
    """One process per GPU, pinned to 4 neighbouring cores (before torch creates its threads): the C2 step is ~0.2 ms of
    GPU work driven by two host threads (Python + autograd engine); left to the scheduler of a 2-socket / 256-thread host they
    wander across CCXs and NUMA nodes and the step time varies 0.21-0.34 ms run to run; pinned it is 0.21 ms every time.
    Rank r of n local ranks gets cores [r * (physical / n), +4) -- ranks spread evenly over both sockets like the GPUs.
    SIGMAN_NO_PIN=1 disables it.  Returns the original affinity (the cpu_baseline leg needs all cores back)."""
    if os.environ.get("SIGMAN_NO_PIN") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        orig = os.sched_getaffinity(0)
        lr, lw = int(os.environ.get("LOCAL_RANK", "0")), max(int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))), 1)
        phys = max((os.cpu_count() or 2) // 2, 1)               # SMT siblings are numbered in the upper half
        start = (lr % lw) * (phys // lw)
        want = {c for c in range(start, start + 4) if c in orig}
        if len(want) >= 2:
            os.sched_setaffinity(0, want)
        return orig
    except OSError:
        return None


_ORIG_AFFINITY = _pin_host_threads()

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from sigman_release_amd import _cabi, cameras, parallel, synthetic  # noqa: E402
from sigman_release_amd import rasterizer as R  # noqa: E402
from sigman_release_amd.losses import clamped_l1_loss  # noqa: E402

VIEWS = (30, 37, 45, 53, 65, 85, 0, 8)
KERNELS = {0: "preprocess_fwd", 1: "scan_block_sums", 2: "duplicate_keys", 3: "radix_sort(all passes)", 4: "tile_ranges",
           5: "render_fwd", 6: "render_bwd", 7: "preprocess_bwd"}
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak (6.3 TB/s achievable)


def algorithmic_bytes(kid: int, P: int, Rn: int, HW: int, tiles: int, views: int) -> float:
    """SURVEY.md 8(d) per-view figures x views per launch."""
    per_view = {0: 76 * P, 1: 8 * P, 2: 20 * P + 12 * Rn, 3: 24 * Rn, 4: 8 * Rn + 8 * tiles, 5: 44 * Rn + 24 * HW,
                6: 88 * Rn + 28 * HW, 7: 108 * P}[kid]
    return float(per_view)     # Rn is already the batch total when views > 1 are launched together


def build_subject(P: int, seed: int, dev):
    g = synthetic.humanoid(P, seed)
    cov = synthetic.covariance_from_gaussians(g)       # host stand-in for distCUDA2 + get_covariance (gs.py:70-73), untimed
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return dict(means3D=t(g["position"]), cov3D=t(cov), opacity=t(g["opacity"].reshape(P, 1)), rgb=t(g["rgb"])), g, cov


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--gaussians", type=int, default=100_000)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--views-per-step", type=int, default=1, help="views per GPU per step (C2 = 1)")
    ap.add_argument("--exchange", choices=("loss", "full"), default="loss",
                    help="N>1: 'loss' = north_star's protocol (replicated attributes, loss all-reduce overlapped with the backward); "
                         "'full' = attribute broadcast + all-reduce of gradients and loss (sigman_release_amd/parallel.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exact-sync", action="store_true", help="read num_rendered back every step (upstream behaviour) instead of the sync-free capacity mode")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback for the product path)"
    local_rank %= max(torch.cuda.device_count(), 1)     # (lets a 1-GPU box exercise the N>1 code path with gloo)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("SIGMAN_BENCH_BACKEND", "nccl"), **({"device_id": dev} if os.environ.get("SIGMAN_BENCH_BACKEND", "nccl") == "nccl" else {}))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    P, H, W = args.gaussians, args.size, args.size
    vps = args.views_per_step
    subj, g_host, cov_host = build_subject(P, 1, dev)
    all_views = [VIEWS[i % len(VIEWS)] for i in range(world * vps)]
    my_views = [all_views[i] for i in parallel.shard_views(len(all_views), rank, world)]
    bg = torch.ones(3, device=dev)
    cv, cvp, cp = cameras.make_cameras(my_views)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    st = R.BatchedRasterizationSettings(H, W, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, bg, 0.5, t(cv), t(cvp), 0, t(cp), len(my_views))
    if not args.exact_sync:
        # sync-free mode: size the binning buffers from one exact (untimed) forward, +25 % head-room; an overflow would raise
        with torch.no_grad():
            probe = R.forward_debug(subj["means3D"][None], subj["opacity"][None], colors_precomp=subj["rgb"][None],
                                    cov3D_precomp=subj["cov3D"][None], settings=st)
        st = st._replace(max_rendered=int(probe["num_rendered"] * 1.25) + 4096)
        del probe
    n_total_views = len(all_views)
    norm = 1.0 / (n_total_views * 3 * H * W)

    # ground truth: render of a perturbed copy (untimed)
    with torch.no_grad():
        rng = torch.Generator(device="cpu").manual_seed(1234)
        pert = lambda x, s: x + s * torch.randn(x.shape, generator=rng).to(dev)
        gt, _, _, _ = R.rasterize_gaussians_batched(pert(subj["means3D"], 2e-3)[None], None, None,
                                                    (subj["rgb"] * 0.9)[None], subj["opacity"][None], None, None,
                                                    subj["cov3D"][None], st)
        gt = gt.clamp(0, 1)

    def render_loss(means3D, cov3D, opacity, rgb, _views=None):
        # gs.py:98-107 rasterize + clamp, whole_loss.py:126-131 L1: one autograd node, loss kernel right behind the compositing kernel
        return R.rasterize_l1_loss_batched(means3D[None], None, None, rgb[None], opacity[None], None, None, cov3D[None], st, gt,
                                           None, norm)[0]

    leaves = {k: v.clone().requires_grad_(True) for k, v in subj.items()}
    one = torch.ones((), device=dev)
    packed = parallel.pack_attributes(subj["means3D"], subj["cov3D"], subj["opacity"], subj["rgb"])

    def step():
        if world == 1:
            for v in leaves.values():
                v.grad = None
            loss = render_loss(leaves["means3D"], leaves["cov3D"], leaves["opacity"], leaves["rgb"])
            loss.backward(one)              # explicit seed: autograd would otherwise launch a ones_like fill kernel every step
            return loss
        loss, grad = parallel.view_parallel_step(packed, all_views, lambda m, c, o, r, mine: render_loss(m, c, o, r, mine),
                                                 exchange=args.exchange, seed_grad=one, pack_grad=False)
        return loss

    L = _cabi.lib()
    L.sgr_prof_configure.argtypes = [C.c_uint32]
    L.sgr_prof_collect.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_uint32)]

    def collect():
        ms = (C.c_double * 16)()
        cnt = (C.c_uint32 * 16)()
        _cabi.check(L.sgr_prof_collect(ms, cnt), "sgr_prof_collect")
        return {k: (ms[k], cnt[k]) for k in range(16) if cnt[k]}

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- warmup (untimed), with a full per-kernel profile of the last 3 warmup steps to pick the dominant kernel
    for i in range(args.warmup):
        if i == max(args.warmup - 3, 0):
            torch.cuda.synchronize()
            L.sgr_prof_configure(0xFFFF)
        step()
    torch.cuda.synchronize()
    prof_all = collect()
    nprof = max(args.warmup - max(args.warmup - 3, 0), 1)
    breakdown = {KERNELS[k]: round(v[0] / nprof, 4) for k, v in prof_all.items() if k in KERNELS}
    dominant = max(prof_all, key=lambda k: prof_all[k][0]) if prof_all else 6
    L.sgr_prof_configure(0)
    for _ in range(3):                           # re-warm without the profiler (lets the launch-graph cache fill)
        step()

    # ---- timed region (no per-kernel events here: event pairs would force plain launches instead of graph replay)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # ---- the same K steps again with HIP-event pairs around the dominant kernel (live roofline measurement)
    L.sgr_prof_configure(1 << dominant)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dom = collect()
    L.sgr_prof_configure(0)

    # ---- workload counters (from the run's own buffers)
    with torch.no_grad():
        dbg = R.forward_debug(subj["means3D"][None], subj["opacity"][None], colors_precomp=subj["rgb"][None],
                              cov3D_precomp=subj["cov3D"][None], settings=st._replace(max_rendered=0))
        Rn = int(dbg["num_rendered"])
        S_visits = int(dbg["n_contrib"].to(torch.int64).sum().item())
    ms_per_step = elapsed / args.steps * 1e3
    views_per_s = n_total_views / (ms_per_step * 1e-3)
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    dom_ms, dom_n = dom.get(dominant, (0.0, 0))
    dom_avg_ms = dom_ms / max(dom_n, 1)
    abytes = algorithmic_bytes(dominant, P * len(my_views), Rn, H * W * len(my_views), tiles * len(my_views), len(my_views))
    achieved = abytes / (dom_avg_ms * 1e-3) / 1e9 if dom_avg_ms > 0 else 0.0

    # HBM traffic of the dominant kernel from the committed rocprofv3 PMC summary of this same command (separate --pmc passes,
    # gfx950 FETCH_SIZE correction applied as MI355X_MICROARCH.md prescribes); null when no summary matches this workload
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_c2.json")))
        if P == 100_000 and H == 512 and len(my_views) == 1:
            traffic = pmc["kernels"].get(KERNELS.get(dominant, ""), {}).get("hbm_bytes_corrected")
    except Exception:
        traffic = None

    out = {
        "metric": f"rendered views/sec (fwd+bwd) at {H}x{W}, {P} Gaussians/view",
        "value": round(views_per_s, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"C2: procedural humanoid (SMPL-X stand-in), {P} Gaussians, {len(my_views)} view(s)/GPU/step "
                               f"{H}x{W}, fwd+bwd, colors_precomp+cov3D_precomp, clamp+L1 loss",
                   "views_per_step_total": n_total_views,
                   "parallelism": (f"view-parallel x{world}, exchange={args.exchange}" if world > 1 else "single GPU"),
                   "num_rendered_per_gpu": Rn, "gaussian_pixel_visits_per_gpu": S_visits,
                   "binning_buffers": "exact (D2H read of num_rendered per step)" if args.exact_sync else f"pre-sized, max_rendered={st.max_rendered} (sync-free)"},
        "gaussian_pixel_ops_per_s_per_gpu": round(S_visits / (ms_per_step * 1e-3), 1),
        "tile_instances_per_s_per_gpu": round(Rn / (ms_per_step * 1e-3), 1),
        "roofline": {"bound": "hbm", "kernel": KERNELS.get(dominant, str(dominant)), "achieved": round(achieved, 2),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "algorithmic_bytes_per_launch": abytes, "avg_launch_ms": round(dom_avg_ms, 5), "launches": dom_n},
        "kernel_ms_per_step": breakdown,
        "loss": float(loss.detach()),
    }

    out["config"]["host_threads"] = ("pinned to cores %s" % sorted(os.sched_getaffinity(0))) if _ORIG_AFFINITY is not None else "not pinned"
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if _ORIG_AFFINITY is not None:
            os.sched_setaffinity(0, _ORIG_AFFINITY)      # the OpenMP oracle gets every host core
        out["cpu_baseline"] = cpu_baseline(g_host, cov_host, my_views[0], H, W, gt[0].cpu().numpy(), norm)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(g_host, cov_host, view, H, W, gt, norm):
    """CPU oracle (test infrastructure) fwd+bwd on the same C2 inputs; bounded sample, all host cores via OpenMP."""
    from oracle import ref
    cv, cvp, cp = cameras.make_cameras([view])
    P = g_host["position"].shape[0]
    kw = dict(viewmatrix=cv[0], projmatrix=cvp[0], campos=cp[0], bg=np.ones(3, np.float32), tanfovx=cameras.TAN_HALF_FOV,
              tanfovy=cameras.TAN_HALF_FOV, image_height=H, image_width=W)
    times = []
    for i in range(4):
        t0 = time.perf_counter()
        st = ref.forward(g_host["position"], g_host["opacity"].reshape(P), colors_precomp=g_host["rgb"], cov3D_precomp=cov_host, **kw)
        img = np.clip(st.color, 0, 1)
        loss_cpu = float(np.abs(img - gt).sum() * norm)   # noqa: F841  (same clamp + L1 epilogue as the GPU step)
        gC = (np.sign(img - gt) * ((st.color > 0) & (st.color < 1)) * norm).astype(np.float32)
        ref.backward(st, gC)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times[1:]))
    return {"value": round(1.0 / med, 4), "unit": "views/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "median of 3 x (1 view fwd+bwd, same C2 inputs) after 1 warm-up; oracle/gsplat_ref.c with OpenMP on all host cores",
            "seconds_per_view": round(med, 4)}


if __name__ == "__main__":
    main()
