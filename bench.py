#!/usr/bin/env python
"""bench.py -- rasterizer hot-path benchmark (contract in the task brief; metric from BASELINE.json).

    python bench.py [--gpus N] [--config c1|c2|c3|c4|c5] [--steps K] [--warmup W]

`--gpus N` with N > 1 LAUNCHES ITS OWN N RANKS (one process per GPU, `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 ...` re-executing this file) when it is not already running under a launcher
(WORLD_SIZE unset); under the driver's own `torch.distributed.run` command line it just joins.  It never falls back to
fewer ranks: `n_gpus` in the JSON line is `dist.get_world_size()`, and a mismatch with `--gpus`, or fewer visible GPUs
than ranks on the RCCL backend, ends with a non-zero exit code.  (`SIGMAN_BENCH_BACKEND=gloo` lets a 1-GPU box run the
N > 1 code path with all ranks on cuda:0 -- a plumbing check, not a measurement.)

Configs (BASELINE.json `configs`; subjects are procedural humanoids, sigman_release_amd.synthetic):
  c1  10 000 random Gaussians (U([-0.8,0.8]^3), isotropic log-uniform scales), 1 view 256x256, fwd+bwd: the reference's CPU-runnable case
      (its CPU time next to C2's); weak scaling like c2.
  c2 (default, the configuration the metric is quoted on): 100 000 Gaussians, 1 view 512x512 per GPU per step, fwd+bwd,
      clamp+L1 loss.  Weak scaling: rank r renders view VIEWS[r] of the same subject.
  c3  VAE render-loss step: 8 subjects x 8 views [30,37,45,53,65,85,0,8] at 512x512, 100 000 Gaussians each, fwd+bwd.
      STRONG scaling: the 8 views of every subject are sharded {v : v mod N = r}; at N=8 one view per GPU per subject.
  c4  decode path: 200 000 Gaussians, the 90-view orbit at 1024x1024, forward only.  Strong scaling (90 views -> 11-12 per GPU).
  c5  1M-Gaussian stress (10 jittered layers), 512x512, depth + alpha gradients on, fwd+bwd.  Weak scaling (1 view per GPU).
For N > 1 the exchange is sigman_release_amd/parallel.py: by default what BASELINE.json's north_star names -- replicated
attributes, RCCL all-reduce of the image-space loss, overlapped with the backward; `--exchange full` adds the attribute
broadcast and the all-reduce of the attribute gradients -- for c3 ONE broadcast of the [8 * 13 * P] pack (42 MB) and ONE all-reduce of the
packed gradients + loss per step; `--exchange full-pipelined [--pipeline-chunks K]` does the same per chunk of subjects (default: per subject)
with chunk c+1's broadcast and chunk c's all-reduce in flight while the other chunk is rendered.  c4 is forward-only and needs no collective.

One "step" = one pass of the hot path over one batch (all view slots of this rank in ONE launch chain).  `value` = views/s
of the whole job with inputs resident in HBM.  The timed step is the rasterizer (+ fused loss): distCUDA2 + get_covariance
(gs.py:70-73) run once per subject outside it; their cost is reported as `frontend_ms_per_subject`, and `variants.renderer_render_ms_per_step`
times what the reference's render() really pays: GaussianRenderer.render (3-NN + covariance + rasterizer + clamp) forward and backward.
A timed region shorter than 300 ms is not a measurement on this pool (its boxes have slow phases of 10-50 ms: a 100-ms region that met one read
0.1474 ms per step with every repeat window behind it at 0.131): when `--steps K` would give a shorter one, as many steps as 300 ms hold (at least
100) are timed instead and the line says so (`steps` = what was timed, `steps_requested` = K).  `gpu_ms_per_step` is the same region between two HIP
events on the launch stream (first kernel to last kernel: it excludes the host's final synchronise, not the gaps the host may leave between
steps); `windows` repeats the K steps `--windows` more times and reports min / median / max per step of wall and event time, so that a slow
phase of the box or of its clocks shows up as spread instead of being folded into one number; `sclk_mhz` = the GPU's shader clock sampled
from sysfs by a side process while those windows run (not during the timed region: each read is a message to the SMU and cost it ~0.5 us per step);
`sclk_mhz_probe` = the clock a wave really ran at right before and right behind the timed region, from the chip's own counters (the sysfs node has
shown 95-157 MHz under full load on boxes of this pool); `host_queue` says who set the pace of the timed region: `queue_drain_steps` = how many
steps' worth of work the GPU still held when the host had issued its last step (<< 1: the GPU was waiting for the host -- every 0.14-0.157-ms
reading of this build was of that kind, with the library's default count wait on a busy shared host; profiles/r05_count_wait_ab.txt -- which is
why the timed step runs with set_count_wait("lazy:4"): the host may be five steps ahead, every step's instance count is still checked, at most
four steps later and once more behind the region; SIGMAN_COUNT_WAIT=own times the library default).  The loss of the timed step is the reference's MASKED L1 (whole_loss.py:126-131:
gt_masks multiplies prediction and target; mask = ground-truth alpha > 0.5) unless `--no-mask`.
`roofline` is for the dominant kernel, timed with HIP events recorded by the library on the launch stream inside the
timed region; `roofline.traffic` comes from the committed rocprofv3 PMC summary of this same command
(profiles/r06_pmc_<config>.json, else the newest older one; null when there is none for the workload).  `cpu_baseline` is the CPU oracle
(oracle/gsplat_ref.c, OpenMP) on a bounded sample of the same inputs, rank 0 at N=1 only.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", choices=("c1", "c2", "c3", "c4", "c5"), default="c2")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--gaussians", type=int, default=None, help="override the config's Gaussians per subject")
    ap.add_argument("--size", type=int, default=None, help="override the config's image size")
    ap.add_argument("--views-per-step", type=int, default=1, help="c2/c5: views per GPU per step")
    ap.add_argument("--exchange", choices=("loss", "full", "full-pipelined"), default="loss",
                    help="N>1: 'loss' = north_star's protocol (replicated attributes, loss all-reduce overlapped with the backward); "
                         "'full' = attribute broadcast + all-reduce of gradients and loss (sigman_release_amd/parallel.py); "
                         "'full-pipelined' = the same per chunk of subjects, chunk c+1's broadcast and chunk c's all-reduce travelling while "
                         "the other chunk is rendered (parallel.view_parallel_subjects)")
    ap.add_argument("--pipeline-chunks", type=int, default=0, help="full-pipelined: pipeline stages per step (default: one per subject)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mask", action="store_true", help="plain L1 instead of the reference's masked L1 (whole_loss.py:126-131: gt_masks = ground-truth alpha > 0.5)")
    ap.add_argument("--no-sclk", action="store_true", help="do not sample the shader clock (a side process reads an amdgpu sysfs node every 10 ms while the repeat windows run)")
    ap.add_argument("--windows", type=int, default=8, help="untimed-for-the-headline repeat windows of the same K steps behind the timed region (spread report)")
    ap.add_argument("--no-variants", action="store_true", help="skip the unpinned / exact-sync / per-view-loop re-runs (N=1, c2/c3)")
    ap.add_argument("--order", choices=("random", "template"), default="random",
                    help="order of a subject's Gaussians in memory: 'random' (headline: a random permutation, the pessimistic case for every gather) or "
                         "'template' (spatially coherent, as the faces of the reference's template mesh: a VARIANT, never the headline)")
    ap.add_argument("--exact-sync", action="store_true", help="read num_rendered back every step (upstream behaviour) instead of the sync-free capacity mode")
    return ap.parse_args(argv)


def _self_launch(args) -> int:
    """--gpus N without a launcher: start N ranks of this file under torch.distributed.run and hand their exit code back."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {args.gpus} without a launcher: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd)


if __name__ == "__main__":
    _ARGS = parse_args()
    if _ARGS.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(_ARGS))          # (before any thread pinning: children inherit the affinity mask)


_COUNT_WAIT_DEFAULT = "lazy:4"      # (the library default stays "own"; SIGMAN_COUNT_WAIT=own times that)


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
_DEBUG_BLOCKS = os.environ.get("SIGMAN_BENCH_DEBUG") == "1"


def _set_affinity_all_threads(mask):
    """sched_setaffinity(0, ..) moves the calling thread only; the autograd engine's thread exists by now."""
    for tid in os.listdir("/proc/self/task"):
        try:
            os.sched_setaffinity(int(tid), mask)
        except OSError:
            pass


def _choose_host_cores(step, sync_all, t_step):
    if _ORIG_AFFINITY is None:
        return "not pinned (SIGMAN_NO_PIN=1)"
    base = sorted(os.sched_getaffinity(0))
    cands = [("cores %s" % base, set(base))]
    if not dist.is_initialized():                               # (N > 1: the same two options on every rank -- the trial times are all-reduced)
        for off in (8, 16):                                     # the same quad one / two CCXs further
            alt = {c + off for c in base}
            if alt <= _ORIG_AFFINITY:
                cands.append(("cores %s" % sorted(alt), alt))
    cands.append(("not pinned", set(_ORIG_AFFINITY)))
    n = max(10, min(2000, int(0.03 / t_step)))
    trial = [float("inf")] * len(cands)
    for _round in range(2):                                     # every option twice, the better time counts: the host's state also moves with time
        for k, (name, mask) in enumerate(cands):
            _set_affinity_all_threads(mask)
            for _ in range(max(3, n // 10)):
                step()
            sync_all()
            t0 = time.perf_counter()
            for _ in range(n):
                step()
            sync_all()
            trial[k] = min(trial[k], (time.perf_counter() - t0) / n)
    if dist.is_initialized():                                   # one decision for the job: every rank takes the option that is best for the slowest rank
        tt = torch.tensor(trial, dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        trial = [float(x) for x in tt.tolist()]
    best = min(range(len(cands)), key=lambda i: trial[i])
    if trial[best] > 0.97 * trial[0]:                           # within the noise of a 30-ms trial: keep the start-up choice
        best = 0
    _set_affinity_all_threads(cands[best][1])
    for _ in range(n):                                          # settle on the chosen cores before the timed region starts
        step()
    sync_all()
    return ("host threads on %s (untimed %d-step trials, ms per step: " % (cands[best][0], n)
            + ", ".join("%s %.4f" % (c[0], t * 1e3) for c, t in zip(cands, trial)) + ")")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

sys.path.insert(0, ROOT)

from sigman_release_amd import _cabi, cameras, parallel, synthetic  # noqa: E402
from sigman_release_amd import rasterizer as R  # noqa: E402

VIEWS = (30, 37, 45, 53, 65, 85, 0, 8)
KERNELS = {0: "preprocess_fwd", 1: "scan_block_sums", 2: "duplicate_keys", 3: "radix_sort(all passes)", 4: "tile_ranges",
           5: "render_fwd", 6: "render_bwd", 7: "preprocess_bwd", 10: "clamped_l1"}
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak (6.3 TB/s achievable)

# name -> (Gaussians per subject, image size, subjects, backward?, depth+alpha grads?, scaling, steps, warmup)
CONFIGS = {
    "c1": dict(P=10_000, size=256, subjects=1, bwd=True, da=False, scaling="weak", steps=200, warmup=20,
               label="C1: 10k random Gaussians (the reference's CPU-runnable plumbing case)"),
    "c2": dict(P=100_000, size=512, subjects=1, bwd=True, da=False, scaling="weak", steps=200, warmup=20,
               label="C2: procedural humanoid (SMPL-X stand-in)"),
    "c3": dict(P=100_000, size=512, subjects=8, bwd=True, da=False, scaling="strong", steps=40, warmup=6,
               label="C3: VAE render-loss step, 8 subjects x 8 views"),
    "c4": dict(P=200_000, size=1024, subjects=1, bwd=False, da=False, scaling="strong", steps=20, warmup=6,
               label="C4: decode path, 90-view orbit, forward only"),
    "c5": dict(P=1_000_000, size=512, subjects=1, bwd=True, da=True, scaling="weak", steps=30, warmup=5,
               label="C5: 1M-Gaussian stress (10 jittered layers), depth+alpha gradients on"),
}


def algorithmic_bytes(kid: int, P: int, Rn: int, HW: int, tiles: int, masked: bool = False) -> float:
    """SURVEY.md 8(d) per-view figures; P, Rn, HW, tiles are totals over the view slots of one launch (loss kernel: colour + target + gradient,
    12 B per pixel each, + 4 B of mask)."""
    return float({0: 76 * P, 1: 8 * P, 2: 20 * P + 12 * Rn, 3: 24 * Rn, 4: 8 * Rn + 8 * tiles, 5: 44 * Rn + 24 * HW,
                  6: 88 * Rn + 28 * HW, 7: 108 * P, 10: (40 if masked else 36) * HW}[kid])


def build_subject(cfg_name: str, P: int, seed: int, dev, order: str = "random"):
    if cfg_name == "c1":
        g = synthetic.random_cloud(P, seed)             # world_scale: isotropic, log-uniform in [5e-3, 5e-2]
    else:
        g = synthetic.humanoid_layers(P, seed, layers=10) if cfg_name == "c5" else synthetic.humanoid(P, seed, order=order)
    if cfg_name == "c4":
        g["position"] = np.clip(g["position"], -1.0, 1.0)      # decode-path positions are clamped to [-1,1]^3
    cov = synthetic.covariance_from_gaussians(g)       # host stand-in for distCUDA2 + get_covariance (gs.py:70-73), untimed
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return dict(means3D=t(g["position"]), cov3D=t(cov), opacity=t(g["opacity"].reshape(P, 1)), rgb=t(g["rgb"])), g, cov


class _SclkSampler:
    """Shader clock of the GPU during the timed region, sampled by a side process (pinned off the step's cores) that reads the amdgpu sysfs
    node every 10 ms -- no HIP call, no GIL shared with the step.  stop() -> {"min", "median", "max", "samples", "source"} in MHz, or a note."""
    _SRC = r'''
import glob, os, re, sys, time
idx = int(sys.argv[1])
try:
    os.sched_setaffinity(0, set(os.sched_getaffinity(0)) - set(range(0, 4)) or os.sched_getaffinity(0))
except Exception:
    pass
cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"), key=lambda p: int(re.search(r"card(\d+)", p).group(1)))
path = cards[idx] if idx < len(cards) else (cards[0] if cards else None)
print(path or "none", flush=True)
while path:
    try:
        for ln in open(path):
            if "*" in ln:
                m = re.search(r"(\d+)\s*[Mm][Hh]z", ln)
                if m: print(m.group(1), flush=True)
    except Exception as e:
        print("err", e, flush=True); break
    time.sleep(0.01)
'''

    def __init__(self, local_rank):
        self.p = None
        try:
            self.p = subprocess.Popen([sys.executable, "-c", self._SRC, str(local_rank)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                                      preexec_fn=(lambda: os.sched_setaffinity(0, _ORIG_AFFINITY)) if _ORIG_AFFINITY else None)
            self.src = self.p.stdout.readline().strip()          # the sampler is up (first line: the node it reads)
        except Exception as e:      # noqa: BLE001
            self.src = f"unavailable ({e})"

    def stop(self):
        if self.p is None:
            return {"note": self.src}
        self.p.terminate()
        try:
            out, _ = self.p.communicate(timeout=5)
        except Exception:      # noqa: BLE001
            self.p.kill()
            out = ""
        v = [int(x) for x in out.split() if x.isdigit()]
        if not v:
            return {"note": f"no samples from {self.src}"}
        return {"min": int(np.min(v)), "median": int(np.median(v)), "max": int(np.max(v)), "samples": len(v), "source": self.src}


def fail(msg: str, code: int = 2):
    print(f"[bench] ERROR: {msg}", file=sys.stderr, flush=True)
    sys.exit(code)


def main(args):
    # stdout carries exactly ONE line (the JSON): libraries that print to fd 1 (gloo's connection banner, RCCL with NCCL_DEBUG set)
    # go to stderr for the length of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    cfg = dict(CONFIGS[args.config])
    steps = args.steps if args.steps is not None else cfg["steps"]
    warmup = args.warmup if args.warmup is not None else cfg["warmup"]
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("SIGMAN_BENCH_BACKEND", "nccl")
    if not torch.cuda.is_available():
        fail("bench.py needs an MI355X (there is no CPU fallback for the product path)")
    if world_env != args.gpus:
        fail(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world_env} ranks; refusing to report a different rank count")
    ndev = torch.cuda.device_count()
    if world_env > ndev and backend == "nccl":
        fail(f"{world_env} ranks but only {ndev} visible GPU(s): RCCL needs one GPU per rank (SIGMAN_BENCH_BACKEND=gloo shares cuda:0 for a plumbing check)")
    local_rank %= max(ndev, 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    world = 1
    # SIGMAN_BENCH_FORCE_PG=1: build the process group even for one rank (under a launcher), so that the RCCL code path of the N > 1 runs
    # -- init with device_id, async all-reduce, barriers -- can be exercised on a 1-GPU box
    dist_on = world_env > 1 or (os.environ.get("SIGMAN_BENCH_FORCE_PG") == "1" and "RANK" in os.environ)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
        world = dist.get_world_size()
        if world != args.gpus:
            fail(f"process group has {world} ranks, --gpus {args.gpus}")

    P = args.gaussians or cfg["P"]
    H = W = args.size or cfg["size"]
    S = cfg["subjects"]
    bwd, da = cfg["bwd"], cfg["da"]
    # ---- the view slots of this rank
    if args.config in ("c1", "c2", "c5"):
        vps = args.views_per_step
        all_views = [VIEWS[i % len(VIEWS)] for i in range(world * vps)]            # weak: one more view per extra GPU
    elif args.config == "c3":
        all_views = list(VIEWS)                                                    # per subject; strong
    else:
        all_views = list(range(90))
    mine = [all_views[i] for i in parallel.shard_views(len(all_views), rank, world)]
    n_total_views = S * len(all_views)
    n_local = S * len(mine)
    seeds = {"c1": [0], "c2": [1], "c3": [100 + b for b in range(S)], "c4": [3], "c5": [4]}[args.config]
    subs = [build_subject(args.config, P, s, dev, args.order) for s in seeds]
    subj = {k: torch.stack([x[0][k] for x in subs]) for k in ("means3D", "cov3D", "opacity", "rgb")}      # [S,P,...]
    g_host, cov_host = subs[0][1], subs[0][2]
    bg = torch.ones(3, device=dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    st = None
    if n_local:
        cv, cvp, cp = cameras.make_cameras(mine * S)                               # slot v -> subject v // len(mine)
        st = R.BatchedRasterizationSettings(H, W, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, bg, 0.5, t(cv), t(cvp), 0, t(cp), len(mine))
        if da:
            st = st._replace(depth_alpha_grads=True)       # C5 differentiates depth and alpha: their checkpoints are written by the forward
        if not args.exact_sync:
            # sync-free mode: size the binning buffers from one exact (untimed) forward, +25 % head-room; an overflow would raise
            with torch.no_grad():
                probe = R.forward_debug(subj["means3D"], subj["opacity"], colors_precomp=subj["rgb"], cov3D_precomp=subj["cov3D"], settings=st)
            st = st._replace(max_rendered=int(probe["num_rendered"] * 1.25) + 4096)
            del probe
    norm = 1.0 / (n_total_views * 3 * H * W)

    gt = gD = gA = gt_mask = None
    if bwd and n_local:
        # ground truth: render of a perturbed copy (untimed); stand-in for the dataset image of whole_loss.py:126-131
        with torch.no_grad():
            rng = torch.Generator(device="cpu").manual_seed(1234)
            pert = lambda x, s: x + s * torch.randn(x.shape, generator=rng).to(dev)
            gt_all = R.rasterize_gaussians_batched(pert(subj["means3D"], 2e-3), None, None, subj["rgb"] * 0.9, subj["opacity"], None, None,
                                                   subj["cov3D"], st)
            gt = gt_all[0].clamp(0, 1)
            # the reference's loss mask (the subject's matte; whole_loss.py:126-131 multiplies prediction AND target by it)
            gt_mask = None if args.no_mask else (gt_all[3] > 0.5).to(torch.float32).contiguous()
            del gt_all
        if da:
            gen = torch.Generator(device="cpu").manual_seed(77)
            gD = (torch.randn(n_local, 1, H, W, generator=gen) * norm).to(dev)
            gA = (torch.randn(n_local, 1, H, W, generator=gen) * norm).to(dev)
    one = torch.ones((), device=dev)

    def render_loss(means3D, cov3D, opacity, rgb, _views=None):
        # gs.py:98-107 rasterize + clamp, whole_loss.py:126-131 L1: one autograd node, loss kernel right behind the compositing kernel
        sp = lambda x, k: x.reshape(S, P, k)
        out = R.rasterize_l1_loss_batched(sp(means3D, 3), None, None, sp(rgb, 3), sp(opacity, 1), None, None, sp(cov3D, 6), st, gt, gt_mask, norm)
        return out if da else out[0]

    leaves = {k: v.clone().requires_grad_(True) for k, v in subj.items()}
    packed = parallel.pack_attributes(subj["means3D"].reshape(S * P, 3), subj["cov3D"].reshape(S * P, 6), subj["opacity"].reshape(S * P),
                                      subj["rgb"].reshape(S * P, 3))

    def backward_of(out):
        if da:      # C5: non-zero dL/ddepth and dL/dalpha go straight into the rasterizer node (no reduction kernels in the step)
            torch.autograd.backward([out[0], out[4], out[5]], [one, gD, gA])
            return out[0]
        out.backward(one)                 # explicit seed: autograd would otherwise launch a ones_like fill kernel every step
        return out

    pending = []          # N > 1, exchange="loss": the previous step's loss all-reduce

    # ---- exchange="full-pipelined": one attribute pack, one settings tuple (with its own sync-free capacity) and one loss function per chunk
    pipe = None
    if dist_on and bwd and args.exchange == "full-pipelined" and n_local:
        K = args.pipeline_chunks or S
        if S % K:
            fail(f"--pipeline-chunks {K} does not divide the {S} subjects of {args.config}")
        Sc = S // K
        cvc, cvpc, cpc = cameras.make_cameras(mine * Sc)
        chunk_packs, chunk_fns = [], []
        for c in range(K):
            sl = slice(c * Sc, (c + 1) * Sc)
            st_c = R.BatchedRasterizationSettings(H, W, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, bg, 0.5, t(cvc), t(cvpc), 0, t(cpc), len(mine))
            if da:
                st_c = st_c._replace(depth_alpha_grads=True)
            if not args.exact_sync:
                with torch.no_grad():
                    probe = R.forward_debug(subj["means3D"][sl], subj["opacity"][sl], colors_precomp=subj["rgb"][sl], cov3D_precomp=subj["cov3D"][sl], settings=st_c)
                st_c = st_c._replace(max_rendered=int(probe["num_rendered"] * 1.25) + 4096)
                del probe
            gt_c = gt[c * Sc * len(mine): (c + 1) * Sc * len(mine)]
            gD_c = None if gD is None else gD[c * Sc * len(mine): (c + 1) * Sc * len(mine)]
            gA_c = None if gA is None else gA[c * Sc * len(mine): (c + 1) * Sc * len(mine)]
            gm_c = None if gt_mask is None else gt_mask[c * Sc * len(mine): (c + 1) * Sc * len(mine)]
            chunk_packs.append(parallel.pack_attributes(subj["means3D"][sl].reshape(Sc * P, 3), subj["cov3D"][sl].reshape(Sc * P, 6),
                                                        subj["opacity"][sl].reshape(Sc * P), subj["rgb"][sl].reshape(Sc * P, 3)))

            def fn(m, cv_, o, r, _views, st_c=st_c, gt_c=gt_c, gD_c=gD_c, gA_c=gA_c, gm_c=gm_c):
                spc = lambda x, k: x.reshape(Sc, P, k)
                out = R.rasterize_l1_loss_batched(spc(m, 3), None, None, spc(r, 3), spc(o, 1), None, None, spc(cv_, 6), st_c, gt_c, gm_c, norm)
                return (out[0], [out[4], out[5]], [gD_c, gA_c]) if da else out[0]
            chunk_fns.append(fn)
        pipe = (chunk_packs, chunk_fns)

    def step():
        if not bwd:
            with torch.no_grad():
                if n_local:
                    R.rasterize_gaussians_batched(subj["means3D"], None, None, subj["rgb"], subj["opacity"], None, None, subj["cov3D"], st)
            return None
        if not dist_on:
            for v in leaves.values():
                v.grad = None
            return backward_of(render_loss(leaves["means3D"], leaves["cov3D"], leaves["opacity"], leaves["rgb"]))
        def rl(m, c, o, r, mv):
            out = render_loss(m, c, o, r, mv)
            return (out[0], [out[4], out[5]], [gD, gA]) if da else out     # C5: dL/ddepth, dL/dalpha seeded with the loss in one backward
        ex = args.exchange
        if ex == "loss":
            # the loss value is only read after the timed region: its all-reduce is waited for one step later (it overlaps the backward AND the
            # next forward; the 4-byte collective's latency never sits between two steps)
            loss, _grad, work = parallel.view_parallel_step(packed, all_views, rl, exchange="loss", seed_grad=one, pack_grad=False, wait=False)
            if pending and pending[0] is not None:
                pending[0].wait()
            pending[:] = [work]
            return loss
        if ex == "full-pipelined" and pipe is not None:
            losses, _grads = parallel.view_parallel_subjects(pipe[0], all_views, pipe[1], seed_grad=one, pipeline=True)
            return losses.sum()
        loss, _grad = parallel.view_parallel_step(packed, all_views, rl, exchange="full" if ex == "full-pipelined" else ex, seed_grad=one, pack_grad=False)
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
        if pending and pending[0] is not None:
            pending[0].wait()
            pending[:] = []
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- warmup (untimed), with a full per-kernel profile of the last 3 warmup steps to pick the dominant kernel
    prof_from = max(warmup - 3, min(1, warmup - 1), 0)      # (never the very first step: code-object loads would be billed to its kernels)
    for i in range(warmup):
        if i == prof_from:
            torch.cuda.synchronize()
            L.sgr_prof_configure(0xFFFF)
        step()
    torch.cuda.synchronize()
    prof_all = collect()
    nprof = max(warmup - prof_from, 1)
    breakdown = {KERNELS[k]: round(v[0] / nprof, 4) for k, v in prof_all.items() if k in KERNELS}
    cand = {k: v for k, v in prof_all.items() if k in KERNELS}
    dominant = max(cand, key=lambda k: cand[k][0]) if cand else 5
    L.sgr_prof_configure(0)
    # re-warm without the profiler: the profiled steps above synchronise after every kernel and let the clocks drop, so run (untimed)
    # until the GPU has been busy for ~50 ms again -- three steps are 0.5 ms at C2.  The count is the same on every rank.
    sync_all()
    t_rw = time.perf_counter()
    for _ in range(3):
        step()
    sync_all()
    t3 = time.perf_counter() - t_rw
    if dist_on:
        tr = torch.tensor([t3], device=dev, dtype=torch.float64)
        dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        t3 = float(tr.item())
    for _ in range(min(2000, int(0.05 / max(t3 / 3.0, 1e-5)))):
        step()
    # a timed region under 300 ms is noise, not a measurement (20 steps of C2 are 3 ms; 20-ms windows of one box within a minute: 0.131 .. 0.159 ms per step;
    # 100-ms regions: 0.1307 .. 0.1474 with the repeat windows behind them at 0.131 -- the slow phases last 10-50 ms and are neither the host's garbage
    # collector nor its run-ahead): time as many steps as 300 ms hold then, and say so
    steps_requested = steps
    t_step = max(t3 / 3.0, 1e-6)
    if steps * t_step < 0.300:
        steps = max(steps, int(np.ceil(0.300 / t_step)), 100)
        if dist_on:                                   # the same count on every rank
            ts = torch.tensor([steps], device=dev, dtype=torch.int64)
            dist.all_reduce(ts, op=dist.ReduceOp.MAX)
            steps = int(ts.item())
    # ---- where the two host threads of the step (Python + autograd engine) run: the 4-core set chosen at start-up, a neighbouring set,
    # or wherever the scheduler puts them -- whichever drives the step fastest on THIS box (a short untimed trial each: on some hosts the
    # first cores carry the interrupts / the launcher's own threads and the pinned step is 30 % slower than the free one; on most the
    # free one wanders across CCXs).  The choice and the trial times are in the line (config.host_threads).
    pin_report = _choose_host_cores(step, sync_all, t_step)

    # ---- timed region (no per-kernel events here)
    # The fused node's backward normally WAITS for its own forward's instance count (it arrives ~15 us into that forward's kernels): the host can
    # then never be more than one step ahead, and a host hiccup longer than the ~30 us of slack a 133-us step leaves becomes a bubble on the GPU
    # (timed regions of 0.131 .. 0.157 ms on one box within a minute, the slow ones with the GPU idle between steps: profiles/r05_count_wait_ab.txt).
    # "lazy:N": the backward only looks; a count that has not arrived yet is waited for N + 1 forwards later (csrc/torch_node.cpp,
    # set_count_wait) -- the overflow check still happens for every step, at most N steps later, and once more behind the timed region
    # (check_pending_overflows below).  SIGMAN_COUNT_WAIT=own times the library default.
    lazy_counts = False
    _node = _cabi.torch_node()
    _cw = os.environ.get("SIGMAN_COUNT_WAIT", _COUNT_WAIT_DEFAULT)
    if _node is not None and not args.exact_sync and _cw != "own":
        R.set_count_wait(_cw)
        lazy_counts = _cw
        for _ in range(3):
            step()
    # no cyclic garbage collection inside the timed region: every step creates a few hundred Python objects, and a generation-2 pass that
    # happens to fall into a 20-ms window costs it milliseconds (regions of 0.131 and 0.159 ms within one minute on one box)
    import gc
    gc.collect()
    gc.disable()
    # (the collection above hands cached blocks back to torch's allocator and the host-core trial may have moved the threads: a few untimed
    # steps in the final state -- without them the first region read ~1 us per step above every repeat window behind it, on every box)
    for _ in range(max(3, min(warmup, 50))):
        step()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # (the library launches on torch's current stream)

    def clock_probe():
        # the shader clock a wave really runs at, from the chip's own counters (sgr_clock_probe: ~0.5 ms, outside the timed region)
        import ctypes as _C
        mhz = _C.c_double(0.0)
        rc = L.sgr_clock_probe(_C.byref(mhz), _C.c_void_p(torch.cuda.current_stream().cuda_stream))
        return round(mhz.value, 1) if rc == 0 else None
    sync_all()
    clk_before = clock_probe()
    sync_all()
    t0 = time.perf_counter()
    ev0.record()
    loss = None
    dbg_marks = []
    for k_ in range(steps):
        loss = step()
        if _DEBUG_BLOCKS and (k_ + 1) % max(steps // 8, 1) == 0:
            dbg_marks.append(time.perf_counter() - t0)           # (host time stamps only: no synchronisation inside the timed region)
    ev1.record()
    t_issued = time.perf_counter() - t0                          # the host has handed over the last step; what the GPU still holds drains below
    sync_all()
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    gpu_elapsed = ev0.elapsed_time(ev1) * 1e-3
    clk_after = clock_probe()
    # ---- the same K steps `--windows` more times: spread of wall and event time per step (not the headline: that is the region above)
    # The shader clock is sampled HERE, over the repeat windows, not over the headline region: the side process (another core, no GIL, no HIP)
    # reads an amdgpu sysfs node every 10 ms, each read is a message to the SMU, and with it running the region read 0.4-0.9 us per step above
    # the windows behind it (four alternating runs on one box; ~0.3 without).
    sclk = _SclkSampler(local_rank) if (rank == 0 and not args.no_sclk and args.windows > 0) else None
    win_wall, win_gpu, win_drain = [], [], []
    for _w in range(max(args.windows, 0)):
        sync_all()
        tw = time.perf_counter()
        ev0.record()
        for _ in range(steps):
            step()
        ev1.record()
        tq = time.perf_counter()
        sync_all()
        win_drain.append((time.perf_counter() - tq) * 1e3)
        win_wall.append((time.perf_counter() - tw) / steps * 1e3)
        win_gpu.append(ev0.elapsed_time(ev1) / steps)
    sclk_report = sclk.stop() if sclk is not None else None
    if sclk_report is not None and "min" in sclk_report:
        sclk_report["sampled_over"] = "the repeat windows behind the timed region"
    gc.enable()
    R.check_pending_overflows(True)                      # every step's instance count has been looked at (raises if one did not fit)
    if _DEBUG_BLOCKS:
        print("[bench] host time stamps inside the timed region (s):", [round(x, 5) for x in dbg_marks], "elapsed", round(elapsed, 5), "affinity now", len(os.sched_getaffinity(0)), file=sys.stderr)
    if dist_on:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # ---- the same K steps again with HIP-event pairs around the dominant kernel (live roofline measurement)
    L.sgr_prof_configure(1 << dominant)
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dom = collect()
    L.sgr_prof_configure(0)

    # ---- workload counters (from the run's own buffers)
    Rn = S_visits = R_visited = 0
    if n_local:
        with torch.no_grad():
            dbg = R.forward_debug(subj["means3D"], subj["opacity"], colors_precomp=subj["rgb"], cov3D_precomp=subj["cov3D"],
                                  settings=st._replace(max_rendered=0))
            Rn = int(dbg["num_rendered"])
            S_visits = int(dbg["n_contrib"].to(torch.int64).sum().item())
            # tile instances the BACKWARD can visit: per tile, the list entries up to the last contributor of any of its pixels (the walk of a
            # tile ends there; at C5 -- deep lists behind an opaque front -- that is a third of num_rendered).  SURVEY 8d prices B1 at 88 B per
            # tile instance: per VISITED instance, or a table prints counter traffic below the algorithmic bytes
            nc = dbg["n_contrib"].reshape(-1, H, W).to(torch.int64)
            Ty_, Tx_ = (H + 15) // 16, (W + 15) // 16
            ncp = torch.zeros(nc.shape[0], Ty_ * 16, Tx_ * 16, dtype=torch.int64, device=nc.device)
            ncp[:, :H, :W] = nc
            R_visited = int(ncp.reshape(-1, Ty_, 16, Tx_, 16).amax(dim=(2, 4)).sum().item())
            del dbg, nc, ncp
    ms_per_step = elapsed / steps * 1e3
    views_per_s = n_total_views / (ms_per_step * 1e-3)
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    dom_ms, dom_n = dom.get(dominant, (0.0, 0))
    dom_avg_ms = dom_ms / max(dom_n, 1)
    abytes = algorithmic_bytes(dominant, P * n_local, R_visited if dominant == 6 else Rn, H * W * n_local, tiles * n_local, gt_mask is not None)     # (B1: per instance within the backward's reach)
    # the fused single-view step (one or two views through the C++ node, SIGMAN_FUSED_STEP != 0): no loss launch -- the compositing kernel also
    # reads the target (12 B per pixel, + 4 B of mask) and writes dL/dcolor (12 B); the colour it would have re-read (12 B) never leaves the chip
    fused_step = bool(bwd and breakdown and "clamped_l1" not in breakdown and "render_bwd" in breakdown)
    if fused_step and dominant == 5:
        abytes += (28 if gt_mask is not None else 24) * H * W * n_local
    achieved = abytes / (dom_avg_ms * 1e-3) / 1e9 if dom_avg_ms > 0 else 0.0
    step_bytes = ((212 if bwd else 104) * P * n_local + (176 if bwd else 88) * Rn + (52 if bwd else 24) * H * W * n_local + 8 * tiles * n_local)

    # HBM traffic of the dominant kernel from the committed rocprofv3 PMC summary of this same command (separate --pmc passes,
    # gfx950 FETCH_SIZE correction applied as MI355X_MICROARCH.md prescribes); null when no summary matches this workload
    traffic, traffic_src = None, None
    for rnd in ("r06", "r05", "r04", "r03", "r02"):               # the newest committed summary for this config
        try:
            path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_{args.config}.json")
            pmc = json.load(open(path))
            if pmc.get("P") == P and pmc.get("size") == H and pmc.get("view_slots") == n_local:
                traffic = pmc["kernels"].get(KERNELS.get(dominant, ""), {}).get("hbm_bytes_corrected")
                traffic_src = f"profiles/{rnd}_pmc_{args.config}.json (kernels of commit {pmc.get('commit', '?')})"
                break
        except Exception:
            traffic = None

    out = {
        "metric": f"rendered views/sec ({'fwd+bwd' if bwd else 'fwd'}) at {H}x{W}, {P} Gaussians/view",
        "value": round(views_per_s, 3), "unit": "views/s", "n_gpus": world, "steps": steps, "steps_requested": steps_requested, "warmup": warmup,
        "ms_per_step": round(ms_per_step, 4), "gpu_ms_per_step": round(gpu_elapsed / steps * 1e3, 4), "higher_is_better": True, "scaling": cfg["scaling"], "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{cfg['label']}, {P} Gaussians/subject, {S} subject(s) x {len(mine)} view(s) on this GPU per step, "
                               f"{H}x{W}, {'fwd+bwd' if bwd else 'forward only'}, colors_precomp+cov3D_precomp"
                               + ((", clamp + masked L1 loss (mask = ground-truth alpha > 0.5)" if gt_mask is not None else ", clamp+L1 loss") if bwd else "") + (", dL/ddepth and dL/dalpha non-zero" if da else ""),
                   "name": args.config, "gaussian_order": args.order,
                   "count_wait": (lazy_counts if lazy_counts else "own") + (" (the library default is \"own\": variants.count_wait_own_ms_per_step is the same step timed with it)" if lazy_counts else " (the library default)"),
                   "views_per_step_total": n_total_views, "view_slots_this_gpu": n_local,
                   "parallelism": (f"view-parallel x{world} ({backend}), exchange={'none (forward only)' if not bwd else args.exchange}" if dist_on else "single GPU"),
                   "fused_step": fused_step, "ranks_seen": world, "num_rendered_per_gpu": Rn, "tile_instances_within_reach_of_the_backward_per_gpu": R_visited, "gaussian_pixel_visits_per_gpu": S_visits,
                   "binning_buffers": "exact (D2H read of num_rendered per step)" if args.exact_sync else (f"pre-sized, max_rendered={st.max_rendered} (sync-free)" if st else "-"),
                   "count_check": "exact read" if args.exact_sync else (f"every step, by a later forward of the thread at the latest (set_count_wait {lazy_counts}: the host may run ahead of the GPU by that many steps + 1; once more behind the timed region)" if lazy_counts else "every step, by its own backward")},
        "gaussian_pixel_ops_per_s_per_gpu": round(S_visits / (ms_per_step * 1e-3), 1),
        "tile_instances_per_s_per_gpu": round(Rn / (ms_per_step * 1e-3), 1),
        "step_hbm": {"algorithmic_bytes_per_step_per_gpu": step_bytes, "achieved_GBps": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                     "frac_of_peak": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
        "roofline": {"bound": "hbm", "kernel": KERNELS.get(dominant, str(dominant)), "achieved": round(achieved, 2),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "traffic_source": traffic_src, "algorithmic_bytes_per_launch": abytes, "avg_launch_ms": round(dom_avg_ms, 5), "launches": dom_n},
        "kernel_ms_per_step": breakdown,
        "loss": None if loss is None else float(loss.detach()),
    }
    # The compositing kernels are instruction-bound, not byte-bound: next to the HBM figure the contract asks for, the roof
    # they really approach -- plain wave64 VALU instructions issued per second against the rate measured on this part with
    # tools/micro/valu_rate.hip (8.55e11/s) -- from the committed SQ-counter summary of this same command (tools/pmc_sq.sh), if there is one
    try:
        sq_path = next(q for q in (os.path.join(ROOT, "profiles", f"{r}_sq_{args.config}.json") for r in ("r06", "r05", "r04", "r03")) if os.path.exists(q))
        sq = json.load(open(sq_path))
        kname = KERNELS.get(dominant, "")
        ent = next((v for k, v in sq["kernels"].items() if k.startswith(kname) and "SQ_INSTS_VALU" in v and v.get("avg_us")), None)
        if ent:
            rate = ent["SQ_INSTS_VALU"] / (ent["avg_us"] * 1e-6)
            out["roofline_valu_issue"] = {"bound": "valu_issue", "kernel": kname, "achieved": round(rate / 1e9, 1), "peak": 855.0, "unit": "G wave-instr/s",
                                          "frac": round(rate / 8.55e11, 4), "source": f"profiles/{os.path.basename(sq_path)} (SQ_INSTS_VALU / kernel duration of that run)"}
    except Exception:      # noqa: BLE001
        pass
    out["config"]["host_threads"] = pin_report
    # Who limited the timed region: when the host has issued its last step, the work the GPU still holds takes `queue_drain_steps` steps' worth of
    # time to finish.  Many steps = the host ran ahead and the GPU set the pace; ~1 or less = the GPU was waiting for the host (a host-bound
    # region: the 0.14-0.15-ms readings of some runs on this pool's shared hosts are of that kind -- DESIGN.md 7).
    out["host_queue"] = {"issue_ms_per_step": round(t_issued / steps * 1e3, 4), "queue_drain_steps": round((elapsed_local - t_issued) / (elapsed_local / steps), 2),
                         "note": "queue_drain_steps ~<= 1: the region was host-bound (the GPU idled between steps); >> 1: GPU-bound"}
    if win_wall:
        q = lambda v: [round(float(x), 4) for x in (np.min(v), np.median(v), np.max(v))]
        out["windows"] = {"n": len(win_wall), "steps_each": steps, "wall_ms_per_step_min_median_max": q(win_wall), "gpu_ms_per_step_min_median_max": q(win_gpu),
                          "queue_drain_steps_min_median_max": q([d / w for d, w in zip(win_drain, win_wall)]),
                          "note": "repeats of the timed region behind it (this rank); the headline is the first region, not their minimum"}
    if sclk_report is not None:
        out["sclk_mhz"] = sclk_report
    out["sclk_mhz_probe"] = {"before_timed_region": clk_before, "behind_timed_region": clk_after,
                             "how": "sgr_clock_probe: one wave's cycle counter against the chip's constant 100-MHz counter over ~0.5 ms"}
    if rank == 0 and world == 1:
        out["frontend_ms_per_subject"] = frontend_ms(g_host, dev)
        if not args.no_variants and args.config in ("c2", "c3"):
            out["variants"] = variants(args, subj, st, gt, norm, S, P, H, W, mine, dev, subs)
        if not args.no_cpu_baseline:
            if _ORIG_AFFINITY is not None:
                os.sched_setaffinity(0, _ORIG_AFFINITY)      # the OpenMP oracle gets every host core
            out["cpu_baseline"] = cpu_baseline(g_host, cov_host, mine[0], H, W, None if gt is None else gt[0].cpu().numpy(), norm, bwd, da)
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


def frontend_ms(g_host, dev):
    """distCUDA2 + get_covariance (gs.py:70-73) for one subject: what GaussianRenderer.render pays once per subject before the
    rasterizer; outside the timed step (the metric is the rasterizer's), reported next to it."""
    from sigman_release_amd.renderer import covariance_from_scale_rotation, dist_cuda2
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    pos, sc, rot = t(g_host["position"]), t(g_host["scale"]), t(g_host["cov3d"])

    def run():
        with torch.no_grad():
            covariance_from_scale_rotation(sc[None], rot[None], dist_cuda2(pos)[None])
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / 10 * 1e3, 4)


def variants(args, subj, st, gt, norm, S, P, H, W, mine, dev, subs):
    """The same workload under the conditions the headline does NOT assume (N=1 only):
      renderer_render_*    GaussianRenderer.render (gs.py:49-117: distCUDA2 + get_covariance + all B*V views + clamp) forward and backward from
                           the `gaussians` dict (position / scale / rotation / opacity / rgb leaves), then the headline's fused clamp+L1 loss kernel
      auto_capacity_*      the batched rasterizer in automatic capacity mode (max_rendered = -1) + the same loss: no caller-supplied capacity
      per_view_loop_*      the reference's own call pattern (gs.py:62-109): Python loop over subjects and views through the
                           upstream-signature GaussianRasterizer (one launch chain + one autograd node per view), then clamp/stack/L1
      unpinned_*, exact_sync_*   the batched step re-run in a subprocess without host-thread pinning / with upstream's blocking read of
                           num_rendered in every forward (--exact-sync) instead of the pre-sized sync-free buffers"""
    from sigman_release_amd.losses import clamped_l1_loss
    out = {}
    one = torch.ones((), device=dev)
    leaves = {k: v.clone().requires_grad_(True) for k, v in subj.items()}
    vm, pm, cp = st.viewmatrix, st.projmatrix, st.campos
    V = len(mine)
    # ---- what the reference's render() pays: the front end (3-NN, covariance) inside the step, leaves = the decoder's outputs
    from types import SimpleNamespace
    from sigman_release_amd.renderer import GaussianRenderer
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    gd = {k: torch.stack([t(x[1][k]) for x in subs]).requires_grad_(True) for k in ("position", "opacity", "scale", "cov3d", "rgb")}
    rend = GaussianRenderer(SimpleNamespace(FoVy=2.0 * float(np.arctan(cameras.TAN_HALF_FOV)), output_size_h=H, output_size_w=W), device=dev)
    cvw, cvp_, cps = vm.reshape(S, V, 4, 4), pm.reshape(S, V, 4, 4), cp.reshape(S, V, 3)

    def renderer_step():
        for v in gd.values():
            v.grad = None
        img = rend.render(gd, cvw, cvp_, cps, bg_color=st.bg)["image"].reshape(S * V, 3, H, W)
        clamped_l1_loss(img, gt, None, norm).backward(one)          # the headline's loss kernel (clamping the clamped image again changes nothing)
    n = 50 if S * V == 1 else 8
    for _ in range(3):
        renderer_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        renderer_step()
    torch.cuda.synchronize()
    out["renderer_render_ms_per_step"] = round((time.perf_counter() - t0) / n * 1e3, 4)

    # ---- the batched step in AUTOMATIC capacity mode (max_rendered = -1: what render() and every caller that does not size the binning
    # buffers itself gets): sync-free with the capacity learned on the first call, count checked inside the call
    st_auto = st._replace(max_rendered=-1)

    def auto_step():
        for v in leaves.values():
            v.grad = None
        color = R.rasterize_gaussians_batched(leaves["means3D"], None, None, leaves["rgb"], leaves["opacity"], None, None, leaves["cov3D"], st_auto)[0]
        clamped_l1_loss(color, gt, None, norm).backward(one)
    for _ in range(5):
        auto_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        auto_step()
    torch.cuda.synchronize()
    out["auto_capacity_ms_per_step"] = round((time.perf_counter() - t0) / n * 1e3, 4)

    def per_view():
        for v in leaves.values():
            v.grad = None
        imgs = []
        for b in range(S):
            for v in range(V):
                i = b * V + v
                rs = R.GaussianRasterizationSettings(H, W, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, st.bg, 0.5, vm[i], pm[i], 0, cp[i], False, False)
                img, _radii, _depth, _alpha = R.GaussianRasterizer(rs)(
                    means3D=leaves["means3D"][b], means2D=torch.zeros_like(leaves["means3D"][b]), opacities=leaves["opacity"][b],
                    colors_precomp=leaves["rgb"][b], cov3D_precomp=leaves["cov3D"][b])
                imgs.append(img.clamp(0, 1))
        clamped_l1_loss(torch.stack(imgs), gt, None, norm).backward()
    n = 30 if S * V == 1 else 5
    for _ in range(3):
        per_view()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        per_view()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    out["per_view_loop_ms_per_step"] = round(dt * 1e3, 4)
    out["per_view_loop_views_per_s"] = round(S * V / dt, 1)
    base = [sys.executable, os.path.abspath(__file__), "--config", args.config, "--no-cpu-baseline", "--no-variants"]
    # count_wait_own: the same step with the LIBRARY DEFAULT count wait ("own": every backward waits for its own forward's instance count, the host
    # is never more than one step ahead) instead of the headline's set_count_wait("lazy:4"); order_template: the subjects' Gaussians in a
    # spatially coherent order (--order template) instead of a random permutation
    for key, env, extra in (("unpinned", {"SIGMAN_NO_PIN": "1"}, []), ("exact_sync", {}, ["--exact-sync"]), ("count_wait_own", {"SIGMAN_COUNT_WAIT": "own"}, []),
                            ("order_template", {}, ["--order", "template"])):
        try:
            if _ORIG_AFFINITY is not None:
                os.sched_setaffinity(0, _ORIG_AFFINITY)
            r = subprocess.run(base + extra, capture_output=True, text=True, timeout=600, env={**os.environ, **env})
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
            out[f"{key}_ms_per_step"] = json.loads(line)["ms_per_step"]
            if key == "order_template":
                out["order_template_kernel_ms_per_step"] = json.loads(line).get("kernel_ms_per_step")
        except Exception as e:      # noqa: BLE001  (a variant that cannot run is reported, not fatal)
            out[f"{key}_ms_per_step"] = f"failed: {e}"
    return out


def cpu_baseline(g_host, cov_host, view, H, W, gt, norm, bwd, da):
    """CPU oracle (test infrastructure) on the same inputs; bounded sample (ONE view slot of the workload, 3 repetitions),
    all host cores via OpenMP."""
    from oracle import ref
    cv, cvp, cp = cameras.make_cameras([view])
    P = g_host["position"].shape[0]
    kw = dict(viewmatrix=cv[0], projmatrix=cvp[0], campos=cp[0], bg=np.ones(3, np.float32), tanfovx=cameras.TAN_HALF_FOV,
              tanfovy=cameras.TAN_HALF_FOV, image_height=H, image_width=W)
    rng = np.random.default_rng(77)
    times = []
    for i in range(4):
        t0 = time.perf_counter()
        st = ref.forward(g_host["position"], g_host["opacity"].reshape(P), colors_precomp=g_host["rgb"], cov3D_precomp=cov_host, **kw)
        if bwd:
            img = np.clip(st.color, 0, 1)
            loss_cpu = float(np.abs(img - gt).sum() * norm)   # noqa: F841  (same clamp + L1 epilogue as the GPU step)
            gC = (np.sign(img - gt) * ((st.color >= 0) & (st.color <= 1)) * norm).astype(np.float32)
            if da:
                ref.backward(st, gC, (rng.normal(size=(1, H, W)) * norm).astype(np.float32), (rng.normal(size=(1, H, W)) * norm).astype(np.float32))
            else:
                ref.backward(st, gC)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times[1:]))
    return {"value": round(1.0 / med, 4), "unit": "views/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"median of 3 x (1 view slot of this workload, {'fwd+bwd' if bwd else 'forward only'}, {P} Gaussians, {H}x{W}) after 1 warm-up; "
                      "oracle/gsplat_ref.c with OpenMP on all host cores",
            "seconds_per_view": round(med, 4)}


if __name__ == "__main__":
    main(_ARGS)
