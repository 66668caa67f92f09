"""View-parallel multi-GPU mode: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference is subject-parallel DDP only (configs/training.yaml): every rank renders its own
B*V views serially (gs.py:62,75).  Views are independent units, so this module shards the V views of
a subject one-view-per-GPU and adds exactly the exchange steps that sharding needs:
  1. broadcast of the packed Gaussian attributes, one flat [13*P] buffer (means | cov3D | opacity | rgb), from the producer rank
  2. all-reduce(sum) of the image-space loss scalar(s)       (the collective BASELINE.json's north_star names)
  3. all-reduce(sum) of the packed attribute gradients [13*P] (needed for training parity: only then is the
     gradient, not just the loss value, global)
xGMI is point-to-point (7 links x ~153 GB/s); 5.2 MB at P=1e5 is latency-bound (~60 us ring time), so the payload is
kept as ONE contiguous tensor = one collective each way: the loss scalar rides in the last element of the gradient
buffer ([13*P + 1]), so 2 and 3 are a single all-reduce.

Two exchange protocols (view_parallel_step(exchange=...)):
  "loss"  exactly what BASELINE.json's north_star names: the attributes are already replicated (in a DDP job every rank runs the
          same decoder on the same subject), each rank renders its views, and ONLY the image-space loss is all-reduced -- issued
          right after the forward and overlapped with the backward.  The returned gradient is the rank's partial sum over its own
          views; it flows into the rank's decoder replica, whose parameter gradients DDP reduces anyway.
  "full"  for callers without a replicated producer: steps 1 + 2 + 3 above (global loss AND global attribute gradient on every
          rank).

Several subjects per step (the reference's batch of 8, configs/training.yaml) in the "full" protocol: `view_parallel_subjects` runs the
subjects (or chunks of subjects) as a PIPELINE -- the broadcast of chunk c+1 travels while chunk c is rendered, the all-reduce of chunk c's
gradients while chunk c+1 is rendered -- instead of one blocking broadcast, the render, one blocking all-reduce.  Each chunk has
its own [n + 1] gradient buffer, so the collectives in flight never alias.  `all_gather_images` is the forward-only counterpart (the
reference's only cross-rank traffic on rendered images: core/loss/eval.py:81-82 gathers images_pred for the metrics).

The render function is injected, so the sharding/collective logic is testable on CPU with gloo (tests/test_parallel_cpu.py).
"""
from __future__ import annotations

import contextlib

from typing import Callable, Sequence

import os

import torch
import torch.distributed as dist

ATTR = 13  # mean 3 | cov3D 6 | opacity 1 | rgb 3



def _collectives(world: int) -> bool:
    """Are collectives issued?  Always for more than one rank; for ONE rank only when a process group exists and SIGMAN_FORCE_COLLECTIVES=1
    asks for them -- a 1-rank broadcast / all-reduce / all-gather is a valid RCCL operation on the backend's own stream, which is how the
    1-GPU test box runs every call (and every stream-ordering wait) of the N > 1 paths."""
    return world > 1 or (dist.is_initialized() and os.environ.get("SIGMAN_FORCE_COLLECTIVES") == "1")

def shard_views(n_views: int, rank: int, world: int) -> list[int]:
    """Views rendered by `rank`: {v : v mod world == rank} (C3: 8 views <-> 8 GPUs; C4: 90 views -> 11-12 per GPU)."""
    return [v for v in range(n_views) if v % world == rank]


def pack_attributes(means3D, cov3D, opacity, rgb) -> torch.Tensor:
    """One flat buffer [13*P] in struct-of-arrays order (means | cov3D | opacity | rgb): ONE collective moves everything and
    every attribute is a contiguous view of it (no unpack copies on the receiving ranks)."""
    P = means3D.shape[0]
    return torch.cat([means3D.reshape(P * 3), cov3D.reshape(P * 6), opacity.reshape(P), rgb.reshape(P * 3)]).contiguous()


def unpack_attributes(packed: torch.Tensor):
    P = packed.numel() // ATTR
    return (packed[: 3 * P].view(P, 3), packed[3 * P: 9 * P].view(P, 6), packed[9 * P: 10 * P].view(P, 1),
            packed[10 * P:].view(P, 3))


def _split(res):
    if isinstance(res, tuple):
        return res[0], list(res[1]), list(res[2])
    return res, [], []


def _backward(loss, seed_grad, extra_out, extra_grad):
    if extra_out:
        torch.autograd.backward([loss] + extra_out, [seed_grad if seed_grad is not None else torch.ones_like(loss)] + extra_grad)
    else:
        loss.backward(seed_grad)


@contextlib.contextmanager
def _fused_step(enabled: bool):
    """The calling thread's fused-single-view-step switch of the library for the length of a forward (thread-local, sgr_set_fused_step)."""
    if enabled:
        yield
        return
    try:
        from . import _cabi
        L = _cabi.lib()
        old = L.sgr_set_fused_step(0)
    except Exception:      # noqa: BLE001  (a render_loss that does not go through this library: nothing to switch)
        L = None
    try:
        yield
    finally:
        if L is not None:
            L.sgr_set_fused_step(old)


def view_parallel_step(packed: torch.Tensor, view_ids: Sequence[int], render_loss: Callable, *, src: int = 0, group=None,
                       broadcast: bool = True, exchange: str = "full", seed_grad: torch.Tensor = None, pack_grad: bool = True,
                       wait: bool = True):
    """One fwd+bwd step of a subject whose views are sharded over the ranks of `group`.

    packed      flat [13*P] attributes (pack_attributes); exchange="full": only rank `src` needs valid contents when broadcast=True
    view_ids    all views of the subject (same list on every rank)
    render_loss (means3D, cov3D, opacity, rgb, my_view_ids) -> scalar loss SUM over my views (differentiable), or a tuple
                (loss, extra_outputs, extra_grads): further outputs of the same graph with fixed upstream gradients (e.g. the
                rasterizer's depth / alpha maps with dL/ddepth, dL/dalpha) that are seeded together with the loss in ONE backward
    exchange    "full": returns (global loss, GLOBAL gradient flat [13*P]);  "loss": returns (global loss, this rank's PARTIAL
                gradient) with the loss all-reduce overlapped with the backward (module docstring)
    seed_grad   optional 0-d ones tensor for loss.backward() (saves the fill kernel autograd would launch for it)
    pack_grad   exchange="loss" only: False returns the partial gradient as the tuple (d_means3D, d_cov3D, d_opacity, d_rgb) of the
                autograd leaves instead of one concatenated [13*P] buffer (saves a copy kernel when the caller consumes them separately)
    wait        exchange="loss" only: False does not wait for the loss all-reduce: returns (loss tensor, grad, work) and the caller calls
                work.wait() before it READS the loss (a training loop only logs it) -- the collective then overlaps the next step as well,
                and the 4-byte all-reduce's latency never sits between two steps
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if exchange not in ("full", "loss"):
        raise ValueError("exchange must be 'full' or 'loss'")
    if exchange == "loss":
        leaves = [x.detach().requires_grad_(True) for x in unpack_attributes(packed)]
        mine = [view_ids[i] for i in shard_views(len(view_ids), rank, world)]
        work = None
        if mine:
            # The loss all-reduce below is meant to travel while the backward runs.  The fused single-view step (rasterize_l1_loss_batched on one or
            # two views: the forward call already queues the compositing backward, csrc/render.hip FusedL1) would leave it only the gather to hide
            # behind (12 of 45 us at C2) for the 2.5 us the fused step saves: with live collectives this rank's forward runs unfused -- unless the
            # caller does not wait for the loss here (wait=False): the all-reduce then has the whole next step to hide behind, and the rank keeps
            # the single-rank launch chain (a weak-scaling run compares exactly these two)
            with _fused_step(not _collectives(world) or not wait):
                loss, extra_out, extra_grad = _split(render_loss(*leaves, mine))
            loss_val = loss.detach().reshape(1).clone()
        else:
            loss, loss_val = None, torch.zeros(1, device=packed.device, dtype=packed.dtype)
        if _collectives(world):
            work = dist.all_reduce(loss_val, op=dist.ReduceOp.SUM, group=group, async_op=True)   # travels while the backward runs
        if loss is not None:
            _backward(loss, seed_grad, extra_out, extra_grad)
            grads = [(l.grad if l.grad is not None else torch.zeros_like(l)) for l in leaves]
        else:
            grads = [torch.zeros_like(l) for l in leaves]
        grad = torch.cat([g.reshape(-1) for g in grads]) if pack_grad else tuple(grads)
        if not wait:
            return loss_val[0], grad, work
        if work is not None:
            work.wait()
        return loss_val[0], grad
    if _collectives(world) and broadcast:
        dist.broadcast(packed, src=src, group=group)
    # the four attributes are contiguous views of the flat buffer: separate autograd leaves without any copy
    leaves = [x.detach().requires_grad_(True) for x in unpack_attributes(packed)]
    mine = [view_ids[i] for i in shard_views(len(view_ids), rank, world)]
    if mine:
        loss, extra_out, extra_grad = _split(render_loss(*leaves, mine))
        _backward(loss, seed_grad, extra_out, extra_grad)
        buf = torch.cat([(l.grad if l.grad is not None else torch.zeros_like(l)).reshape(-1) for l in leaves]
                        + [loss.detach().reshape(1).to(packed.dtype)])
    else:
        buf = torch.zeros(packed.numel() + 1, device=packed.device, dtype=packed.dtype)
    if _collectives(world):
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)     # gradients [13*P] + loss scalar: one collective
    return buf[-1], buf[:-1]


def view_parallel_subjects(chunks: Sequence[torch.Tensor], view_ids: Sequence[int], render_loss, *, srcs: Sequence[int] = None, group=None,
                           pipeline: bool = True, seed_grad: torch.Tensor = None):
    """exchange="full" for a step of several subjects, pipelined over chunks of subjects.

    chunks      flat attribute buffers, one per pipeline stage (one subject's [13*P] pack, or several subjects packed by the caller);
                chunk c needs valid contents on rank srcs[c] only (default: every chunk comes from rank 0)
    view_ids    the views of a subject (the same for every chunk); rank r renders {v : index(v) mod world = r} of every chunk
    render_loss callable (means3D, cov3D, opacity, rgb, my_view_ids) -> loss | (loss, extra_outputs, extra_grads) as in view_parallel_step,
                or a sequence of one callable per chunk (each with its own targets); the four attribute views are those of
                unpack_attributes(chunk): a caller that packed several subjects reshapes them itself
    pipeline    True: chunk c+1's broadcast is issued before chunk c is rendered and waited for after it; chunk c's gradient all-reduce is
                issued right behind its backward and waited for at the end.  With RCCL the collectives run on the backend's own stream
                (work.wait() makes the compute stream wait, never the host), i.e. the broadcast overlaps the previous chunk's kernels and the
                all-reduce the next chunk's; the first broadcast and the last all-reduce of a step stay exposed.
                False: the same collectives, blocking, in program order -- the same numbers bit for bit (tests).
    -> (losses [n_chunks] tensor, [global gradient of chunk c, flat like chunk c] list)
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = len(chunks)
    srcs = [0] * n if srcs is None else list(srcs)
    fns = list(render_loss) if isinstance(render_loss, (list, tuple)) else [render_loss] * n
    mine = [view_ids[i] for i in shard_views(len(view_ids), rank, world)]
    bwork = [None] * n
    rwork = [None] * n
    bufs = [None] * n

    def start_broadcast(c):
        if _collectives(world):
            bwork[c] = dist.broadcast(chunks[c], src=srcs[c], group=group, async_op=pipeline)

    if n:
        start_broadcast(0)
    for c in range(n):
        if pipeline and c + 1 < n:
            start_broadcast(c + 1)                       # travels while chunk c is rendered
        if bwork[c] is not None:
            bwork[c].wait()
        leaves = [x.detach().requires_grad_(True) for x in unpack_attributes(chunks[c])]
        if mine:
            loss, extra_out, extra_grad = _split(fns[c](*leaves, mine))
            _backward(loss, seed_grad, extra_out, extra_grad)
            bufs[c] = torch.cat([(l.grad if l.grad is not None else torch.zeros_like(l)).reshape(-1) for l in leaves]
                                + [loss.detach().reshape(1).to(chunks[c].dtype)])
        else:
            bufs[c] = torch.zeros(chunks[c].numel() + 1, device=chunks[c].device, dtype=chunks[c].dtype)
        if _collectives(world):
            rwork[c] = dist.all_reduce(bufs[c], op=dist.ReduceOp.SUM, group=group, async_op=pipeline)   # travels while chunk c+1 is rendered
        if not pipeline and c + 1 < n:
            start_broadcast(c + 1)
    for w in rwork:
        if w is not None:
            w.wait()
    if not n:
        return torch.zeros(0), []
    return torch.stack([b[-1] for b in bufs]), [b[:-1] for b in bufs]


def all_gather_images(local: torch.Tensor, n_views: int, group=None) -> torch.Tensor:
    """Forward-only / evaluation path: every rank rendered its shard {v : v mod world = rank} of n_views views (`local` [n_mine, ...] in
    that order); returns all views [n_views, ...] in view order on every rank (core/loss/eval.py:81-82 gathers the predicted images for
    PSNR / SSIM / LPIPS).  One collective: the shards are padded to ceil(n_views / world) views (90 views on 8 ranks: 11 or 12 each)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_mine = len(shard_views(n_views, rank, world))
    if local.shape[0] != n_mine:
        raise ValueError(f"rank {rank} of {world} holds {local.shape[0]} views, its shard of {n_views} views has {n_mine}")
    if not _collectives(world):
        return local
    per = (n_views + world - 1) // world
    pad = local if n_mine == per else torch.cat([local, local.new_zeros((per - n_mine,) + tuple(local.shape[1:]))])
    out = local.new_empty((world, per) + tuple(local.shape[1:]))
    # The collective is chosen UP FRONT from the backend -- every rank takes the same branch -- never by catching an error: a real failure
    # (an RCCL error, a shape mismatch on one rank) must surface, and ranks that caught different errors would issue different collectives.
    if dist.get_backend(group) == "nccl":               # RCCL: the flat form, one kernel
        dist.all_gather_into_tensor(out.view((world * per,) + tuple(local.shape[1:])), pad.contiguous(), group=group)
    else:                                               # gloo (CPU tests, host-staged): the list form, then the same layout
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad.contiguous(), group=group)
        out = torch.stack(parts)
    # view v was rendered by rank v mod world as its (v // world)-th view
    idx = torch.arange(n_views, device=local.device)
    return out[idx % world, idx // world]
