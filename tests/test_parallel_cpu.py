"""CPU tests of the view-parallel path: world_size-2 gloo processes, render function injected (the CPU oracle stands in
for the HIP rasterizer here -- test infrastructure only).  Checks that sharded loss/gradients equal the single-process ones."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
from sigman_release_amd import parallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VIEWS = [30, 37, 65]


class _OracleRender(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, cov3D, opacity, rgb, sv, H, W):
        from oracle import ref
        r = ref.forward(means3D.numpy(), opacity.numpy().reshape(-1), colors_precomp=rgb.numpy(), cov3D_precomp=cov3D.numpy(), **sv)
        ctx.r = r
        return torch.from_numpy(r.color.copy())

    @staticmethod
    def backward(ctx, g):
        from oracle import ref
        gr = ref.backward(ctx.r, g.numpy())
        t = torch.from_numpy
        return t(gr["means3D"]), t(gr["cov3D_precomp"]), t(gr["opacities"]), t(gr["colors_precomp"]), None, None, None


def _render_loss_factory(st, H, W):
    def render_loss(means3D, cov3D, opacity, rgb, my_views):
        total = 0.0
        for v in my_views:
            sv = cases.single_view(st, VIEWS.index(v))
            img = _OracleRender.apply(means3D, cov3D, opacity, rgb, sv, H, W)
            total = total + (img.clamp(0, 1) - 0.5).abs().sum() / (3 * H * W * len(VIEWS))
        return total
    return render_loss


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H = W = 48
    inp, st = cases.humanoid(P=800, H=H, W=W, seed=5, views=tuple(VIEWS))
    t = torch.from_numpy
    packed = parallel.pack_attributes(t(inp["means3D"]), t(inp["cov3D_precomp"]), t(inp["opacities"]), t(inp["colors_precomp"]))
    if rank != 0:
        packed = torch.zeros_like(packed)               # only the producer rank holds the attributes
    loss, grad = parallel.view_parallel_step(packed, VIEWS, _render_loss_factory(st, H, W), src=0)
    # north_star protocol: replicated attributes (here: what the broadcast above left on every rank), loss all-reduce only,
    # overlapped with the backward; the gradient stays the rank's partial sum over its own views
    loss2, grad2 = parallel.view_parallel_step(packed, VIEWS, _render_loss_factory(st, H, W), exchange="loss")
    # deferred wait: same numbers, the caller waits before it reads the loss
    loss3, grad3, work = parallel.view_parallel_step(packed, VIEWS, _render_loss_factory(st, H, W), exchange="loss", wait=False)
    if work is not None:
        work.wait()
    assert float(loss3) == float(loss2) and torch.equal(grad3, grad2)
    gsum = grad2.clone()
    dist.all_reduce(gsum)
    q.put((rank, float(loss), grad.numpy(), float(loss2), gsum.numpy(), float(grad2.abs().sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_view_parallel_gloo_matches_single_process():
    H = W = 48
    inp, st = cases.humanoid(P=800, H=H, W=W, seed=5, views=tuple(VIEWS))
    t = torch.from_numpy
    packed = parallel.pack_attributes(t(inp["means3D"]), t(inp["cov3D_precomp"]), t(inp["opacities"]), t(inp["colors_precomp"]))
    loss1, grad1 = parallel.view_parallel_step(packed.clone(), VIEWS, _render_loss_factory(st, H, W))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=300) for _ in range(2)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    partial_norms = []
    for rank, loss, grad, loss2, gsum, pnorm in res:
        assert abs(loss - float(loss1)) <= 1e-6 * max(1.0, abs(float(loss1)))
        np.testing.assert_allclose(grad, grad1.numpy(), atol=1e-6 * np.abs(grad1.numpy()).max())
        # exchange="loss": same global loss; the rank-local partial gradients add up to the global gradient
        assert abs(loss2 - float(loss1)) <= 1e-6 * max(1.0, abs(float(loss1)))
        np.testing.assert_allclose(gsum, grad1.numpy(), atol=1e-6 * np.abs(grad1.numpy()).max())
        partial_norms.append(pnorm)
    assert all(p > 0 for p in partial_norms) and abs(partial_norms[0] - partial_norms[1]) > 0      # really partial, really different


def test_shard_views():
    assert parallel.shard_views(8, 3, 8) == [3]
    assert parallel.shard_views(90, 0, 8) == list(range(0, 90, 8))
    assert sorted(sum((parallel.shard_views(90, r, 8) for r in range(8)), [])) == list(range(90))
    assert parallel.shard_views(2, 5, 8) == []


def test_extra_outputs_are_seeded_with_the_loss():
    """render_loss may return (loss, extra_outputs, extra_grads): the extras are seeded in the SAME backward (how bench.py's C5 step
    feeds dL/ddepth and dL/dalpha into the rasterizer node without reduction kernels)."""
    P = 7
    packed = torch.arange(13 * P, dtype=torch.float32) / 10.0
    w = torch.linspace(0.5, 1.5, 3 * P).reshape(P, 3)

    def rl_scalar(m, c, o, r, mine):
        return (m * m).sum() + ((m * 2.0) * w).sum()

    def rl_tuple(m, c, o, r, mine):
        return (m * m).sum(), [m * 2.0], [w]
    for exchange in ("full", "loss"):
        l1, g1 = parallel.view_parallel_step(packed.clone(), [30], rl_scalar, exchange=exchange)
        l2, g2 = parallel.view_parallel_step(packed.clone(), [30], rl_tuple, exchange=exchange, seed_grad=torch.ones(()))
        assert torch.allclose(g1, g2) and float(g1.abs().sum()) > 0
        assert abs(float(l2) - float((packed[:3 * P] ** 2).sum())) < 1e-4


def test_bench_refuses_to_report_fewer_ranks_than_asked():
    """`bench.py --gpus 2` launches its own two ranks (no launcher needed) and, when they cannot run (no GPU here), exits non-zero
    instead of printing a single-process line with n_gpus: 1 (missing #2)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode != 0
    assert "starting 2 ranks" in r.stderr and "torch.distributed.run" in r.stderr
    assert '"n_gpus"' not in r.stdout
    # under a launcher that started a different number of ranks than --gpus: refuse as well
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                        timeout=600, env=env2)
    assert r2.returncode != 0 and '"n_gpus"' not in r2.stdout


# ---- several subjects per step: the pipelined "full" exchange and the image gather (world-size-2 gloo, a cheap analytic "renderer")
_PV = [30, 37, 45, 53, 65]


def _toy_render_loss(chunk):
    """A differentiable stand-in with the renderer's interface: depends on the view ids, on every attribute and on the chunk."""
    def render_loss(m, c, o, r, my_views):
        total = 0.0
        for v in my_views:
            w = 1.0 + 0.01 * v + 0.1 * chunk
            total = total + (torch.sin(m * w).sum() + (c * c).sum() * w + (o * r.sum(1, keepdim=True)).sum() * (w - 0.5)) / (len(_PV) * m.shape[0])
        return total
    return render_loss


def _pipe_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S, P = 4, 50
    gen = torch.Generator().manual_seed(11)
    truth = [torch.randn(13 * P, generator=gen) for _ in range(S)]
    srcs = [0, 1, 1, 0]                                                        # producers differ per subject
    fns = [_toy_render_loss(c) for c in range(S)]
    out = {}
    for pipeline in (False, True):
        chunks = [t.clone() if rank == srcs[c] else torch.zeros_like(t) for c, t in enumerate(truth)]
        for _ in range(2):                                                     # twice: the second step starts from broadcast contents
            losses, grads = parallel.view_parallel_subjects(chunks, _PV, fns, srcs=srcs, pipeline=pipeline)
        out[pipeline] = (losses.clone(), [g.clone() for g in grads], [c.clone() for c in chunks])
    assert torch.equal(out[True][0], out[False][0])
    assert all(torch.equal(a, b) for a, b in zip(out[True][1], out[False][1]))
    assert all(torch.equal(a, t) for a, t in zip(out[True][2], truth))          # every rank ends up holding every subject's attributes
    # image gather: rank r "rendered" views {v : v mod world = r} of 5 views; every rank gets all 5 in view order
    mine = parallel.shard_views(5, rank, world)
    local = torch.stack([torch.full((3, 4, 6), float(v)) for v in mine])
    allv = parallel.all_gather_images(local, 5)
    assert allv.shape == (5, 3, 4, 6) and all(float(allv[v].mean()) == float(v) for v in range(5))
    q.put((rank, out[True][0].numpy(), [g.numpy() for g in out[True][1]]))
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_subject_exchange_matches_blocking_and_single_process():
    """view_parallel_subjects: 4 subjects from different producer ranks, 5 views sharded over 2 ranks; pipelined == blocking bit for bit on
    every rank, and both equal the single-process loss / gradient of every subject; all_gather_images returns the views in view order."""
    S, P = 4, 50
    gen = torch.Generator().manual_seed(11)
    truth = [torch.randn(13 * P, generator=gen) for _ in range(S)]
    want = parallel.view_parallel_subjects([t.clone() for t in truth], _PV, [_toy_render_loss(c) for c in range(S)])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_pipe_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=300) for _ in range(2)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, losses, grads in res:
        np.testing.assert_allclose(losses, want[0].numpy(), rtol=1e-5)
        for c in range(S):
            np.testing.assert_allclose(grads[c], want[1][c].numpy(), atol=1e-6 * np.abs(want[1][c].numpy()).max())
    np.testing.assert_array_equal(res[0][1], res[1][1])
    for c in range(S):
        np.testing.assert_array_equal(res[0][2][c], res[1][2][c])              # all-reduced: identical on both ranks


# ---- the world size north_star names: 8 ranks (gloo, this box's CPU cores), a cheap analytic "renderer" with the renderer's interface
_V8 = [30, 37, 45, 53, 65, 85, 0, 8]            # C3: the rig's 8 training views, one per rank


def _toy8(views_all, chunk=0):
    def render_loss(m, c, o, r, my_views):
        total = m.sum() * 0.0
        for v in my_views:
            w = 1.0 + 0.01 * v + 0.1 * chunk
            total = total + (torch.sin(m * w).sum() + (c * c).sum() * w + (o * r.sum(1, keepdim=True)).sum() * (w - 0.5)) / (len(views_all) * m.shape[0])
        return total
    return render_loss


def _world8_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = 40
    gen = torch.Generator().manual_seed(23)
    packed_true = torch.randn(13 * P, generator=gen)
    # (1) C3's shape: 8 views <-> 8 ranks, one view each.  "loss": replicated attributes, the loss all-reduce only, partial gradients;
    #     "full": broadcast from the producer, packed-gradient all-reduce
    assert parallel.shard_views(len(_V8), rank, world) == [rank]
    loss_l, grad_l = parallel.view_parallel_step(packed_true.clone(), _V8, _toy8(_V8), exchange="loss")
    gsum = grad_l.clone()
    dist.all_reduce(gsum)
    packed = packed_true.clone() if rank == 3 else torch.zeros_like(packed_true)          # (producer: rank 3)
    loss_f, grad_f = parallel.view_parallel_step(packed, _V8, _toy8(_V8), src=3)
    assert torch.equal(packed, packed_true)
    # (2) C4's shape: 90 views -> 11-12 per rank, forward only, one padded all-gather in view order
    mine = parallel.shard_views(90, rank, world)
    assert len(mine) == (12 if rank < 2 else 11) and mine[0] == rank
    local = torch.stack([torch.full((3, 2, 3), float(v)) + torch.arange(3.0).reshape(3, 1, 1) for v in mine])
    allv = parallel.all_gather_images(local, 90)
    assert allv.shape == (90, 3, 2, 3)
    assert all(torch.equal(allv[v], torch.full((3, 2, 3), float(v)) + torch.arange(3.0).reshape(3, 1, 1)) for v in range(90))
    # (3) 8 subjects as 8 pipelined chunks with alternating producers: pipelined == blocking, bit for bit
    S = 8
    truth = [torch.randn(13 * P, generator=gen) for _ in range(S)]
    srcs = [c % world for c in range(S)]
    fns = [_toy8(_V8, c) for c in range(S)]
    out = {}
    for pipeline in (False, True):
        chunks = [t.clone() if rank == srcs[c] else torch.zeros_like(t) for c, t in enumerate(truth)]
        losses, grads = parallel.view_parallel_subjects(chunks, _V8, fns, srcs=srcs, pipeline=pipeline)
        out[pipeline] = (losses.clone(), [g.clone() for g in grads])
        assert all(torch.equal(a, t) for a, t in zip(chunks, truth))
    assert torch.equal(out[True][0], out[False][0]) and all(torch.equal(a, b) for a, b in zip(out[True][1], out[False][1]))
    q.put((rank, float(loss_l), gsum.numpy(), float(grad_l.abs().sum()), float(loss_f), grad_f.numpy(), out[True][0].numpy(), [g.numpy() for g in out[True][1]]))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_c3_c4_shapes_and_pipelined_chunks():
    """World size 8 (the node north_star names), gloo on the CPU: C3's 8 views sharded one per rank -- loss all-reduce with partial gradients
    that add up to the single-process gradient ("loss"), attribute broadcast from a non-zero producer rank + packed-gradient all-reduce
    ("full") --; C4's 90 views -> 12, 12, 11, ... per rank through the padded `all_gather_images` (uneven shards) in view order;
    `view_parallel_subjects` with 8 chunks from 8 different producers, pipelined == blocking bit for bit, == single process."""
    world, P = 8, 40
    gen = torch.Generator().manual_seed(23)
    packed_true = torch.randn(13 * P, generator=gen)
    loss1, grad1 = parallel.view_parallel_step(packed_true.clone(), _V8, _toy8(_V8))
    truth = [torch.randn(13 * P, generator=gen) for _ in range(8)]
    want = parallel.view_parallel_subjects([t.clone() for t in truth], _V8, [_toy8(_V8, c) for c in range(8)])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_world8_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=300) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    g1 = grad1.numpy()
    norms = []
    for rank, loss_l, gsum, pnorm, loss_f, grad_f, sub_losses, sub_grads in sorted(res, key=lambda x: x[0]):
        assert abs(loss_l - float(loss1)) <= 1e-5 * max(1.0, abs(float(loss1))) and abs(loss_f - float(loss1)) <= 1e-5 * max(1.0, abs(float(loss1)))
        np.testing.assert_allclose(gsum, g1, atol=2e-6 * np.abs(g1).max())              # the 8 partial gradients add up to the global one
        np.testing.assert_allclose(grad_f, g1, atol=2e-6 * np.abs(g1).max())
        np.testing.assert_allclose(sub_losses, want[0].numpy(), rtol=1e-5)
        for c in range(8):
            np.testing.assert_allclose(sub_grads[c], want[1][c].numpy(), atol=2e-6 * np.abs(want[1][c].numpy()).max())
        norms.append(pnorm)
    assert all(n > 0 for n in norms) and len(set(round(n, 6) for n in norms)) == 8       # eight different, really partial gradients
    for r in res[1:]:
        np.testing.assert_array_equal(r[5], res[0][5])                                 # all-reduced gradients: identical on every rank
