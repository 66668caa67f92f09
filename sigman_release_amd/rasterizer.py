"""Drop-in Python surface of the third-party `diff_gaussian_rasterization` package, backed by the
gfx950 HIP library (include/sigman_gsplat.h), plus a view-batched variant.

Names, argument order, return order and error strings mirror what the reference imports and calls at
  /root/reference/core/gaussians/gs.py:8-11    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
  /root/reference/core/gaussians/gs.py:82-106  settings(...)  ->  rasterizer(means3D=, means2D=, shs=, colors_precomp=, opacities=, cov3D_precomp=)
and return (color [3,H,W], radii [P] int32, depth [1,H,W], alpha [1,H,W]).  Gradients come back in the
forward-argument order (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, None)
exactly like upstream's _RasterizeGaussians.backward.

The batched entry point `rasterize_gaussians_batched` renders all B*V views of a step in ONE launch chain
(the reference loops `for b ... for v ...` at gs.py:62,75 with a D2H sync per view).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import NamedTuple, Optional

import torch

from . import _cabi


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class BatchedRasterizationSettings(NamedTuple):
    """Settings for n_views = S * views_per_subject view slots rendered together."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor            # [3]
    scale_modifier: float
    viewmatrix: torch.Tensor    # [n_views,4,4]  (= cam_view, i.e. w2c^T)
    projmatrix: torch.Tensor    # [n_views,4,4]  (= cam_view_proj)
    sh_degree: int
    campos: torch.Tensor        # [n_views,3]
    views_per_subject: int
    debug: bool = False
    # 0: exact mode (one device->host read of num_rendered per batched forward, like upstream does per view).
    # >0: sync-free mode: binning buffers are pre-sized for this many tile instances and the count reaches the host through an async
    #     copy into a pinned slot (one slot per pending forward).  An overflow raises RuntimeError: in the call itself when no input
    #     needs a gradient (inference: nothing else would ever look), otherwise at its backward -- or at the start of this thread's
    #     next forward / at check_pending_overflows(), whichever comes first.  Never a memory fault (the kernels bounds-check).
    # -1: automatic: the first call with a given (P, n_views, H, W) runs in exact mode and remembers its count; later calls are
    #     sync-free with 1.3x that capacity.  The count reaches a pinned host slot right after the duplicate kernel, i.e. while the
    #     host is still queueing the rest of the forward: it is checked once everything is queued, and an overflow silently re-runs
    #     the forward in exact mode (and raises the remembered capacity) -- no blocking read, no user-visible failure mode.
    max_rendered: int = 0
    # Will the backward be handed dL/ddepth or dL/dalpha?  True: the forward stores their checkpoints (a third of its checkpoint stream).
    # None / False (default): it does not -- no call path of the reference differentiates depth or alpha (gs.py:99,107-109 drops depth,
    # SURVEY 8a A6b) -- and a backward that does get such gradients transparently produces the checkpoints with a second compositing
    # pass first: same results either way, this is a performance hint only.
    depth_alpha_grads: Optional[bool] = None




_EMPTY = torch.Tensor([])          # upstream passes torch.Tensor([]) for every missing optional; one shared instance (never written)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None or t.numel() == 0 else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(dev: Optional[torch.device] = None):
    """hipStream_t of PyTorch's current stream (torch.cuda.current_stream() costs ~20 us per call on the bench host; the raw
    query costs well under 1 us)."""
    if _raw_stream is not None:
        idx = dev.index if dev is not None and dev.index is not None else torch.cuda.current_device()
        return _raw_stream(idx)
    return torch.cuda.current_stream(dev).cuda_stream


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype is torch.float32 and t.is_contiguous():
        return t
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_OVF_BIT = 1 << 63


class _Slot:
    """One pinned 16-byte host slot + one event for the asynchronous num_rendered read-back of ONE forward.  Sync-free mode: the
    library publishes word [0] = count | overflow << 63 (a single 8-byte store / copy; -1 = not there yet).  Exact mode: [0] = count,
    [1] = overflow flag, both final when the call returns."""
    __slots__ = ("pool", "np", "ptr", "ev", "handle", "capacity", "checked", "result", "by_copy")

    def arrived(self):
        # a kernel publishes the word with ONE 8-byte store: polling it is safe.  An async device-to-host copy may land piecewise
        # (observed: bytes 0-4 of the count next to three bytes of the -1 sentinel), so there only the event says "complete".
        return self.ev.query() if self.by_copy else int(self.np[0]) != -1

    def read(self, block: bool):
        """(count, overflow) or None if the word has not arrived and block is False."""
        if not self.arrived():
            if not block:
                return None
            if self.by_copy:
                self.ev.synchronize()
            else:
                # the word is stored by the emission kernel itself (no event is recorded for it: an event between two kernels of the
                # chain costs a ~5 us bubble on the GPU): poll, and drain the device if it takes unusually long
                spins = 0
                while int(self.np[0]) == -1 and spins < 200000:
                    spins += 1
                if int(self.np[0]) == -1:
                    torch.cuda.synchronize()            # everything queued has run now: the word is there, or it never will be
                    if int(self.np[0]) == -1:
                        raise RuntimeError("the forward's instance count never reached its pinned host slot (a launch of the forward chain "
                                           "failed or was aborted, or the library and this binding disagree about SgrForwardState.nr_by_copy)")
        w = int(self.np[0]) & 0xFFFFFFFFFFFFFFFF
        return w & (_OVF_BIT - 1), 1 if (w & _OVF_BIT) else 0


class _SlotPool:
    """Per-device pool of _Slot objects (allocating pinned memory or events per call would cost more than the sync it replaces).
    Every forward takes its OWN slot and hands it back once its count has been looked at, so any number of forwards may be pending
    (the reference's loop issues B*V = 64 forwards before one backward, gs.py:62-109); the pool grows in chunks of 32."""

    def __init__(self):
        self.free, self.chunks = [], []

    def _grow(self, n=32):
        buf = torch.zeros(n, 2, dtype=torch.int64).pin_memory()
        arr = buf.numpy()                       # same memory: cheap host-side reads / sentinel writes
        self.chunks.append(buf)
        for k in range(n):
            sl = _Slot()
            sl.pool, sl.np, sl.ptr = self, arr[k], buf[k].data_ptr()
            sl.ev = torch.cuda.Event()
            sl.ev.record()                      # materialises the underlying hipEvent_t so its handle can cross the C ABI
            sl.handle = sl.ev.cuda_event
            self.free.append(sl)

    def acquire(self, capacity):
        if not self.free:
            self._grow()
        sl = self.free.pop()
        sl.np[0] = -1                           # sentinel: "the count has not arrived yet"
        sl.np[1] = 0
        sl.capacity, sl.checked, sl.result, sl.by_copy = capacity, False, None, True
        return sl

    def release(self, sl):
        if sl is not None and sl.pool is self:
            sl.pool = None                      # (guards double release)
            fresh = _Slot()
            fresh.pool, fresh.np, fresh.ptr, fresh.ev, fresh.handle = self, sl.np, sl.ptr, sl.ev, sl.handle
            self.free.append(fresh)


_slot_pools = {}      # device index -> _SlotPool


def _pool(dev: torch.device) -> _SlotPool:
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    p = _slot_pools.get(idx)
    if p is None:
        with torch.cuda.device(idx):
            p = _slot_pools[idx] = _SlotPool()
    return p


def _overflow_error(count, capacity, earlier=False):
    which = "an EARLIER forward of this thread (reported now: nobody had looked at its count yet) are" if earlier else "this forward are"
    return RuntimeError(f"num_rendered {count} exceeds max_rendered {capacity}: results of {which} truncated; "
                        "raise BatchedRasterizationSettings.max_rendered (or use 0 = exact mode)")


def check_pending_overflows(block: bool = True):
    """Sync-free mode (max_rendered > 0): look at the counts of this thread's earlier forwards that nobody has checked yet (a forward
    whose backward never ran, e.g. under torch.no_grad() or when its output was dropped) and raise if one of them was truncated.
    Called without blocking at the start of every forward; call it yourself with block=True after the last forward of a loop."""
    if block and _cabi._node not in (None, False):
        _cabi._node.check_pending_batched()         # forwards that went through the C++ batched node (rasterize_l1_loss_batched)
    pend = getattr(_pending, "slots", None)
    if not pend:
        return
    keep, err = [], None
    for sl in pend:
        if sl.checked:
            continue
        r = sl.read(block)
        if r is None:
            keep.append(sl)
            continue
        sl.checked, sl.result = True, r
        sl.pool.release(sl)
        if r[1] and err is None:
            err = _overflow_error(r[0], sl.capacity, earlier=True)
    _pending.slots = keep
    if err is not None:
        raise err


def set_count_wait(mode: str = "own"):
    """Sync-free mode (max_rendered > 0) through the C++ batched nodes: WHEN a backward looks at its forward's instance count.
    "own" (default): it waits for it -- an overflow raises before the optimizer step, and the host is at most one step ahead of the GPU.
    "lazy" / "lazy:N" (N = 1..16): it only looks; a count that has not arrived is waited for by the thread's forward N + 1 later at the latest,
    so the host may run N + 1 steps ahead (what keeps the GPU fed on a busy shared host: profiles/r05_count_wait_ab.txt) and an overflow is reported
    at most N steps late -- never lost: call check_pending_overflows(True) behind the loop.  Process-wide; == SIGMAN_COUNT_WAIT."""
    node = _cabi.torch_node()
    if node is None:
        raise RuntimeError("set_count_wait: the C++ autograd nodes (lib/sgr_torch_node.so) are not loaded (not built, or SIGMAN_PY_NODE=1); "
                           "the Python nodes always wait for their own forward's count")
    node.set_count_wait(mode)


# one persistent allocator callback for the C ABI (creating a ctypes callback per call costs ~10 us); it serves the call
# that is currently in flight on this thread: PyTorch allocates, the library only receives the pointer.
# (thread-local: ctypes releases the GIL during the C call, and the forward (caller's thread) and a backward (autograd engine
# thread) of different graphs may be in flight at the same time)
import threading as _threading
_alloc_target = _threading.local()
_pending = _threading.local()          # .slots: this thread's forwards whose instance count nobody has looked at yet


def _alloc_cb(_user, which, nbytes):
    # An exception must not leave this function: ctypes would print it and hand the C side an UNINITIALISED return value (a garbage device pointer:
    # kernels launched on it -- found by tools/fuzz_determinism.py with a scene whose image blob did not fit the GPU).  NULL makes the library
    # return an error before it launches anything on that blob; the message travels in _alloc_target.error.
    try:
        t = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=_alloc_target.dev)
    except Exception as e:      # noqa: BLE001  (torch.OutOfMemoryError above all)
        _alloc_target.error = f"allocating {int(nbytes) / 2 ** 30:.2f} GiB for blob {int(which)}: {e}"
        return 0
    _alloc_target.blobs[which] = t
    return t.data_ptr()


_ALLOC = _cabi.ALLOC_FN(_alloc_cb)
_NO_ALLOC = _cabi.ALLOC_FN(0)                 # NULL allocator: the blobs are pre-allocated (sgr_rasterize_forward, include/sigman_gsplat.h)


class _Ctx:
    """What the forward leaves behind for the backward (== upstream geomBuffer / binningBuffer / imgBuffer + num_rendered)."""
    __slots__ = ("state", "blobs", "radii", "dims", "slot", "capacity", "true_rendered", "keep", "pb")

    def check_overflow(self):
        """Sync-free mode: raise if the forward needed more tile instances than `max_rendered` (cheap: the copy finished long ago)."""
        sl = self.slot
        if sl is not None:
            self.slot = None
            if not sl.checked:
                sl.checked, sl.result = True, sl.read(True)
                sl.pool.release(sl)
            count, overflow = sl.result
            self.true_rendered = count
            if overflow:
                raise _overflow_error(count, self.capacity)

    def view(self, which, off, count, dtype):
        """Typed tensor view into one of the three blobs (debug / parity tests)."""
        esz = torch.empty(0, dtype=dtype).element_size()
        return self.blobs[which][off: off + count * esz].view(dtype)


_pb_cache = _threading.local()       # last problem struct per thread: a training loop presents the same shapes and pointers step after step


def _make_problem(means3D, opacities, colors_precomp, shs, cov3D_precomp, scales, rotations, st: BatchedRasterizationSettings):
    S, P = means3D.shape[0], means3D.shape[1]
    nv = st.viewmatrix.shape[0]
    if nv != S * st.views_per_subject:
        raise RuntimeError(f"viewmatrix has {nv} views but inputs describe {S} subjects x {st.views_per_subject} views")
    M = 0 if shs is None else shs.shape[2]
    sig = (P, nv, st.views_per_subject, int(st.image_height), int(st.image_width), int(st.sh_degree), M,
           float(st.tanfovx), float(st.tanfovy), float(st.scale_modifier),
           _ptr(means3D), _ptr(opacities), _ptr(colors_precomp), _ptr(shs), _ptr(cov3D_precomp), _ptr(scales),
           _ptr(rotations), _ptr(st.viewmatrix), _ptr(st.projmatrix), _ptr(st.campos), _ptr(st.bg))
    if getattr(_pb_cache, "sig", None) == sig:
        return _pb_cache.pb             # (the struct is only read by the library)
    pb = _cabi.SgrProblem(*sig)
    _pb_cache.sig, _pb_cache.pb = sig, pb
    return pb


_auto_capacity = {}   # (device, P, n_views, H, W) -> remembered capacity of max_rendered = -1 (automatic) mode
_blob_sizes = {}      # (device, P, n_views, H, W, capacity, aux, has_sh) -> (geom, binning, image) bytes of the last forward with these shapes


class _debug_scope:
    """upstream's debug=True: every kernel launch of the enclosed library calls is followed by a device synchronise and a fault
    names the kernel (sgr_set_debug, include/sigman_gsplat.h)."""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        if self.on:
            self.old = _cabi.lib().sgr_set_debug(1)

    def __exit__(self, *exc):
        if self.on:
            _cabi.lib().sgr_set_debug(self.old)
        return False


def _forward_impl(means3D, opacities, colors_precomp, shs, cov3D_precomp, scales, rotations, st: BatchedRasterizationSettings,
                  need_ctx: bool, with_aux: bool = True, clear: Optional[torch.Tensor] = None, l1=None):
    L = _cabi.lib()
    dev = means3D.device
    if dev.type != "cuda":
        raise RuntimeError("sigman_release_amd rasterizer needs tensors on a ROCm device (there is no CPU fallback)")
    S, P = means3D.shape[0], means3D.shape[1]
    H, W = int(st.image_height), int(st.image_width)
    nv = st.viewmatrix.shape[0]
    pb = _make_problem(means3D, opacities, colors_precomp, shs, cov3D_precomp, scales, rotations, st)
    f32 = torch.float32
    color = torch.empty(nv, 3, H, W, dtype=f32, device=dev)
    depth = torch.empty(nv, 1, H, W, dtype=f32, device=dev)
    alpha = torch.empty(nv, 1, H, W, dtype=f32, device=dev)
    radii = torch.empty(nv, P, dtype=torch.int32, device=dev)
    check_pending_overflows(block=False)               # an earlier sync-free forward that nobody checked (no backward ran)
    capacity = int(getattr(st, "max_rendered", 0) or 0)
    auto_key = None
    didx = dev.index if dev.index is not None else torch.cuda.current_device()
    if capacity < 0:                                   # automatic mode: exact the first time, then sync-free with the remembered capacity
        auto_key = (didx, P, nv, H, W)
        capacity = _auto_capacity.get(auto_key, 0)
    slot = _pool(dev).acquire(capacity)
    nr_host, nr_event, nr_ptr, nr_handle = slot.np, slot.ev, slot.ptr, slot.handle
    state = _cabi.SgrForwardState()
    blobs = [None, None, None, None]
    use_aux = (1 if getattr(st, "depth_alpha_grads", None) else 3) if (need_ctx and with_aux) else 0    # 3: (depth, alpha) checkpoints on demand
    clear_ptr, clear_bytes = (None, 0) if clear is None else (clear.data_ptr(), clear.numel() * clear.element_size())
    args = (C.byref(pb), capacity, use_aux)
    outs = (color.data_ptr(), depth.data_ptr(), alpha.data_ptr(), radii.data_ptr(), nr_ptr, nr_handle if capacity > 0 else None,
            clear_ptr, clear_bytes, C.byref(state), _stream(dev))
    forward_fn = L.sgr_rasterize_forward
    if l1 is not None:
        # fused image loss (l1 = a callable that, given the colour tensor, returns the filled SgrL1Epilogue): the loss kernel is queued by
        # the same C call, right behind the compositing kernel
        ep = l1(color)
        outs = outs[:-1] + (C.byref(ep), outs[-1])
        forward_fn = L.sgr_rasterize_forward_l1
    # sync-free mode: blob sizes only depend on the shapes, so from the second call on the blobs are allocated here and handed over
    # directly (no allocator callbacks through ctypes)
    fused_key = 0
    if l1 is not None and ep.fuse_backward:      # (the fused step keeps its partial records in the image blob: the thread's switch is part of the key)
        fused_key = L.sgr_set_fused_step(1)
        L.sgr_set_fused_step(fused_key)
    size_key = (didx, P, nv, H, W, capacity, use_aux, shs is not None, fused_key) if capacity > 0 else None
    sizes = _blob_sizes.get(size_key) if size_key is not None else None
    status = 2
    with _debug_scope(getattr(st, "debug", False)):
        if sizes is not None:
            u8 = torch.uint8
            blobs[0], blobs[1], blobs[2] = (torch.empty(sizes[0], dtype=u8, device=dev), torch.empty(sizes[1], dtype=u8, device=dev),
                                            torch.empty(sizes[2], dtype=u8, device=dev))
            state.geom, state.binning, state.image = blobs[0].data_ptr(), blobs[1].data_ptr(), blobs[2].data_ptr()
            state.geom_bytes, state.binning_bytes, state.image_bytes = sizes
            status = forward_fn(*args, _NO_ALLOC, None, *outs)
        if status == 2:                    # first call with these shapes (or sizes changed): the library asks for memory through the callback
            _alloc_target.dev, _alloc_target.blobs, _alloc_target.error = dev, blobs, None
            status = forward_fn(*args, _ALLOC, None, *outs)
            if status == 0 and size_key is not None:
                _blob_sizes[size_key] = (max(int(state.geom_bytes), 256), max(int(state.binning_bytes), 256), max(int(state.image_bytes), 256))
    if status != 0:
        _pool(dev).release(slot)
    _cabi.check(status, "sgr_rasterize_forward", getattr(_alloc_target, "error", None))
    slot.by_copy = bool(state.nr_by_copy)
    pending = capacity > 0 and P > 0           # sync-free: the count is still on its way to the pinned slot
    if pending and (auto_key is not None or not use_aux):
        # automatic mode, and explicit sync-free forwards that will never see a backward (no input needs a gradient: torch.no_grad(),
        # eval, a ground-truth render): look at the count now.  Everything is queued; the count was published right after the
        # duplicate kernel (or by the copy behind the scan kernel), so this wait is short and the GPU stays busy.
        spins = 0
        while not slot.arrived() and spins < 20000:
            spins += 1
        count, overflow = slot.read(True)
        slot.checked, slot.result = True, (count, overflow)
        pending = False
        if overflow:
            if auto_key is None:
                _pool(dev).release(slot)
                raise _overflow_error(count, capacity)
            _pool(dev).release(slot)
            _auto_capacity[auto_key] = 0               # re-run exactly; the exact run below re-learns the capacity
            return _forward_impl(means3D, opacities, colors_precomp, shs, cov3D_precomp, scales, rotations, st, need_ctx,
                                 with_aux, clear, l1)
    if auto_key is not None and P > 0:
        count = int(state.true_rendered) if capacity == 0 else slot.result[0]
        if capacity == 0 or count * 1.1 > capacity:
            _auto_capacity[auto_key] = min(int(count * 1.3) + 4096, 0xFFFFFFE0)
    if pending:
        pend = getattr(_pending, "slots", None)
        if pend is None:
            pend = _pending.slots = []
        pend.append(slot)                      # found again by the backward's check, or by the next forward if no backward ever runs
        if len(pend) > 256:
            check_pending_overflows(block=True)
    elif not slot.checked:
        slot.checked, slot.result = True, ((int(state.true_rendered), 0) if capacity == 0 and P > 0 else (0, 0))
    ctx = None
    if need_ctx:
        ctx = _Ctx()
        ctx.state, ctx.blobs, ctx.radii = state, blobs, radii
        ctx.dims = (S, P, nv, H, W)
        ctx.slot, ctx.capacity = (slot if pending else None), capacity
        ctx.true_rendered = slot.result[0] if slot.result is not None else None
        ctx.keep = (st.viewmatrix, st.projmatrix, st.campos, st.bg)
        ctx.pb = pb            # the backward sees the same tensors (saved_tensors share their storage), so the struct is reused
    if not pending:
        _pool(dev).release(slot)
    return color, radii, depth, alpha, ctx


_GATHER_ONLY = object()       # grad_color marker for _backward_impl: sgr_rasterize_backward with grad_color = NULL (include/sigman_gsplat.h)
# The Python node of the fused rasterizer + L1 loss runs the UNFUSED chain (compositing, loss kernel, compositing backward, gather): it is the
# reference the C++ node's fused single-view step is compared with, bit for bit.  True (tests): it asks for the fused step as well, which
# brings the flavours the C++ node does not take (SH, scales + rotations, exact mode) through SgrL1Epilogue.fuse_backward.
FUSE_STEP_IN_PYTHON_NODE = False


def _backward_impl(ctx: _Ctx, means3D, opacities, colors_precomp, shs, cov3D_precomp, scales, rotations,
                   st: BatchedRasterizationSettings, grad_color, grad_depth, grad_alpha, img, grad_color_scale=None, want_means2D=True):
    L = _cabi.lib()
    S, P, nv, H, W = ctx.dims
    dev = means3D.device
    f32 = torch.float32
    pb = ctx.pb
    if pb.means3D != _ptr(means3D):      # (cannot happen through autograd; guards direct callers that pass other tensors)
        pb = _make_problem(means3D, opacities, colors_precomp, shs, cov3D_precomp, scales, rotations, st)
    if grad_color is _GATHER_ONLY:      # a fused forward (state.fused_bwd) already ran the compositing backward of its own loss: gather, scaled
        gC = None
    else:
        if grad_color is None:
            grad_color = torch.zeros(nv, 3, H, W, dtype=f32, device=dev)
        gC = _f32c(grad_color)
    gD = None if grad_depth is None else _f32c(grad_depth)
    gA = None if grad_alpha is None else _f32c(grad_alpha)
    d_means3D = torch.empty(S, P, 3, dtype=f32, device=dev)
    d_means2D = torch.empty(nv, P, 3, dtype=f32, device=dev) if want_means2D else None     # (dL/dNDC: a [n_views,P,3] write nobody reads on the reference path)
    d_op = torch.empty(S, P, dtype=f32, device=dev)
    d_cov = torch.empty(S, P, 6, dtype=f32, device=dev)
    d_col = torch.empty(S, P, 3, dtype=f32, device=dev) if shs is None else None
    d_sh = torch.empty_like(shs) if shs is not None else None
    d_sc = torch.empty(S, P, 3, dtype=f32, device=dev) if scales is not None else None
    d_rot = torch.empty(S, P, 4, dtype=f32, device=dev) if scales is not None else None
    blobs = ctx.blobs
    _alloc_target.dev, _alloc_target.blobs, _alloc_target.error = dev, blobs, None
    with _debug_scope(getattr(st, "debug", False)):
        _cabi.check(L.sgr_rasterize_backward(C.byref(pb), C.byref(ctx.state), _ptr(ctx.radii), _ptr(img[0]), _ptr(img[1]), _ptr(img[2]),
                                             _ptr(gC), _ptr(gD), _ptr(gA), _ptr(grad_color_scale), _ALLOC, None, _ptr(d_means3D), _ptr(d_means2D), _ptr(d_op),
                                             _ptr(d_col), _ptr(d_sh), _ptr(d_cov), _ptr(d_sc), _ptr(d_rot), _stream(dev)),
                    "sgr_rasterize_backward", getattr(_alloc_target, "error", None))
    grec = blobs[3]
    ctx.check_overflow()        # after the backward is queued: the host never idles the GPU while it waits for the forward's counter
    return d_means3D, d_means2D, d_sh, d_col, d_op, d_sc, d_rot, d_cov, grec


def _fwd_common(ctx, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, st, epilogue=None, clear=None, l1=None):
    # fp32-only op: inputs are cast here, so an enclosing autocast region (gs.py:98) cannot downcast them
    opt = lambda t: None if t is None or t.numel() == 0 else _f32c(t)
    means3D = _f32c(means3D)
    ctx.op_shape = tuple(opacities.shape)            # [S,P,1] (upstream's form) or [S,P]: the gradient goes back in the same shape
    opacities = _f32c(opacities).reshape(means3D.shape[0], means3D.shape[1])
    sh, colors_precomp, scales, rotations, cov3Ds_precomp = map(opt, (sh, colors_precomp, scales, rotations, cov3Ds_precomp))
    f32 = torch.float32
    if not (st.viewmatrix.dtype is f32 and st.projmatrix.dtype is f32 and st.campos.dtype is f32 and st.bg.dtype is f32
            and st.viewmatrix.is_contiguous() and st.projmatrix.is_contiguous() and st.campos.is_contiguous() and st.bg.is_contiguous()):
        st = st._replace(viewmatrix=_f32c(st.viewmatrix), projmatrix=_f32c(st.projmatrix), campos=_f32c(st.campos), bg=_f32c(st.bg))
    # inference (no input needs a gradient, e.g. under torch.no_grad()): skip the backward's auxiliary outputs (compact lists,
    # per-row checkpoints, bucket descriptors) -- the forward kernels then run their lighter variant
    wants_grad = any(ctx.needs_input_grad)
    color, radii, depth, alpha, c = _forward_impl(means3D, opacities, colors_precomp, sh, cov3Ds_precomp, scales, rotations,
                                                  st, need_ctx=True, with_aux=wants_grad, clear=clear, l1=l1)
    ctx.sgr = c
    ctx.st = st
    # unused outputs (depth / alpha on the reference path, gs.py:99,107-109) then arrive as None in backward instead of as
    # materialised zero tensors: no fill kernels, and the backward kernel variant without the depth/alpha channels runs
    ctx.set_materialize_grads(False)
    ctx.has = (sh is not None, colors_precomp is not None, scales is not None, cov3Ds_precomp is not None)
    # the outputs go through save_for_backward (a plain attribute would create a ctx -> output -> grad_fn -> ctx cycle and keep
    # every step's buffers alive until the garbage collector runs)
    extra = epilogue(color) if epilogue is not None else ()          # fused image-loss epilogue: tensors it needs in backward
    ctx.save_for_backward(means3D, opacities, colors_precomp, sh, cov3Ds_precomp, scales, rotations, color, depth, alpha, *extra)
    return color, radii, depth, alpha


def _bwd_common(ctx, grad_color, grad_depth, grad_alpha, grad_color_scale=None):
    means3D, opacities, colors_precomp, sh, cov3Ds_precomp, scales, rotations, color, depth, alpha = ctx.saved_tensors[:10]
    d_means3D, d_means2D, d_sh, d_col, d_op, d_sc, d_rot, d_cov, _ = _backward_impl(
        ctx.sgr, means3D, opacities, colors_precomp, sh, cov3Ds_precomp, scales, rotations, ctx.st, grad_color, grad_depth,
        grad_alpha, (color, depth, alpha), grad_color_scale, want_means2D=ctx.has_means2D)
    has_sh, has_col, has_sr, has_cov = ctx.has
    return (d_means3D, d_means2D if ctx.has_means2D else None, d_sh if has_sh else None, d_col if has_col else None, d_op.reshape(ctx.op_shape),
            d_sc if has_sr else None, d_rot if has_sr else None, d_cov if has_cov else None)


class _RasterizeGaussiansBatched(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, st):
        color, radii, depth, alpha = _fwd_common(ctx, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, st)
        ctx.has_means2D = means2D is not None and ctx.needs_input_grad[1]
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        return _bwd_common(ctx, grad_color, grad_depth, grad_alpha) + (None,)


def rasterize_gaussians_batched(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                raster_settings: BatchedRasterizationSettings):
    """Batched counterpart of upstream `rasterize_gaussians`.

    means3D [S,P,3], means2D [n_views,P,3] (dummy, receives dL/dNDC), opacities [S,P,1], sh [S,P,M,3] | colors_precomp [S,P,3],
    scales [S,P,3] + rotations [S,P,4] | cov3Ds_precomp [S,P,6].  Returns color [n_views,3,H,W], radii [n_views,P],
    depth [n_views,1,H,W], alpha [n_views,1,H,W].
    """
    st = raster_settings
    if (sh is None and scales is None and rotations is None and means2D is None and colors_precomp is not None and cov3Ds_precomp is not None
            and not getattr(st, "debug", False) and means3D.ndim == 3 and means3D.shape[1] > 0):
        node = _cabi.torch_node()
        if node is not None:
            # the reference's input flavour: the same node in C++ (csrc/torch_node.cpp, RenderBatchedNode) -- same capacity policy
            # (max_rendered < 0 automatic with the inline check and the transparent exact re-run, > 0 explicit, 0 exact), same results,
            # a third of the host time per call
            return tuple(node.rasterize_batched(means3D, colors_precomp, opacities, cov3Ds_precomp, st.viewmatrix, st.projmatrix, st.campos, st.bg,
                                                int(st.image_height), int(st.image_width), float(st.tanfovx), float(st.tanfovy), float(st.scale_modifier),
                                                int(st.views_per_subject), int(getattr(st, "max_rendered", 0) or 0), bool(getattr(st, "depth_alpha_grads", None))))
    return _RasterizeGaussiansBatched.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                            raster_settings)


class _RasterizeL1Batched(torch.autograd.Function):
    """Rasterizer + fused clamp/L1 image loss as ONE autograd node: the loss kernel runs right behind the
    compositing kernel and leaves dL/dcolor for the backward, which takes the upstream scalar dL/dloss as a device pointer
    (no elementwise kernel, no second Python autograd node).  == clamped_l1_loss(rasterize_gaussians_batched(...)[0], ...)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, st, target, mask, weight):
        L = _cabi.lib()
        nv = st.viewmatrix.shape[0]
        # [per-view partial sums | total]: zeroed on the side by the rasterizer's own kernels (caller_clear), no memset launch
        sums = torch.empty(nv + 1, dtype=torch.float32, device=means3D.device)

        keep = []

        def l1(color):
            tgt = _f32c(target)
            msk = None if mask is None else _f32c(mask)
            gimg = torch.empty_like(color)
            keep[:] = [gimg, tgt, msk]
            p = sums.data_ptr()
            return _cabi.SgrL1Epilogue(target=_ptr(tgt), mask=_ptr(msk), grad_color=gimg.data_ptr(), loss_per_view=p, loss_total=p + 4 * nv,
                                       weight=float(weight), sums_already_zero=1,
                                       fuse_backward=1 if (FUSE_STEP_IN_PYTHON_NODE and not getattr(st, "depth_alpha_grads", None)) else 0)

        color, radii, depth, alpha = _fwd_common(ctx, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, st,
                                                 lambda _color: (keep[0],), clear=sums, l1=l1)
        ctx.has_means2D = means2D is not None and ctx.needs_input_grad[1]
        loss, per_view = sums[nv], sums[:nv]
        ctx.mark_non_differentiable(radii, per_view)
        return loss, per_view, color, radii, depth, alpha

    @staticmethod
    def backward(ctx, g_loss, g_per_view, g_color, g_radii, g_depth, g_alpha):
        gimg = ctx.saved_tensors[10]
        if ctx.sgr.state.fused_bwd and g_loss is not None and g_color is None and g_depth is None and g_alpha is None:
            return _bwd_common(ctx, _GATHER_ONLY, None, None, g_loss.reshape(1).to(torch.float32)) + (None, None, None, None)
        if g_loss is None and g_color is None:
            g_color, scale = torch.zeros_like(gimg), None
        elif g_color is None:
            g_color, scale = gimg, g_loss.reshape(1).to(torch.float32)        # the common case: only the loss is used
        else:
            g_color, scale = (g_color if g_loss is None else g_color + gimg * g_loss), None
        return _bwd_common(ctx, g_color, g_depth, g_alpha, scale) + (None, None, None, None)


def rasterize_l1_loss_batched(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                              raster_settings: BatchedRasterizationSettings, target, mask=None, weight: float = 1.0):
    """-> (loss, per_view_loss [n_views], color, radii, depth, alpha);  loss = weight * sum(mask * |clamp(color, 0, 1) - target|)."""
    st = raster_settings
    cap = getattr(st, "max_rendered", 0) or 0
    if (cap > 0 and sh is None and scales is None and rotations is None and means2D is None and colors_precomp is not None
            and cov3Ds_precomp is not None and not getattr(st, "debug", False) and means3D.ndim == 3 and means3D.shape[1] > 0):
        node = _cabi.torch_node()
        if node is not None:
            # the reference's input flavour in the explicit sync-free mode: the same node in C++ (csrc/torch_node.cpp) -- half the host time per
            # step, which at one view is what decides whether the host keeps up with 137 us of kernels
            # (with or without the reference's loss mask, whole_loss.py:126-131; an upstream gradient into the colour output -- LPIPS next to
            # the L1 -- is added to the L1's inside the node's backward)
            return tuple(node.rasterize_l1_batched(means3D, colors_precomp, opacities, cov3Ds_precomp, st.viewmatrix, st.projmatrix, st.campos, st.bg, target,
                                                   _EMPTY if mask is None else mask, int(st.image_height), int(st.image_width), float(st.tanfovx), float(st.tanfovy), float(st.scale_modifier),
                                                   int(st.views_per_subject), int(cap), float(weight), bool(getattr(st, "depth_alpha_grads", None))))
    return _RasterizeL1Batched.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings, target, mask, weight)


class _RasterizeGaussians(torch.autograd.Function):
    """Single-view op with upstream's exact signature (9 forward args, 4 outputs, 9 grads)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        rs = raster_settings
        if not rs.debug and (means3D.ndim != 2 or means3D.shape[1] != 3):
            raise RuntimeError("means3D must have dimensions (num_points, 3)")
        st = BatchedRasterizationSettings(rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, rs.bg, rs.scale_modifier,
                                          rs.viewmatrix.reshape(1, 4, 4), rs.projmatrix.reshape(1, 4, 4), rs.sh_degree,
                                          rs.campos.reshape(1, 3), 1, rs.debug, -1)      # automatic sync-free mode
        u = lambda t: None if t is None or t.numel() == 0 else t.unsqueeze(0)
        P = means3D.shape[0]
        if rs.debug:
            # upstream's debug=True: the library synchronises after every kernel (st.debug -> sgr_set_debug) and a failure -- argument
            # errors included, upstream raises them from inside the native call -- leaves a snapshot of the inputs behind before it
            # is re-raised
            cpu_args = _cpu_copy((means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, tuple(rs)))
            try:
                if means3D.ndim != 2 or means3D.shape[1] != 3:
                    raise RuntimeError("means3D must have dimensions (num_points, 3)")
                color, radii, depth, alpha = _fwd_common(ctx, means3D.unsqueeze(0), u(sh), u(colors_precomp), opacities.reshape(1, P, 1),
                                                         u(scales), u(rotations), u(cov3Ds_precomp), st)
                torch.cuda.synchronize(means3D.device)
            except Exception:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise
        else:
            color, radii, depth, alpha = _fwd_common(ctx, means3D.unsqueeze(0), u(sh), u(colors_precomp), opacities.reshape(1, P, 1),
                                                     u(scales), u(rotations), u(cov3Ds_precomp), st)
        if rs.prefiltered and P > 0:
            # upstream traps ("Point is filtered although prefiltered is set. This shouldn't happen!") when a caller that promised
            # pre-filtered input hands over a point behind the near plane; here it is a Python error (costs one tiny kernel + a sync,
            # only when the flag is set -- the reference never sets it, gs.py:93)
            if not bool(mark_visible(means3D.detach(), rs.viewmatrix).all()):
                raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
        color, radii, depth, alpha = color[0], radii[0], depth[0], alpha[0]
        ctx.has_means2D = means2D is not None and ctx.needs_input_grad[1]
        ctx.debug = bool(rs.debug)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        ub = lambda t: None if t is None else t.unsqueeze(0)
        if ctx.debug:
            cpu_args = _cpu_copy((grad_color, grad_depth, grad_alpha) + tuple(ctx.saved_tensors[:7]))
            try:
                g = _bwd_common(ctx, ub(grad_color), ub(grad_depth), ub(grad_alpha))
                torch.cuda.synchronize(ctx.saved_tensors[0].device)
            except Exception:
                torch.save(cpu_args, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise
        else:
            g = _bwd_common(ctx, ub(grad_color), ub(grad_depth), ub(grad_alpha))
        return tuple(None if x is None else x[0] for x in g) + (None,)


def _cpu_copy(obj):
    if isinstance(obj, torch.Tensor):
        return obj.detach().cpu().clone()
    if isinstance(obj, (tuple, list)):
        return tuple(_cpu_copy(x) for x in obj)
    return obj


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    rs = raster_settings
    node = _cabi.torch_node()
    if node is not None and not rs.debug and not rs.prefiltered:
        # the same op as a C++ autograd node (csrc/torch_node.cpp): the reference calls this once per view from a Python loop
        # (gs.py:75-106), and the host time of the Python node below exceeded the kernels' time
        return tuple(node.rasterize_gaussians(means3D, _EMPTY if means2D is None else means2D, sh, colors_precomp, opacities, scales, rotations,
                                              cov3Ds_precomp, int(rs.image_height), int(rs.image_width), float(rs.tanfovx), float(rs.tanfovy), rs.bg,
                                              float(rs.scale_modifier), rs.viewmatrix, rs.projmatrix, int(rs.sh_degree), rs.campos))
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings)


def mark_visible(positions: torch.Tensor, viewmatrix: torch.Tensor) -> torch.Tensor:
    L = _cabi.lib()
    positions = _f32c(positions)
    out = torch.zeros(positions.shape[0], dtype=torch.uint8, device=positions.device)
    vm = _f32c(viewmatrix)
    _cabi.check(L.sgr_mark_visible(positions.shape[0], _ptr(positions), _ptr(vm), _ptr(out), _stream(positions.device)), "sgr_mark_visible")
    return out.bool()


class GaussianRasterizer(torch.nn.Module):
    """Same constructor / markVisible / forward contract as upstream's GaussianRasterizer (gs.py:96-106)."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            return mark_visible(positions, self.raster_settings.viewmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        e = _EMPTY
        shs = e if shs is None else shs
        colors_precomp = e if colors_precomp is None else colors_precomp
        scales = e if scales is None else scales
        rotations = e if rotations is None else rotations
        cov3D_precomp = e if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings)


def forward_debug(means3D, opacities, *, colors_precomp=None, shs=None, cov3D_precomp=None, scales=None, rotations=None,
                  settings: BatchedRasterizationSettings):
    """No-grad batched forward that also returns every intermediate artefact (for the bit-exact parity tests)."""
    with torch.no_grad():
        S, P = means3D.shape[0], means3D.shape[1]
        st = settings._replace(viewmatrix=_f32c(settings.viewmatrix), projmatrix=_f32c(settings.projmatrix),
                               campos=_f32c(settings.campos), bg=_f32c(settings.bg))
        opt = lambda t: None if t is None else _f32c(t)
        old_keep = _cabi.lib().sgr_set_keep_sorted_keys(1)          # the production forward stores only the point list behind the register sort
        try:
            color, radii, depth, alpha, c = _forward_impl(_f32c(means3D), _f32c(opacities).reshape(S, P), opt(colors_precomp),
                                                          opt(shs), opt(cov3D_precomp), opt(scales), opt(rotations), st,
                                                          need_ctx=True, with_aux=False)
        finally:
            _cabi.lib().sgr_set_keep_sorted_keys(old_keep)
        nv = st.viewmatrix.shape[0]
        H, W = int(st.image_height), int(st.image_width)
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        c.check_overflow()
        R = c.true_rendered
        s_ = c.state
        i32, i64, f32, u8 = torch.int32, torch.int64, torch.float32, torch.uint8
        in_b = bool(s_.result_in_b)
        return dict(color=color, radii=radii, depth=depth, alpha=alpha,
                    rec=c.view(0, s_.off_rec, nv * P * 16, f32).view(nv, P, 16),
                    rect=c.view(0, s_.off_rect, nv * P * 4, i32).view(nv, P, 4)[:, :, :2],
                    clamped=c.view(0, s_.off_clamped, nv * P, u8) if shs is not None else None,
                    point_list=c.view(1, s_.off_vals_b if in_b else s_.off_vals_a, R, i32),
                    keys=c.view(1, s_.off_keys_b if in_b else s_.off_keys_a, R, i64),
                    ranges=c.view(2, s_.off_ranges, nv * tiles * 2, i32).view(nv, tiles, 2),
                    final_T=c.view(2, s_.off_final_T, nv * H * W, f32).view(nv, H, W),
                    n_contrib=c.view(2, s_.off_n_contrib, nv * H * W, i32).view(nv, H, W),
                    num_rendered=R, ctx=c, settings=st)
