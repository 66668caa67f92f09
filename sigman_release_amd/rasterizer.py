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


# SIGMAN_BWD_V1=1 selects the pixel-parallel backward (no auxiliary forward outputs) for A/B comparisons
_USE_BWD_V1 = os.environ.get("SIGMAN_BWD_V1", "0") == "1"


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None or t.numel() == 0 else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class _Ctx:
    """Plain holder for the forward's device buffers (== upstream geomBuffer / binningBuffer / imgBuffer)."""
    __slots__ = ("pb", "keep", "rec", "radii", "rect", "clamped", "point_list", "keys", "ranges", "final_T", "n_contrib",
                 "num_rendered", "dims", "aux", "images")


def _make_problem(means3D, opacities, colors_precomp, shs, cov3D_precomp, scales, rotations, st: BatchedRasterizationSettings):
    S, P = means3D.shape[0], means3D.shape[1]
    nv = st.viewmatrix.shape[0]
    if nv != S * st.views_per_subject:
        raise RuntimeError(f"viewmatrix has {nv} views but inputs describe {S} subjects x {st.views_per_subject} views")
    M = 0 if shs is None else shs.shape[2]
    pb = _cabi.SgrProblem(P, nv, st.views_per_subject, int(st.image_height), int(st.image_width), int(st.sh_degree), M,
                          float(st.tanfovx), float(st.tanfovy), float(st.scale_modifier),
                          _ptr(means3D), _ptr(opacities), _ptr(colors_precomp), _ptr(shs), _ptr(cov3D_precomp), _ptr(scales),
                          _ptr(rotations), _ptr(st.viewmatrix), _ptr(st.projmatrix), _ptr(st.campos), _ptr(st.bg))
    return pb


def _forward_impl(means3D, opacities, colors_precomp, shs, cov3D_precomp, scales, rotations, st: BatchedRasterizationSettings,
                  need_ctx: bool, keep_keys: bool = False):
    L = _cabi.lib()
    dev = means3D.device
    if dev.type != "cuda":
        raise RuntimeError("sigman_release_amd rasterizer needs tensors on a ROCm device (there is no CPU fallback)")
    S, P = means3D.shape[0], means3D.shape[1]
    H, W = int(st.image_height), int(st.image_width)
    nv = st.viewmatrix.shape[0]
    pb = _make_problem(means3D, opacities, colors_precomp, shs, cov3D_precomp, scales, rotations, st)
    stream = _stream()
    Tx, Ty = (W + 15) // 16, (H + 15) // 16
    tiles = Tx * Ty
    i32, u8, f32 = torch.int32, torch.uint8, torch.float32
    nq = max(nv * P, 1)
    rec = torch.empty(nq * 12, dtype=f32, device=dev)
    radii = torch.empty(nq, dtype=i32, device=dev)
    rect = torch.empty(nq * 2, dtype=i32, device=dev)
    clamped = torch.empty(nq, dtype=u8, device=dev) if shs is not None else None
    nbx = L.sgr_preprocess_blocks_per_view(P)
    block_offsets = torch.empty(2 * (nbx * nv + 1), dtype=i32, device=dev)
    num_rendered = torch.zeros(2, dtype=torch.int64, device=dev)
    color = torch.empty(nv, 3, H, W, dtype=f32, device=dev)
    depth = torch.empty(nv, 1, H, W, dtype=f32, device=dev)
    alpha = torch.empty(nv, 1, H, W, dtype=f32, device=dev)
    final_T = torch.empty(nv, H, W, dtype=f32, device=dev)
    n_contrib = torch.empty(nv, H, W, dtype=i32, device=dev)
    ranges = torch.empty(nv * tiles * 2, dtype=i32, device=dev)
    R = 0
    if P > 0:
        _cabi.check(L.sgr_preprocess_forward(C.byref(pb), _ptr(rec), _ptr(radii), _ptr(rect), _ptr(clamped),
                                             _ptr(block_offsets), _ptr(num_rendered), stream), "sgr_preprocess_forward")
        nr = num_rendered.tolist()              # the ONE device->host sync of a batched forward (upstream: one per view)
        R = int(nr[0])
        if nr[1] != 0:
            raise RuntimeError(f"num_rendered {R} exceeds the 32-bit instance index")
    keys_a = torch.empty(max(R, 1), dtype=torch.int64, device=dev)
    keys_b = torch.empty(max(R, 1), dtype=torch.int64, device=dev)
    vals_a = torch.empty(max(R, 1), dtype=i32, device=dev)
    vals_b = torch.empty(max(R, 1), dtype=i32, device=dev)
    ws_bytes = L.sgr_bin_workspace_bytes(R)
    ws = torch.empty(ws_bytes, dtype=u8, device=dev)
    in_b = C.c_int32(0)
    _cabi.check(L.sgr_bin(C.byref(pb), _ptr(rec), _ptr(radii), _ptr(rect), _ptr(block_offsets), R, _ptr(keys_a), _ptr(keys_b),
                          _ptr(vals_a), _ptr(vals_b), _ptr(ws), ws_bytes, _ptr(ranges), C.byref(in_b), stream), "sgr_bin")
    point_list = vals_b if in_b.value else vals_a
    keys = keys_b if in_b.value else keys_a
    aux = None
    if need_ctx and not keep_keys and R > 0 and not _USE_BWD_V1:
        NS = L.sgr_bucket_slots(R, nv * tiles)
        aux = (torch.empty(4 * R * 2, dtype=i32, device=dev), torch.empty(4 * NS * 64 * 4, dtype=f32, device=dev),
               torch.empty(4 * NS * 64 * 2, dtype=f32, device=dev), torch.empty(4 * NS * 2, dtype=i32, device=dev))
    ax = aux if aux is not None else (None, None, None, None)
    _cabi.check(L.sgr_render_forward(C.byref(pb), _ptr(ranges), _ptr(point_list), _ptr(rec), _ptr(color), _ptr(depth),
                                     _ptr(alpha), _ptr(final_T), _ptr(n_contrib), R, _ptr(ax[0]), _ptr(ax[1]), _ptr(ax[2]),
                                     _ptr(ax[3]), stream), "sgr_render_forward")
    radii_out = radii[: nv * P].view(nv, P)
    ctx = None
    if need_ctx:
        ctx = _Ctx()
        ctx.pb = pb
        ctx.rec, ctx.radii, ctx.rect, ctx.clamped = rec, radii, rect, clamped
        ctx.point_list, ctx.ranges, ctx.final_T, ctx.n_contrib = point_list, ranges, final_T, n_contrib
        ctx.keys = keys if keep_keys else None
        ctx.num_rendered = R
        ctx.dims = (S, P, nv, H, W)
        ctx.keep = (st.viewmatrix, st.projmatrix, st.campos, st.bg)
        ctx.aux = aux
        ctx.images = (color, depth, alpha)
    return color, radii_out, depth, alpha, ctx


def _backward_impl(ctx: _Ctx, means3D, opacities, colors_precomp, shs, cov3D_precomp, scales, rotations,
                   st: BatchedRasterizationSettings, grad_color, grad_depth, grad_alpha):
    L = _cabi.lib()
    S, P, nv, H, W = ctx.dims
    dev = means3D.device
    f32 = torch.float32
    pb = _make_problem(means3D, opacities, colors_precomp, shs, cov3D_precomp, scales, rotations, st)
    stream = _stream()
    grec = torch.empty(max(nv * P, 1) * 12, dtype=f32, device=dev)
    gC = _f32c(grad_color)
    gD = None if grad_depth is None else _f32c(grad_depth)
    gA = None if grad_alpha is None else _f32c(grad_alpha)
    ax = ctx.aux if ctx.aux is not None else (None, None, None, None)
    img = ctx.images
    _cabi.check(L.sgr_render_backward(C.byref(pb), _ptr(ctx.ranges), _ptr(ctx.point_list), _ptr(ctx.rec), _ptr(ctx.final_T),
                                      _ptr(ctx.n_contrib), _ptr(img[0]), _ptr(img[1]), _ptr(img[2]), _ptr(gC), _ptr(gD), _ptr(gA),
                                      ctx.num_rendered, _ptr(ax[0]), _ptr(ax[1]), _ptr(ax[2]), _ptr(ax[3]), _ptr(grec), stream),
                "sgr_render_backward")
    d_means3D = torch.empty(S, P, 3, dtype=f32, device=dev)
    d_means2D = torch.empty(nv, P, 3, dtype=f32, device=dev)
    d_op = torch.empty(S, P, dtype=f32, device=dev)
    d_cov = torch.empty(S, P, 6, dtype=f32, device=dev)
    d_col = torch.empty(S, P, 3, dtype=f32, device=dev) if shs is None else None
    d_sh = torch.empty_like(shs) if shs is not None else None
    d_sc = torch.empty(S, P, 3, dtype=f32, device=dev) if scales is not None else None
    d_rot = torch.empty(S, P, 4, dtype=f32, device=dev) if scales is not None else None
    _cabi.check(L.sgr_preprocess_backward(C.byref(pb), _ptr(ctx.radii), _ptr(ctx.clamped), _ptr(grec), _ptr(d_means3D),
                                          _ptr(d_means2D), _ptr(d_op), _ptr(d_col), _ptr(d_sh), _ptr(d_cov), _ptr(d_sc),
                                          _ptr(d_rot), stream), "sgr_preprocess_backward")
    return d_means3D, d_means2D, d_sh, d_col, d_op, d_sc, d_rot, d_cov, grec


def _fwd_common(ctx, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, st):
    # fp32-only op: inputs are cast here, so an enclosing autocast region (gs.py:98) cannot downcast them
    opt = lambda t: None if t is None or t.numel() == 0 else _f32c(t)
    means3D = _f32c(means3D)
    opacities = _f32c(opacities).reshape(means3D.shape[0], means3D.shape[1])
    sh, colors_precomp, scales, rotations, cov3Ds_precomp = map(opt, (sh, colors_precomp, scales, rotations, cov3Ds_precomp))
    st = st._replace(viewmatrix=_f32c(st.viewmatrix), projmatrix=_f32c(st.projmatrix), campos=_f32c(st.campos), bg=_f32c(st.bg))
    color, radii, depth, alpha, c = _forward_impl(means3D, opacities, colors_precomp, sh, cov3Ds_precomp, scales, rotations,
                                                  st, need_ctx=True)
    ctx.sgr = c
    ctx.st = st
    ctx.has = (sh is not None, colors_precomp is not None, scales is not None, cov3Ds_precomp is not None)
    ctx.save_for_backward(means3D, opacities, colors_precomp, sh, cov3Ds_precomp, scales, rotations)
    return color, radii, depth, alpha


def _bwd_common(ctx, grad_color, grad_depth, grad_alpha):
    means3D, opacities, colors_precomp, sh, cov3Ds_precomp, scales, rotations = ctx.saved_tensors
    d_means3D, d_means2D, d_sh, d_col, d_op, d_sc, d_rot, d_cov, _ = _backward_impl(
        ctx.sgr, means3D, opacities, colors_precomp, sh, cov3Ds_precomp, scales, rotations, ctx.st, grad_color, grad_depth,
        grad_alpha)
    has_sh, has_col, has_sr, has_cov = ctx.has
    return (d_means3D, d_means2D if ctx.has_means2D else None, d_sh if has_sh else None, d_col if has_col else None, d_op.unsqueeze(-1),
            d_sc if has_sr else None, d_rot if has_sr else None, d_cov if has_cov else None)


class _RasterizeGaussiansBatched(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, st):
        color, radii, depth, alpha = _fwd_common(ctx, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, st)
        ctx.has_means2D = means2D is not None
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
    return _RasterizeGaussiansBatched.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                            raster_settings)


class _RasterizeGaussians(torch.autograd.Function):
    """Single-view op with upstream's exact signature (9 forward args, 4 outputs, 9 grads)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        if means3D.ndim != 2 or means3D.shape[1] != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")
        rs = raster_settings
        st = BatchedRasterizationSettings(rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, rs.bg, rs.scale_modifier,
                                          rs.viewmatrix.reshape(1, 4, 4), rs.projmatrix.reshape(1, 4, 4), rs.sh_degree,
                                          rs.campos.reshape(1, 3), 1, rs.debug)
        u = lambda t: None if t is None or t.numel() == 0 else t.unsqueeze(0)
        P = means3D.shape[0]
        color, radii, depth, alpha = _fwd_common(ctx, means3D.unsqueeze(0), u(sh), u(colors_precomp), opacities.reshape(1, P, 1),
                                                 u(scales), u(rotations), u(cov3Ds_precomp), st)
        color, radii, depth, alpha = color[0], radii[0], depth[0], alpha[0]
        ctx.has_means2D = means2D is not None
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        ub = lambda t: None if t is None else t.unsqueeze(0)
        g = _bwd_common(ctx, grad_color.unsqueeze(0), ub(grad_depth), ub(grad_alpha))
        return tuple(None if x is None else x[0] for x in g) + (None,)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings)


def mark_visible(positions: torch.Tensor, viewmatrix: torch.Tensor) -> torch.Tensor:
    L = _cabi.lib()
    positions = _f32c(positions)
    out = torch.zeros(positions.shape[0], dtype=torch.uint8, device=positions.device)
    vm = _f32c(viewmatrix)
    _cabi.check(L.sgr_mark_visible(positions.shape[0], _ptr(positions), _ptr(vm), _ptr(out), _stream()), "sgr_mark_visible")
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
        e = torch.Tensor([])
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
        color, radii, depth, alpha, c = _forward_impl(_f32c(means3D), _f32c(opacities).reshape(S, P), opt(colors_precomp),
                                                      opt(shs), opt(cov3D_precomp), opt(scales), opt(rotations), st,
                                                      need_ctx=True, keep_keys=True)
        nv = st.viewmatrix.shape[0]
        R = c.num_rendered
        return dict(color=color, radii=radii, depth=depth, alpha=alpha, rec=c.rec[: nv * P * 12].view(nv, P, 12),
                    rect=c.rect[: nv * P * 2].view(nv, P, 2), clamped=c.clamped, point_list=c.point_list[:R],
                    keys=c.keys[:R], ranges=c.ranges.view(nv, -1, 2), final_T=c.final_T, n_contrib=c.n_contrib,
                    num_rendered=R, ctx=c, settings=st)
