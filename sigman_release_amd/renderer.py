"""`GaussianRenderer` with the interface of /root/reference/core/gaussians/gs.py:41-117, on the batched HIP path.

    GaussianRenderer(opt).render(gaussians, cam_view, cam_view_proj, cam_pos, bg_color=None, scale_modifier=0.5)
        gaussians: dict  position [B,P,3], opacity [B,P,1], scale [B,P,3] in (-1,1), cov3d [B,P,3,3], rgb [B,P,3]
        cam_view / cam_view_proj [B,V,4,4], cam_pos [B,V,3]
        -> {"image": [B,V,3,H,W] (clamped to [0,1], gs.py:107), "alpha": [B,V,1,H,W]}

What changes underneath (same numbers out):
  * simple_knn.distCUDA2 (gs.py:70)            -> `dist_cuda2` (HIP grid-hash exact 3-NN, sgr_knn_dist2), detached
  * get_covariance/strip_lowerdiag (gs.py:71-73) -> `covariance_from_scale_rotation` (one fused HIP kernel + backward)
  * the B x V Python loop (gs.py:62,75)         -> ONE batched launch chain over all B*V views
`scale_modifier` is accepted and ignored exactly like upstream ignores it on the cov3D_precomp path.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _cabi
from .rasterizer import BatchedRasterizationSettings, _f32c, _ptr, _stream, rasterize_gaussians_batched

_KNN_MAX_CELLS = (1 << 22) - 1      # upper bound of the uniform grid; the grid actually used is bounded by 16 cells per point


def dist_cuda2(points: torch.Tensor) -> torch.Tensor:
    """Drop-in for simple_knn._C.distCUDA2: [P,3] -> [P] mean squared distance to the 3 nearest other points.
    Also accepts a batch [B,P,3] -> [B,P] (all point sets in one launch sequence)."""
    L = _cabi.lib()
    if points.device.type != "cuda":
        raise RuntimeError("dist_cuda2 needs a ROCm device tensor (there is no CPU fallback)")
    pts = _f32c(points.detach())
    batched = pts.ndim == 3
    B, P = (pts.shape[0], pts.shape[1]) if batched else (1, pts.shape[0])
    out = torch.empty(B, P, dtype=torch.float32, device=pts.device)
    max_cells = min(_KNN_MAX_CELLS, max(16 * P, 4096))        # the cell counters are cleared every call: keep them proportional to P
    stride = (L.sgr_knn_workspace_bytes(P, max_cells) + 255) // 256 * 256
    ws = torch.empty(stride * B, dtype=torch.uint8, device=pts.device)
    _cabi.check(L.sgr_knn_dist2_batched(B, P, _ptr(pts), _ptr(out), _ptr(ws), stride * B, max_cells, _stream(pts.device)), "sgr_knn_dist2")
    return out if batched else out[0]


class _Cov3D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scale_raw, rotation, dist2):
        L = _cabi.lib()
        scale_raw, rotation, dist2 = _f32c(scale_raw), _f32c(rotation), _f32c(dist2)
        n = dist2.numel()
        cov = torch.empty(*dist2.shape, 6, dtype=torch.float32, device=dist2.device)
        _cabi.check(L.sgr_cov3d_forward(n, _ptr(scale_raw), _ptr(rotation), _ptr(dist2), _ptr(cov), _stream(cov.device)), "sgr_cov3d_forward")
        ctx.save_for_backward(scale_raw, rotation, dist2)
        return cov

    @staticmethod
    def backward(ctx, g):
        L = _cabi.lib()
        scale_raw, rotation, dist2 = ctx.saved_tensors
        g = _f32c(g)
        gs, gr = torch.empty_like(scale_raw), torch.empty_like(rotation)
        _cabi.check(L.sgr_cov3d_backward(dist2.numel(), _ptr(scale_raw), _ptr(rotation), _ptr(dist2), _ptr(g), _ptr(gs), _ptr(gr),
                                         _stream(g.device)), "sgr_cov3d_backward")
        return gs, gr, None


def covariance_from_scale_rotation(scale_raw, rotation, dist2):
    """[...,3], [...,3,3], [...] -> [...,6]: strip_lowerdiag(R diag(((s+1)*sqrt(max(dist2,1e-7)))^2) R^T)."""
    if scale_raw.device.type != "cuda":
        raise RuntimeError("covariance_from_scale_rotation needs ROCm device tensors (there is no CPU fallback)")
    return _Cov3D.apply(scale_raw, rotation, dist2)


class GaussianRenderer:
    def __init__(self, opt, device="cuda"):
        self.opt = opt
        self.bg_color = torch.tensor([1, 1, 1], dtype=torch.float32, device=device)
        self.tan_half_fov = float(np.tan(0.5 * self.opt.FoVy))

    # 3DGS PLY I/O with the reference's semantics (gs.py:120-252); implementation in ply.py
    def save_ply(self, gaussians, path, compatible=True):
        from . import ply
        ply.save_ply(gaussians, path, compatible)

    def load_ply(self, path, compatible=True):
        from . import ply
        return ply.load_ply(path, compatible)

    def load_gaussians_from_ply(self, path):
        from . import ply
        return ply.load_gaussians_from_ply(path)

    def render(self, gaussians, cam_view, cam_view_proj, cam_pos, bg_color=None, scale_modifier=0.5):
        B, V = cam_view.shape[:2]
        H, W = self.opt.output_size_h, self.opt.output_size_w
        node = _cabi.torch_node() if gaussians["position"].device.type == "cuda" and gaussians["position"].shape[1] > 0 else None
        if node is not None:
            # everything below as ONE C++ autograd node above the C ABI (csrc/torch_node.cpp, RenderBatchedNode: 3-NN, covariance build,
            # batched rasterizer with the automatic capacity, clamp; backward: clamp mask, rasterizer, covariance): same numbers, one node
            # issued from the interpreter instead of four
            bg = self.bg_color if bg_color is None else bg_color
            image, _radii, _depth, alpha = node.render_batched(gaussians["position"], gaussians["rgb"], gaussians["opacity"], gaussians["scale"],
                                                               gaussians["cov3d"], cam_view.reshape(B * V, 4, 4), cam_view_proj.reshape(B * V, 4, 4),
                                                               cam_pos.reshape(B * V, 3), bg, int(H), int(W), self.tan_half_fov, self.tan_half_fov,
                                                               float(scale_modifier), int(V), -1)
            return {"image": image.view(B, V, 3, H, W), "alpha": alpha.view(B, V, 1, H, W)}
        position = gaussians["position"].float()
        P = position.shape[1]
        with torch.no_grad():
            dist2 = dist_cuda2(position)                                               # [B,P], detached (gs.py:70-71), one batched launch
        cov3D = covariance_from_scale_rotation(gaussians["scale"].float(), gaussians["cov3d"].float(), dist2)   # [B,P,6]
        st = BatchedRasterizationSettings(H, W, self.tan_half_fov, self.tan_half_fov,
                                          self.bg_color if bg_color is None else bg_color, scale_modifier,
                                          cam_view.reshape(B * V, 4, 4), cam_view_proj.reshape(B * V, 4, 4), 0,
                                          cam_pos.reshape(B * V, 3), V, False, -1)      # max_rendered = -1: automatic sync-free mode
        color, radii, depth, alpha = rasterize_gaussians_batched(position, None, None, gaussians["rgb"].float(),
                                                                 gaussians["opacity"].float(), None, None, cov3D, st)
        return {"image": color.clamp(0, 1).view(B, V, 3, H, W), "alpha": alpha.view(B, V, 1, H, W)}
