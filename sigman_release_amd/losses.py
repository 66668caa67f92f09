"""Fused image-space loss epilogue on the HIP path.

`clamped_l1_loss(color, target, mask=None, weight=1.0)` == `weight * ((color.clamp(0,1) - target) * mask).abs().sum()`
i.e. the reference's `rendered_image.clamp(0, 1)` (core/gaussians/gs.py:107) followed by the masked L1 of
core/loss/whole_loss.py:126-131, in one kernel that also emits dL/dcolor for the rasterizer backward and per-view
partial sums (the "image-space losses" the view-parallel mode all-reduces)."""
from __future__ import annotations

import torch

from . import _cabi
from .rasterizer import _f32c, _ptr, _stream


class _ClampedL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, target, mask, weight):
        L = _cabi.lib()
        if color.device.type != "cuda":
            raise RuntimeError("clamped_l1_loss needs ROCm device tensors (there is no CPU fallback)")
        color, target = _f32c(color), _f32c(target)
        mask = None if mask is None else _f32c(mask)
        nv, _, H, W = color.shape
        grad = torch.empty_like(color)
        sums = torch.empty(nv + 1, dtype=torch.float32, device=color.device)      # [per-view partial sums | total]
        p = sums.data_ptr()
        _cabi.check(L.sgr_clamped_l1_loss(nv, H, W, _ptr(color), _ptr(target), _ptr(mask), float(weight), _ptr(grad),
                                          p, p + 4 * nv, 0, _stream(color.device)), "sgr_clamped_l1_loss")
        ctx.save_for_backward(grad)
        ctx.per_view = sums[:nv]
        return sums[nv]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None


def clamped_l1_loss(color, target, mask=None, weight: float = 1.0):
    """color/target [n_views,3,H,W], mask [n_views,1,H,W] or None -> scalar."""
    return _ClampedL1.apply(color, target, mask, weight)
