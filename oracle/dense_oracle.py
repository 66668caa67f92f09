"""Dense autograd oracle: an INDEPENDENT O(P*H*W) PyTorch evaluation of the splatting algorithm.

TEST INFRASTRUCTURE ONLY (tests/ imports it; the product never does).  Parity status: "parity
unpinned" at the reference level -- the third-party rasterizer bound at
/root/reference/core/gaussians/gs.py:82-106 is absent from /root/reference and has no golden vectors.
This module is what pins oracle/gsplat_ref.c instead: it shares NO code with it, evaluates every
Gaussian at every pixel in fp64 (or fp32), orders by (depth bits, index), composites with the
published discrete rules expressed as masks, and obtains every gradient from torch.autograd --
so the hand-derived backward formulas of gsplat_ref.c / the HIP kernels are checked against the
chain rule itself.

Discrete rules: cull z<=0.2; cov2D += 0.3 I; det==0 skip; radius =
ceil(3*sqrt(max eig)) with the 0.1 floor; 16x16 tile rectangle membership; power>0 skip;
alpha = min(0.99, op*exp(power)) with a STRAIGHT-THROUGH gradient (upstream differentiates
through op*G even when capped); alpha<1/255 skip; stop before the Gaussian that would push
T below 1e-4.  The 1.3*tanfov clamp passes gradient only where inactive (x_grad_mul).
"""
from __future__ import annotations

import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


def _sh_color(deg, sh, dirs):
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6]
               + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
               + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
               + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
               + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return torch.clamp_min(res + 0.5, 0.0)


def _quat_to_rot(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
        torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
        torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], 1)


def render(means3D, opacities, *, colors_precomp=None, shs=None, cov3D_precomp=None, scales=None, rotations=None,
           means2D=None, viewmatrix, projmatrix, campos, bg, tanfovx, tanfovy, image_height, image_width,
           scale_modifier=1.0, sh_degree=0, dtype=torch.float64):
    """-> dict(color[3,H,W], depth[1,H,W], alpha[1,H,W], radii[P], n_contrib[H,W], margin).

    All tensor inputs may require grad.  `means2D` ([P,3], zeros) is the dummy screen-space tensor whose
    gradient upstream reports (d L / d NDC).  `margin` = smallest distance of any discrete comparison from
    its threshold (a value < ~1e-6 means an fp32 implementation may legitimately flip that decision).
    """
    t = lambda a: None if a is None else torch.as_tensor(a).to(dtype)
    means3D, opacities = t(means3D).reshape(-1, 3), t(opacities).reshape(-1)
    P = means3D.shape[0]
    H, W = int(image_height), int(image_width)
    V = t(viewmatrix).reshape(4, 4)     # memory order of the reference tensor: V[c][r] = w2c[r][c]
    M = t(projmatrix).reshape(4, 4)
    campos, bg = t(campos).reshape(3), t(bg).reshape(3)
    w2c, full = V.T, M.T
    ones = torch.ones(P, 1, dtype=dtype)
    ph = torch.cat([means3D, ones], 1)
    pview = ph @ w2c.T                                   # [P,4]
    tz = pview[:, 2]
    hom = ph @ full.T
    pw = 1.0 / (hom[:, 3] + 1e-7)
    ndc = hom[:, :2] * pw[:, None]
    if means2D is not None:
        ndc = ndc + t(means2D)[:, :2]
    pix = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], 1)
    # ---- 3D covariance
    if cov3D_precomp is not None:
        c6 = t(cov3D_precomp).reshape(P, 6)
        S = torch.stack([torch.stack([c6[:, 0], c6[:, 1], c6[:, 2]], -1),
                         torch.stack([c6[:, 1], c6[:, 3], c6[:, 4]], -1),
                         torch.stack([c6[:, 2], c6[:, 4], c6[:, 5]], -1)], 1)
    else:
        R = _quat_to_rot(t(rotations).reshape(P, 4))
        s = scale_modifier * t(scales).reshape(P, 3)
        Mx = R * s[:, None, :]
        S = Mx @ Mx.transpose(1, 2)
    # ---- 2D covariance
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    safe_tz = torch.where(tz > 0.2, tz, torch.ones_like(tz))
    txtz, tytz = pview[:, 0] / safe_tz, pview[:, 1] / safe_tz
    cx_, cy_ = (txtz < -limx) | (txtz > limx), (tytz < -limy) | (tytz > limy)
    tx = torch.where(cx_, (txtz.clamp(-limx, limx) * safe_tz).detach(), pview[:, 0])
    ty = torch.where(cy_, (tytz.clamp(-limy, limy) * safe_tz).detach(), pview[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / safe_tz, zero, -(fx * tx) / (safe_tz * safe_tz)], -1),
                     torch.stack([zero, fy / safe_tz, -(fy * ty) / (safe_tz * safe_tz)], -1)], 1)   # [P,2,3]
    Mj = J @ w2c[:3, :3]                                                                            # [P,2,3]
    cov = Mj @ S @ Mj.transpose(1, 2)
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    ok = (tz > 0.2) & (det != 0)
    det_s = torch.where(ok, det, torch.ones_like(det))
    conx, cony, conz = c / det_s, -b / det_s, a / det_s
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    Tx, Ty = (W + 15) // 16, (H + 15) // 16
    pd = pix.detach()
    rminx = torch.trunc((pd[:, 0] - radius) / 16).clamp(0, Tx); rmaxx = torch.trunc((pd[:, 0] + radius + 15) / 16).clamp(0, Tx)
    rminy = torch.trunc((pd[:, 1] - radius) / 16).clamp(0, Ty); rmaxy = torch.trunc((pd[:, 1] + radius + 15) / 16).clamp(0, Ty)
    ok = ok & ((rmaxx - rminx) * (rmaxy - rminy) > 0)
    radii = torch.where(ok, radius, torch.zeros_like(radius)).to(torch.int32)
    # ---- colours
    if colors_precomp is not None:
        rgb = t(colors_precomp).reshape(P, 3)
    else:
        d = means3D - campos[None]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = _sh_color(sh_degree, t(shs).reshape(P, -1, 3), d)
    # ---- order: (fp32 depth bits, index) ascending == the published stable radix sort on float bits
    order = torch.argsort(tz.detach().to(torch.float32).to(torch.float64) * 1.0, stable=True)
    # ---- dense evaluation
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dtype), torch.arange(W, dtype=dtype), indexing="ij")
    tile_x, tile_y = torch.div(xs, 16, rounding_mode="floor"), torch.div(ys, 16, rounding_mode="floor")
    o = order
    dx = pix[o, 0][:, None, None] - xs[None]
    dy = pix[o, 1][:, None, None] - ys[None]
    power = -0.5 * (conx[o][:, None, None] * dx * dx + conz[o][:, None, None] * dy * dy) - cony[o][:, None, None] * dx * dy
    in_rect = ((tile_x[None] >= rminx[o][:, None, None]) & (tile_x[None] < rmaxx[o][:, None, None])
               & (tile_y[None] >= rminy[o][:, None, None]) & (tile_y[None] < rmaxy[o][:, None, None])
               & ok[o][:, None, None])
    a_raw = opacities[o][:, None, None] * torch.exp(torch.clamp_max(power, 0.0))
    alpha = a_raw + (torch.clamp_max(a_raw, 0.99) - a_raw).detach()          # straight-through cap
    valid = in_rect & (power <= 0) & (alpha.detach() >= 1.0 / 255.0)
    alpha_v = torch.where(valid, alpha, torch.zeros_like(alpha))
    one_m = 1.0 - alpha_v
    T_incl = torch.cumprod(one_m, 0)
    T_excl = torch.cat([torch.ones_like(T_incl[:1]), T_incl[:-1]], 0)
    keep = valid & (T_incl.detach() >= 1e-4)
    wgt = torch.where(keep, alpha_v * T_excl, torch.zeros_like(alpha_v))
    one_keep = torch.where(keep, one_m, torch.ones_like(one_m))
    T_final = torch.prod(one_keep, 0)
    color = torch.einsum("phw,pc->chw", wgt, rgb[o]) + T_final[None] * bg[:, None, None]
    depth = torch.einsum("phw,p->hw", wgt, tz[o])[None]
    alpha_img = wgt.sum(0)[None]
    # n_contrib: 1-based position (within the pixel's tile list) of the last kept Gaussian
    pos_in_list = torch.cumsum(in_rect.to(torch.int64), 0)
    n_contrib = torch.where(keep, pos_in_list, torch.zeros_like(pos_in_list)).max(0).values
    # ---- decision margins
    with torch.no_grad():
        m1 = (alpha.detach() - 1.0 / 255.0).abs()[in_rect & (power <= 0)]
        m2 = (T_incl - 1e-4).abs()[valid]
        m3 = power.abs()[in_rect]
        m4 = (tz - 0.2).abs()
        margin = min([x.min().item() if x.numel() else math.inf for x in (m1, m2, m3, m4)])
    return dict(color=color, depth=depth, alpha=alpha_img, radii=radii, n_contrib=n_contrib, margin=margin,
                rect=torch.stack([rminx, rminy, rmaxx, rmaxy], 1).to(torch.int32) * ok[:, None].to(torch.int32))
