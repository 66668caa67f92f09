"""ctypes/numpy front end of the CPU oracle (oracle/gsplat_ref.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing under sigman_release_amd/ imports this module.  Parity status: see gsplat_ref.c
header ("parity unpinned" at the reference level; pinned by oracle/dense_oracle.py + golden vectors).

Mirrors the call shape of the third-party rasterizer the reference binds at
/root/reference/core/gaussians/gs.py:82-106: one view per call, inputs
(means3D, opacities, colors_precomp|shs, cov3D_precomp|scales+rotations) + camera settings,
outputs (color[3,H,W], radii[P], depth[1,H,W], alpha[1,H,W]) plus every intermediate integer
artefact (rect, tiles_touched, sorted keys, point_list, ranges, n_contrib) for bit-exact checks.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgsplat_ref.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds). Safe to call repeatedly."""
    src = os.path.join(_HERE, "gsplat_ref.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libgsplat_ref.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _RefCam(C.Structure):
    _fields_ = [("P", C.c_int), ("H", C.c_int), ("W", C.c_int), ("sh_degree", C.c_int), ("M", C.c_int),
                ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
                ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
                ("bg", C.c_void_p)]


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.ref_bin.restype = C.c_int64
    return _lib


def set_threads(n: int) -> None:
    """OpenMP team size of the CALLING Python thread's later oracle calls (test harnesses that run several calls side by side)."""
    lib().ref_set_num_threads(int(n))


def set_alpha_mode(mode: int) -> int:
    """0 = reproducible exponent (explicit FMA chain on the pre-scaled conic) + correctly rounded exp2: the default, what the HIP kernels
    are pinned against; 1 = the published expression left to right without FMA + libm expf (gsplat_ref.c header).  Returns the old mode."""
    L = lib()
    old = int(L.ref_get_alpha_mode())
    L.ref_set_alpha_mode(int(mode))
    return old


def exp2_cr(x) -> np.ndarray:
    """ref_exp2_cr over an array: 2^x rounded to fp32 from the fp64 polynomial both sides share step for step."""
    x = np.ascontiguousarray(np.asarray(x, np.float32)).reshape(-1)
    y = np.empty_like(x)
    lib().ref_exp2_cr_array(C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data), C.c_int64(x.size))
    return y


def alpha_threshold(opacities) -> np.ndarray:
    """p*(opacity): the smallest fp32 exponent (exp2 domain, <= 0) that passes the published alpha test, +inf if none does --
    the value the HIP preprocess kernel stores in float 11 of every record (gsplat_ref.c, ref_alpha_threshold)."""
    op = np.ascontiguousarray(np.asarray(opacities, np.float32)).reshape(-1)
    out = np.empty_like(op)
    lib().ref_alpha_threshold_array(C.c_void_p(op.ctypes.data), C.c_void_p(out.ctypes.data), C.c_int64(op.size))
    return out


def alpha_test(opacities, p) -> np.ndarray:
    """The alpha test evaluated directly, as the renderer does: p <= 0 and min(0.99, op * 2^p) >= 1/255."""
    op = np.ascontiguousarray(np.asarray(opacities, np.float32)).reshape(-1)
    p = np.ascontiguousarray(np.asarray(p, np.float32)).reshape(-1)
    out = np.empty(op.size, np.uint8)
    lib().ref_alpha_test_array(C.c_void_p(op.ctypes.data), C.c_void_p(p.ctypes.data), C.c_void_p(out.ctypes.data), C.c_int64(op.size))
    return out.astype(bool)


def _f32(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if shape is not None:
        a = a.reshape(shape)
    return a


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@dataclass
class RefState:
    """Everything the forward produced; the backward needs it (== upstream geom/binning/img buffers)."""
    P: int
    H: int
    W: int
    cam: dict
    means3D: np.ndarray
    opacities: np.ndarray
    scales: Optional[np.ndarray]
    rotations: Optional[np.ndarray]
    shs: Optional[np.ndarray]
    depths: np.ndarray = None
    xy: np.ndarray = None
    conic_opacity: np.ndarray = None
    radii: np.ndarray = None
    rect: np.ndarray = None
    tiles_touched: np.ndarray = None
    cov3D: np.ndarray = None
    rgb: np.ndarray = None
    clamped: np.ndarray = None
    point_offsets: np.ndarray = None
    keys: np.ndarray = None
    point_list: np.ndarray = None
    ranges: np.ndarray = None
    R: int = 0
    color: np.ndarray = None
    depth: np.ndarray = None
    alpha: np.ndarray = None
    final_T: np.ndarray = None
    n_contrib: np.ndarray = None
    _keep: list = field(default_factory=list)


def _mk_cam(P, H, W, tanfovx, tanfovy, viewmatrix, projmatrix, campos, bg, scale_modifier, sh_degree, M):
    vm = _f32(viewmatrix).reshape(16)
    pm = _f32(projmatrix).reshape(16)
    cp = _f32(campos).reshape(3)
    bgc = _f32(bg).reshape(3)
    cam = _RefCam(P, H, W, sh_degree, M, float(tanfovx), float(tanfovy), float(scale_modifier),
                  _p(vm), _p(pm), _p(cp), _p(bgc))
    return cam, [vm, pm, cp, bgc]


def forward(means3D, opacities, *, colors_precomp=None, shs=None, cov3D_precomp=None, scales=None,
            rotations=None, viewmatrix, projmatrix, campos, bg, tanfovx, tanfovy, image_height,
            image_width, scale_modifier=1.0, sh_degree=0, render=True) -> RefState:
    L = lib()
    means3D = _f32(means3D).reshape(-1, 3)
    P = means3D.shape[0]
    H, W = int(image_height), int(image_width)
    opacities = _f32(opacities).reshape(P)
    if (colors_precomp is None) == (shs is None):
        raise ValueError("provide exactly one of shs / colors_precomp")
    if (cov3D_precomp is None) == (scales is None or rotations is None):
        raise ValueError("provide exactly one of (scales, rotations) / cov3D_precomp")
    colors_precomp = None if colors_precomp is None else _f32(colors_precomp).reshape(P, 3)
    M = 0
    if shs is not None:
        shs = _f32(shs).reshape(P, -1, 3)
        M = shs.shape[1]
    cov3D_precomp = None if cov3D_precomp is None else _f32(cov3D_precomp).reshape(P, 6)
    scales = None if scales is None else _f32(scales).reshape(P, 3)
    rotations = None if rotations is None else _f32(rotations).reshape(P, 4)
    camd = dict(tanfovx=tanfovx, tanfovy=tanfovy, viewmatrix=_f32(viewmatrix).reshape(16),
                projmatrix=_f32(projmatrix).reshape(16), campos=_f32(campos).reshape(3), bg=_f32(bg).reshape(3),
                scale_modifier=scale_modifier, sh_degree=sh_degree, M=M)
    cam, keep = _mk_cam(P, H, W, tanfovx, tanfovy, viewmatrix, projmatrix, campos, bg, scale_modifier, sh_degree, M)
    st = RefState(P=P, H=H, W=W, cam=camd, means3D=means3D, opacities=opacities, scales=scales,
                  rotations=rotations, shs=shs)
    n = max(P, 1)
    st.depths = np.zeros(n, np.float32); st.xy = np.zeros((n, 2), np.float32)
    st.conic_opacity = np.zeros((n, 4), np.float32); st.radii = np.zeros(n, np.int32)
    st.rect = np.zeros((n, 4), np.int32); st.tiles_touched = np.zeros(n, np.uint32)
    st.cov3D = np.zeros((n, 6), np.float32); st.rgb = np.zeros((n, 3), np.float32)
    st.clamped = np.zeros((n, 3), np.uint8)
    L.ref_preprocess(C.byref(cam), _p(means3D), _p(opacities), _p(cov3D_precomp), _p(scales), _p(rotations),
                     _p(colors_precomp), _p(shs), _p(st.depths), _p(st.xy), _p(st.conic_opacity), _p(st.radii),
                     _p(st.rect), _p(st.tiles_touched), _p(st.cov3D), _p(st.rgb), _p(st.clamped))
    Tx, Ty = (W + 15) // 16, (H + 15) // 16
    Rtot = int(st.tiles_touched[:P].astype(np.int64).sum())
    st.point_offsets = np.zeros(n, np.uint32)
    st.keys = np.zeros(max(Rtot, 1), np.uint64); st.point_list = np.zeros(max(Rtot, 1), np.uint32)
    st.ranges = np.zeros((Tx * Ty, 2), np.uint32)
    R = L.ref_bin(P, H, W, _p(st.radii), _p(st.rect), _p(st.depths), _p(st.tiles_touched), _p(st.point_offsets),
                  _p(st.keys), _p(st.point_list), _p(st.ranges))
    assert R == Rtot
    st.R = int(R)
    st.keys = st.keys[:R]; st.point_list = st.point_list[:R]
    for k in ("depths", "xy", "conic_opacity", "radii", "rect", "tiles_touched", "cov3D", "rgb", "clamped",
              "point_offsets"):
        setattr(st, k, getattr(st, k)[:P])
    if render:
        st.color = np.zeros((3, H, W), np.float32); st.depth = np.zeros((1, H, W), np.float32)
        st.alpha = np.zeros((1, H, W), np.float32); st.final_T = np.zeros((H, W), np.float32)
        st.n_contrib = np.zeros((H, W), np.uint32)
        pl = np.ascontiguousarray(st.point_list) if R else np.zeros(1, np.uint32)
        L.ref_render_fwd(H, W, _p(st.ranges), _p(pl), _p(_c(st.xy)), _p(_c(st.conic_opacity)), _p(_c(st.rgb)),
                         _p(_c(st.depths)), _p(camd["bg"]), _p(st.color), _p(st.depth), _p(st.alpha),
                         _p(st.final_T), _p(st.n_contrib))
    return st


def _c(a):
    a = np.ascontiguousarray(a)
    return a if a.size else np.zeros(1, a.dtype)


def backward(st: RefState, grad_color, grad_depth=None, grad_alpha=None) -> dict:
    """Returns grads in the published order/naming: means3D, means2D[P,3], colors/sh, opacities, cov3D | scales, rotations."""
    L = lib()
    P, H, W = st.P, st.H, st.W
    n = max(P, 1)
    gC = _f32(grad_color).reshape(3, H, W)
    gD = None if grad_depth is None else _f32(grad_depth).reshape(H, W)
    gA = None if grad_alpha is None else _f32(grad_alpha).reshape(H, W)
    dmean2D = np.zeros((n, 2), np.float32); dconic = np.zeros((n, 3), np.float32)
    dop = np.zeros(n, np.float32); dcol = np.zeros((n, 3), np.float32); ddep = np.zeros(n, np.float32)
    pl = _c(st.point_list)
    L.ref_render_bwd(P, H, W, C.c_int64(st.R), _p(st.ranges), _p(pl), _p(_c(st.xy)), _p(_c(st.conic_opacity)),
                     _p(_c(st.rgb)), _p(_c(st.depths)), _p(st.cam["bg"]), _p(st.final_T), _p(st.n_contrib),
                     _p(gC), _p(gD), _p(gA), _p(dmean2D), _p(dconic), _p(dop), _p(dcol), _p(ddep))
    c = st.cam
    cam, keep = _mk_cam(P, H, W, c["tanfovx"], c["tanfovy"], c["viewmatrix"], c["projmatrix"], c["campos"], c["bg"],
                        c["scale_modifier"], c["sh_degree"], c["M"])
    dmeans = np.zeros((n, 3), np.float32); dcov = np.zeros((n, 6), np.float32)
    dsh = np.zeros((n, max(c["M"], 1), 3), np.float32) if st.shs is not None else None
    dsc = np.zeros((n, 3), np.float32) if st.scales is not None else None
    drot = np.zeros((n, 4), np.float32) if st.scales is not None else None
    L.ref_preprocess_bwd(C.byref(cam), _p(_c(st.means3D)), _p(_c(st.radii)), _p(_c(st.cov3D)), _p(st.scales),
                         _p(st.rotations), _p(st.shs), _p(_c(st.clamped)), _p(dmean2D), _p(dconic), _p(dcol), _p(ddep),
                         _p(dmeans), _p(dcov), _p(dsh), _p(dsc), _p(drot))
    means2D = np.zeros((P, 3), np.float32); means2D[:, :2] = dmean2D[:P]
    out = dict(means3D=dmeans[:P], means2D=means2D, opacities=dop[:P].reshape(P, 1), conic=dconic[:P], depth=ddep[:P])
    if st.shs is not None:
        out["sh"] = dsh[:P]
    else:
        out["colors_precomp"] = dcol[:P]
    if st.scales is not None:
        out["scales"] = dsc[:P]; out["rotations"] = drot[:P]
    else:
        out["cov3D_precomp"] = dcov[:P]
    out["colors_internal"] = dcol[:P]
    out["cov3D_internal"] = dcov[:P]
    return out


def mark_visible(means3D, viewmatrix) -> np.ndarray:
    L = lib()
    m = _f32(means3D).reshape(-1, 3)
    out = np.zeros(max(m.shape[0], 1), np.uint8)
    L.ref_mark_visible(m.shape[0], _p(_c(m)), _p(_f32(viewmatrix).reshape(16)), _p(out))
    return out[: m.shape[0]].astype(bool)
