"""ctypes binding of libsigman_gsplat.so (the C ABI declared in include/sigman_gsplat.h).

The product path has NO fallback: if the HIP library is missing or a call fails, this module raises.
PyTorch is used only as the owner of device memory and streams (data_ptr() / current stream).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SIGMAN_GSPLAT_LIB") or os.path.join(_HERE, "lib", "libsigman_gsplat.so")   # env override: dev A/B builds only
_lib = None

SGR_TILE = 16


class SgrProblem(C.Structure):
    _fields_ = [
        ("P", C.c_int32), ("n_views", C.c_int32), ("views_per_subject", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("sh_degree", C.c_int32), ("M", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("means3D", C.c_void_p), ("opacities", C.c_void_p), ("colors_precomp", C.c_void_p), ("shs", C.c_void_p),
        ("cov3D_precomp", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p),
        ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p), ("bg", C.c_void_p),
        ("color_clamped", C.c_void_p), ("clamp_grad", C.c_int32), ("reserved0", C.c_int32),      # optional: zero when not given
    ]


class SgrForwardState(C.Structure):
    _fields_ = [("R_alloc", C.c_uint64), ("true_rendered", C.c_uint64), ("NS", C.c_uint64), ("with_aux", C.c_int32), ("result_in_b", C.c_int32),
                ("flags_cleared", C.c_int32), ("aux_no_da", C.c_int32), ("fwd_kind", C.c_int32), ("nr_by_copy", C.c_int32),
                ("geom", C.c_void_p), ("binning", C.c_void_p), ("image", C.c_void_p),
                ("geom_bytes", C.c_uint64), ("binning_bytes", C.c_uint64), ("image_bytes", C.c_uint64)] + \
               [(n, C.c_uint64) for n in ("off_rec", "off_rect", "off_clamped", "off_block_offsets", "off_num_rendered", "off_keys_a",
                                          "off_keys_b", "off_vals_a", "off_vals_b", "off_sort_ws", "off_ranges", "off_final_T",
                                          "off_n_contrib", "off_compact", "off_ckpt_tc", "off_ckpt_da", "off_desc", "off_order", "off_flags",
                                          "off_part", "off_loss_part")] + \
               [("fused_bwd", C.c_int32), ("order_kind", C.c_int32), ("off_flags_fused", C.c_uint64)]


class SgrL1Epilogue(C.Structure):
    _fields_ = [("target", C.c_void_p), ("mask", C.c_void_p), ("grad_color", C.c_void_p), ("loss_per_view", C.c_void_p),
                ("loss_total", C.c_void_p), ("weight", C.c_float), ("sums_already_zero", C.c_int32), ("fuse_backward", C.c_int32), ("reserved0", C.c_int32)]


ABI_VERSION = 9          # include/sigman_gsplat.h: SGR_ABI_VERSION
ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int32, C.c_size_t)

_SIGNATURES = {
    "sgr_abi_version": (C.c_int, []),
    "sgr_last_error": (C.c_char_p, []),
    "sgr_preprocess_blocks_per_view": (C.c_int32, [C.c_int32]),
    "sgr_rasterize_forward": (C.c_int, [C.POINTER(SgrProblem), C.c_uint64, C.c_int32, ALLOC_FN, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                        C.POINTER(SgrForwardState), C.c_void_p]),
    "sgr_rasterize_forward_l1": (C.c_int, [C.POINTER(SgrProblem), C.c_uint64, C.c_int32, ALLOC_FN, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                           C.POINTER(SgrForwardState), C.POINTER(SgrL1Epilogue), C.c_void_p]),
    "sgr_rasterize_backward": (C.c_int, [C.POINTER(SgrProblem), C.POINTER(SgrForwardState)] + [C.c_void_p] * 8 + [ALLOC_FN, C.c_void_p]
                               + [C.c_void_p] * 9),
    "sgr_preprocess_forward": (C.c_int, [C.POINTER(SgrProblem)] + [C.c_void_p] * 6 + [C.c_uint64, C.c_void_p]),
    "sgr_bin_workspace_bytes": (C.c_size_t, [C.c_uint64, C.c_uint64]),
    "sgr_bin": (C.c_int, [C.POINTER(SgrProblem), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_int32),
                          C.c_void_p]),
    "sgr_bucket_slots": (C.c_uint64, [C.c_uint64, C.c_uint64]),
    "sgr_set_forward_mode": (C.c_int, [C.c_int]),
    "sgr_set_backward_gather": (C.c_int, [C.c_int]),
    "sgr_set_preprocess_view_group": (C.c_int, [C.c_int]),
    "sgr_set_keep_sorted_keys": (C.c_int, [C.c_int]),
    "sgr_set_fused_step": (C.c_int, [C.c_int]),
    "sgr_set_debug": (C.c_int, [C.c_int]),
    "sgr_set_sort_mode": (C.c_int, [C.c_int]),
    "sgr_set_sort_deep": (C.c_int, [C.c_int]),
    "sgr_render_forward": (C.c_int, [C.POINTER(SgrProblem)] + [C.c_void_p] * 8 + [C.c_uint64] + [C.c_void_p] * 6),
    "sgr_render_backward": (C.c_int, [C.POINTER(SgrProblem)] + [C.c_void_p] * 11 + [C.c_uint64] + [C.c_void_p] * 7),
    "sgr_preprocess_backward": (C.c_int, [C.POINTER(SgrProblem)] + [C.c_void_p] * 14),
    "sgr_mark_visible": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sgr_knn_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "sgr_knn_dist2": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    "sgr_knn_dist2_batched": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    "sgr_cov3d_forward": (C.c_int, [C.c_int32] + [C.c_void_p] * 5),
    "sgr_cov3d_backward": (C.c_int, [C.c_int32] + [C.c_void_p] * 7),
    "sgr_clamped_l1_loss": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "sgr_clock_probe": (C.c_int, [C.POINTER(C.c_double), C.c_void_p]),
    "sgr_prof_configure": (C.c_int, [C.c_uint32]),
    "sgr_prof_collect": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def lib():
    """Load the HIP library; raise loudly (no CPU fallback) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C sigman_release_amd/csrc`). There is no CPU fallback for the rasterizer.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.sgr_abi_version() != ABI_VERSION:
            raise RuntimeError(f"ABI version mismatch: library reports {L.sgr_abi_version()}, binding expects {ABI_VERSION}")
        # (the A/B knobs SIGMAN_SORT_MODE / SIGMAN_SORT_DEEP / SIGMAN_FWD_MODE are read by the library itself, by every thread at its first
        # use: the setters -- sgr_set_sort_mode & co -- are per thread, the environment is the process-wide default)
        _lib = L
    return _lib


_node = False


def torch_node():
    """The C++ autograd node of the single-view op (csrc/torch_node.cpp -> lib/sgr_torch_node.so), or None if it is not built or
    SIGMAN_PY_NODE=1 asks for the Python node (both drive the same C ABI; the C++ one only costs less host time per call)."""
    global _node
    if _node is False:
        _node = None
        path = os.path.join(_HERE, "lib", "sgr_torch_node.so")
        if os.environ.get("SIGMAN_PY_NODE", "0") != "1" and os.path.exists(path):
            lib()                                               # libsigman_gsplat.so first (the node links against it)
            import importlib.util
            spec = importlib.util.spec_from_file_location("sgr_torch_node", path)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            if mod.abi_version() != ABI_VERSION:
                raise RuntimeError("sgr_torch_node.so was built against another ABI version of libsigman_gsplat.so: rebuild (make -C sigman_release_amd/csrc)")
            _node = mod
    return _node


def check(status: int, what: str, why: str = None):
    if status != 0:
        msg = lib().sgr_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed: {msg}" + (f" ({why})" if why else ""))
