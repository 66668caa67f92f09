"""CPU tests: the C-ABI library loads and exports every symbol include/sigman_gsplat.h declares (no compute calls),
and the product path fails loudly instead of falling back when it cannot run."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "sigman_gsplat.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sgr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    import ctypes
    from sigman_release_amd import _cabi
    L = ctypes.CDLL(_cabi.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/sigman_gsplat.h but not exported"
    assert _cabi.lib().sgr_abi_version() == _cabi.ABI_VERSION == 9
    assert _cabi.lib().sgr_preprocess_blocks_per_view(1000) == 4
    assert _cabi.lib().sgr_bin_workspace_bytes(10_000, 1024) >= 3 * 256 * 4


def test_binding_covers_the_header():
    from sigman_release_amd import _cabi
    missing = [n for n in _declared_symbols() if n not in _cabi.EXPORTED_SYMBOLS and not n.startswith("sgr_prof_")]
    assert not missing, f"ctypes binding lacks {missing}"


def test_no_cpu_fallback():
    """CPU tensors must raise, not silently render through some fallback."""
    from sigman_release_amd import rasterizer as R
    eye = torch.eye(4)
    rs = R.GaussianRasterizationSettings(16, 16, 0.5, 0.5, torch.ones(3), 1.0, eye, eye, 0, torch.zeros(3), False, False)
    m = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        R.GaussianRasterizer(rs)(means3D=m, means2D=m, opacities=torch.ones(4, 1), colors_precomp=torch.ones(4, 3),
                                 cov3D_precomp=torch.ones(4, 6))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sigman_release_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("dense autograd oracle", "").replace("CPU oracle", "") or f in (), \
                    f"{f} mentions the oracle package"


def test_ply_roundtrip_and_checkpoint_loader(tmp_path):
    """3DGS PLY I/O with the reference's semantics (gs.py:120-252): prune, inverse activations, property order, BGR flip."""
    import numpy as np
    import torch
    from sigman_release_amd import ply
    rng = np.random.default_rng(3)
    N = 500
    g = np.concatenate([rng.normal(size=(N, 3)), rng.uniform(0.0, 1.0, (N, 1)), rng.uniform(0.002, 0.05, (N, 3)),
                        rng.normal(size=(N, 4)), rng.uniform(0.05, 0.95, (N, 3))], 1).astype(np.float32)
    g[:7, 3] = 0.001                                              # below the 0.005 prune threshold
    g[7:, 3] = np.clip(g[7:, 3], 0.01, 0.99)
    t = torch.from_numpy(g)[None]
    for compatible in (True, False):
        path = str(tmp_path / f"g_{compatible}.ply")
        ply.save_ply(t, path, compatible)
        head = open(path, "rb").read(600).decode("ascii", "replace")
        names = [l.split()[-1] for l in head.split("end_header")[0].splitlines() if l.startswith("property")]
        assert names == ["x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2", "opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
        back = ply.load_ply(path, compatible).numpy()
        assert back.shape == (N - 7, 14)
        np.testing.assert_allclose(back, g[7:], rtol=2e-5, atol=2e-6)
    # a "training checkpoint" style file: pre-activation values + f_rest_*; ascii format on purpose
    raw = ply.read_vertex_ply(str(tmp_path / "g_True.ply"))
    cols = [(k, raw[k]) for k in ("x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2")] + [(f"f_rest_{i}", np.zeros(N - 7, np.float32)) for i in range(45)] + \
           [(k, raw[k]) for k in ("opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3")]
    p2 = str(tmp_path / "ckpt.ply")
    with open(p2, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment test\nelement vertex %d\n" % (N - 7) + "".join(f"property float {k}\n" for k, _ in cols) + "end_header\n")
        np.savetxt(f, np.stack([c for _, c in cols], 1), fmt="%.9g")
    ck = ply.load_gaussians_from_ply(p2).numpy()
    np.testing.assert_allclose(ck[:, :7], g[7:, :7], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(ck[:, 7:11], g[7:, 7:11] / np.linalg.norm(g[7:, 7:11], axis=1, keepdims=True), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(ck[:, 11:], g[7:, 11:][:, [2, 1, 0]], rtol=2e-5, atol=2e-6)


def test_ctypes_structs_match_the_c_header(tmp_path):
    """The ctypes mirror of SgrProblem / SgrForwardState / SgrL1Epilogue must have the C compiler's size and field offsets (no GPU needed)."""
    import ctypes as C
    import os
    import subprocess
    from sigman_release_amd import _cabi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fields = {"SgrProblem": [n for n, _ in _cabi.SgrProblem._fields_], "SgrForwardState": [n for n, _ in _cabi.SgrForwardState._fields_],
              "SgrL1Epilogue": [n for n, _ in _cabi.SgrL1Epilogue._fields_]}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "sigman_gsplat.h"', 'int main(void) {']
    for st, names in fields.items():
        src.append(f'printf("{st} %zu\\n", sizeof({st}));')
        src += [f'printf("{st}.{n} %zu\\n", offsetof({st}, {n}));' for n in names]
    src += ['return 0; }']
    c_file, exe = tmp_path / "abi.c", tmp_path / "abi"
    c_file.write_text("\n".join(src))
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(c_file), "-o", str(exe)])
    out = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for st in fields:
        cls = getattr(_cabi, st)
        assert int(out[st]) == C.sizeof(cls), st
        for n in fields[st]:
            assert int(out[f"{st}.{n}"]) == getattr(cls, n).offset, f"{st}.{n}"


def test_c_program_links_against_the_library(tmp_path):
    """A plain C translation unit including include/sigman_gsplat.h links against libsigman_gsplat.so and can call the entry points
    that need no GPU (the boundary really is a C ABI: no C++ names, no torch types)."""
    import os
    import subprocess
    from sigman_release_amd import _cabi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _cabi.lib()                                                    # raises with build instructions if the library is missing
    src = r'''
#include <stdio.h>
#include "sigman_gsplat.h"
int main(void) {
    SgrProblem pb; SgrForwardState st; (void)pb; (void)st;
    printf("%d %llu %llu\n", sgr_abi_version(), (unsigned long long)sgr_bucket_slots(1000, 16),
           (unsigned long long)sgr_bin_workspace_bytes(1000, 16));
    return sgr_last_error() == 0;
}
'''
    c_file, exe = tmp_path / "link.c", tmp_path / "link"
    c_file.write_text(src)
    libdir = os.path.dirname(_cabi.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(c_file), "-o", str(exe),
                           "-L", libdir, "-lsigman_gsplat", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe)], text=True).split()
    assert int(out[0]) == 9 and int(out[1]) == (1000 >> 6) + 16 + 1 and int(out[2]) > 0


def test_full_size_record_matches_kernel_sources():
    """tests/golden/full_size_observed.json (the recorded decision-flip counts the full-size GPU tests use as a ceiling) is bound to the
    kernel sources it was taken from: after any change of csrc/*.hip / common.h it must be re-recorded on the GPU
    (SIGMAN_RECORD_OBSERVED=1 pytest -m gpu -k full_size; tools/record_full_size_observed.py) -- this catches a stale record here, without a GPU."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("tgp", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_gpu_parity.py"))
    tgp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tgp)
    rec = json.load(open(tgp._OBSERVED_PATH))
    assert rec.get("_csrc_sha16") == tgp.kernel_sources_sha16(), "full_size_observed.json is stale: re-record it on the GPU box"
    for cfg in ("c2", "c3", "c4", "c5"):
        assert cfg in rec, cfg


def test_torch_node_module_loads_and_exports_both_ops():
    """lib/sgr_torch_node.so (csrc/torch_node.cpp): importable without a GPU, built against this ABI, exports the single-view op and the
    batched op with the fused L1 loss."""
    import os
    from sigman_release_amd import _cabi
    if os.environ.get("SIGMAN_PY_NODE", "0") == "1":
        pytest.skip("SIGMAN_PY_NODE=1")
    node = _cabi.torch_node()
    assert node is not None, "sgr_torch_node.so is not built (make -C sigman_release_amd/csrc)"
    assert node.abi_version() == _cabi.lib().sgr_abi_version()
    for name in ("rasterize_gaussians", "rasterize_l1_batched", "check_pending", "check_pending_batched", "set_count_check", "set_count_wait", "slot_stats"):
        assert callable(getattr(node, name)), name
    # the count-wait modes of the explicit-capacity batched nodes (host-only state): "own" | "lazy" | "lazy:N", N = 1..16; anything else is an error
    try:
        from sigman_release_amd import rasterizer as R
        for good in ("lazy", "lazy:1", "lazy:4", "lazy:16", "own"):
            node.set_count_wait(good)
            R.set_count_wait(good)                              # (the public wrapper)
        for bad in ("", "eager", "lazy:", "lazy:0", "lazy:17", "lazy:x", "lazy:4 ", "lazy:-1", "LAZY"):
            with pytest.raises(RuntimeError, match="set_count_wait"):
                node.set_count_wait(bad)
    finally:
        node.set_count_wait("own")


def test_allocator_callback_never_raises():
    """The ctypes allocator callback of the Python nodes must turn a failed allocation into NULL (an escaping exception would leave ctypes an
    uninitialised return value, i.e. hand the library a garbage device pointer): here the allocation fails because this box has no GPU."""
    import torch
    from sigman_release_amd import rasterizer as R
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU (the allocation has to fail)")
    R._alloc_target.dev, R._alloc_target.blobs, R._alloc_target.error = torch.device("cuda", 0), [None] * 4, None
    assert R._alloc_cb(None, 2, 1 << 20) == 0
    assert R._alloc_target.blobs[2] is None and "blob 2" in R._alloc_target.error
