#!/usr/bin/env python3
"""Compile the csrc/*.hip files to gfx950 assembly and list, per kernel, the number of global loads and of s_waitcnt vmcnt instructions.
A kernel whose vmcnt(0) waits are about as many as its loads issues them one round trip at a time -- usually a guarded load inside an
unrolled loop (`if (k < n) x = p[k]` compiles to a branch, a load and a wait of its own); request everything first with clamped indices
instead.  (Found vseg_upsweep this way: 32 serial round trips per thread, 110 -> 99 us at C4.)     usage: tools/isa_load_waits.py [min_loads]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sigman_release_amd", "csrc")
EXTRA = {"binning": ["-fno-honor-nans"], "render": ["-fno-slp-vectorize"]}
min_loads = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for f in ("preprocess", "binning", "render", "knn", "loss"):
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-ffp-contract=off",
                        *EXTRA.get(f, []), "--cuda-device-only", "-S", os.path.join(ROOT, f + ".hip"), "-o", tmp.name],
                       check=True, stderr=subprocess.DEVNULL)
        txt = open(tmp.name).read()
    for m in re.finditer(r"^(_Z[^\n:]*):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        loads = len(re.findall(r"global_load_", body))
        waits = len(re.findall(r"s_waitcnt vmcnt", body))
        w0 = len(re.findall(r"s_waitcnt vmcnt\(0\)", body))
        if loads >= min_loads:
            flag = "  <-- serial?" if w0 >= 0.6 * loads else ""
            print(f"{f:10s} {re.sub(r'^_ZN12_GLOBAL__N_1[0-9]+', '', name)[:56]:56s} loads {loads:3d}  waits {waits:3d}  of them vmcnt(0) {w0:3d}{flag}")
