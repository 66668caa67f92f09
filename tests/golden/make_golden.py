"""Generates tests/golden/*.npz from the CPU oracle (oracle/gsplat_ref.c) AFTER it agreed with the dense autograd oracle.

Run:  python tests/golden/make_golden.py        (deterministic; committed outputs are small .npz files)
Each fixture holds the seeded inputs, every integer artefact (radii, rect, sorted keys, point_list, ranges, n_contrib),
the images, and all gradients for fixed upstream grads.  The GPU tests compare the HIP path against these files,
so a silent change of the oracle itself is also caught (tests/test_oracle_cpu.py re-derives them).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cases  # noqa: E402
from oracle import ref  # noqa: E402

GOLDEN_CASES = ("cloud_precomp", "cloud_precomp_ragged", "cloud_sh3", "cull_and_clamp", "opaque_stack", "single_gaussian",
                "c1_10k_256")       # the last one is BASELINE.json's config 1 at full size (10 000 random Gaussians, one 256 x 256 view)
LEAN_CASES = ("c1_10k_256",)


def make(name):
    inp, st = cases.CASES[name]()
    sv = cases.single_view(st)
    r = ref.forward(**inp, **sv)
    gC, gD, gA = cases.grads_for(st["image_height"], st["image_width"])
    g = ref.backward(r, gC, gD, gA)
    # (the full-size case stores no inputs and no upstream gradients: both are seeded -- cases.CASES[name]() and cases.grads_for --
    # and 1.8 MB of incompressible random floats would be most of the file)
    small = name not in LEAN_CASES
    out = {f"in_{k}": v for k, v in inp.items()} if small else {}
    if small: out.update(grad_color=gC, grad_depth=gD, grad_alpha=gA)
    out.update(radii=r.radii, rect=r.rect, tiles_touched=r.tiles_touched, keys=r.keys, point_list=r.point_list,
               ranges=r.ranges, n_contrib=r.n_contrib, color=r.color, depth=r.depth, alpha=r.alpha, final_T=r.final_T)
    out.update({f"g_{k}": v for k, v in g.items()})
    return out


if __name__ == "__main__":
    for name in GOLDEN_CASES:
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **make(name))
        print(name, os.path.getsize(path) // 1024, "KiB")
