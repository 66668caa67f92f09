"""Regenerates the round-2 tables at the end of profiles/README.md from profiles/r02_<cfg>_table.md and the bench lines.
usage: python tools/update_profiles_readme.py <bench prefix, e.g. r02j>"""
import json, sys
pfx = sys.argv[1]
p = "profiles/README.md"
s = open(p).read()
i = s.index("**C2** (100 000 Gaussians, 1 view 512²")
names = {"radix_sort(all passes)": {"c2": "sort (order-free wide tile pass + per-tile register sort)", "c3": "sort (view-segmented: tile pass + per-tile register sort)",
                                    "c4": "sort (view-segmented: tile pass + per-tile register sort)", "c5": "sort (three-kernel LSD over the whole key, 6 passes: deep tile lists)"},
         "render_fwd": {"c2": "render_fwd (segment-parallel)", "c3": "render_fwd (one wave per quadrant, row checkpoints)", "c4": "render_fwd (one wave per quadrant, no checkpoints: forward only)",
                        "c5": "render_fwd (segment-parallel)"},
         "render_bwd": {"c2": "render_bwd (bucket-parallel, SPLIT)", "c3": "render_bwd (bucket-parallel)", "c5": "render_bwd (bucket-parallel, SPLIT, depth/alpha gradients)"}}
def table(c):
    out = []
    for line in open(f"profiles/r02_{c}_table.md").read().strip().splitlines():
        cells = [x.strip() for x in line.strip("|").split("|")]
        if cells[0] in names and c in names[cells[0]]:
            line = line.replace("| " + cells[0] + " |", "| " + names[cells[0]][c] + " |", 1)
        out.append(line)
    return "\n".join(out)
b = {c: json.load(open(f"profiles/{pfx}_bench_{c}.json")) for c in ("c2", "c3", "c4", "c5")}
hb = lambda c: (b[c]["step_hbm"]["achieved_GBps"] / 1000, 100 * b[c]["step_hbm"]["frac_of_peak"])
tail = s[s.index("Per-tile register sort alone"):] if "Per-tile register sort alone" in s else ""
new = f'''**C2** (100 000 Gaussians, 1 view 512², R = 2.0e5; step {b["c2"]["ms_per_step"]:.3f} ms):

{table("c2")}

**C3** (8 subjects × 8 views 512² in one launch chain, 100 000 Gaussians each, R = 1.33e7; step {b["c3"]["ms_per_step"]:.2f} ms):

{table("c3")}

(round 1, same shape: sort 1 050 µs at 10.5× — six whole-key passes; emission 3.2×; backward 1.32 ms; gather 0.52 ms.)

**C4** (90-view orbit at 1024², 200 000 Gaussians, forward only, R = 4.4e7; step {b["c4"]["ms_per_step"]:.2f} ms — round 1: 7.9 ms):

{table("c4")}

(round 1: seven onesweep passes, 3.56 ms.  The scatter of the tile pass still costs 2.5× its bytes in HBM write requests at 4096
tiles per view: 8-byte stores to ≈ 570 open tile segments per view, `r02_pmc_c4.json`.)

**C5** (1M Gaussians, 1 view 512², R = 2.2e6, depth + alpha gradients on; step {b["c5"]["ms_per_step"]:.3f} ms):

{table("c5")}

(`render_bwd` below 1×: C5's 1M Gaussians are 10 jittered layers, most tile instances lie behind the point where their pixels
saturate and are never gathered by the backward — the algorithmic figure counts every instance.  The view-segmented sort does not
apply: 134 of the 437 occupied tiles hold more than 8192 entries, 23 more than the register sort's 16 384.)
Whole step: C5 616 MB algorithmic in {b["c5"]["ms_per_step"]:.3f} ms = {hb("c5")[0]:.2f} TB/s = {hb("c5")[1]:.0f} % of the HBM roofline;
C3: 4.57 GB in {b["c3"]["ms_per_step"]:.2f} ms = {hb("c3")[0]:.2f} TB/s = {hb("c3")[1]:.0f} %; C4: 8.03 GB in {b["c4"]["ms_per_step"]:.2f} ms = {hb("c4")[0]:.2f} TB/s = {hb("c4")[1]:.0f} %.

'''
open(p, "w").write(s[:i] + new + tail)
