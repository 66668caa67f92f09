"""Regenerates the per-config tables and the SQ-counter table of the "Round 3" section of profiles/README.md from profiles/r03_<cfg>_table.md,
profiles/r03_sq_<cfg>.json and the bench lines profiles/<prefix>_bench_<cfg>.json.        usage: python tools/update_profiles_readme_r3.py r03c"""
import json, re, sys
pfx = sys.argv[1]
p = "profiles/README.md"
s = open(p).read()
names = {"radix_sort(all passes)": {"c2": "sort (order-free wide tile pass + per-tile LDS distribution sort)", "c3": "sort (view-segmented: tile pass + per-tile register sort)",
                                    "c4": "sort (view-segmented: tile pass + per-tile register sort)",
                                    "c5": "sort (view-segmented tile pass + per-tile LDS distribution sort; round 2: six whole-key radix passes, 223 µs, 10.2x)"},
         "render_fwd": {"c2": "render_fwd (segment-parallel)", "c3": "render_fwd (one wave per quadrant, row checkpoints without depth/alpha)",
                        "c4": "render_fwd (one wave per quadrant, no checkpoints: forward only)", "c5": "render_fwd (segment-parallel, all checkpoints)"},
         "render_bwd": {"c2": "render_bwd (bucket-parallel, SPLIT)", "c3": "render_bwd (bucket-parallel)", "c5": "render_bwd (bucket-parallel, SPLIT, depth/alpha gradients)"}}
def table(c):
    out = []
    for line in open(f"profiles/r03_{c}_table.md").read().strip().splitlines():
        cells = [x.strip() for x in line.strip("|").split("|")]
        if cells[0] in names and c in names[cells[0]]:
            line = line.replace("| " + cells[0] + " |", "| " + names[cells[0]][c] + " |", 1)
        out.append(line)
    return "\n".join(out)
b = {c: json.loads([l for l in open(f"profiles/{pfx}_bench_{c}.json") if l.startswith("{")][-1]) for c in ("c2", "c3", "c4", "c5")}
rows = []
for c, kern in (("c4", "render_fwd_wave_kernel"), ("c3", "render_fwd_wave_kernel"), ("c3", "render_bwd_bucket_kernel"), ("c2", "render_fwd_seg_kernel"), ("c2", "render_bwd_bucket_kernel")):
    v = json.load(open(f"profiles/r03_sq_{c}.json"))["kernels"][kern]
    rate = v["SQ_INSTS_VALU"] / v["avg_us"] * 1e6
    rows.append(f"| {c.upper()} `{kern}` | {v['avg_us']:.0f} | {v['SQ_INSTS_VALU']:.3g} | {v['SQ_INSTS_SALU']:.3g} | {rate:.3g} | {100 * rate / 8.55e11:.0f} % | {v['SQ_INSTS_LDS']:.3g} | "
                f"{v['SQ_LDS_BANK_CONFLICT']:.3g} / {v['SQ_LDS_IDX_ACTIVE']:.3g} | {v['SQ_WAVES']:.0f} |")
i = s.index("| kernel | µs | VALU instructions |")
j = s.index("**C2** (100 000 Gaussians, 1 view 512², R = 2.0e5; step", i)
sq = ("| kernel | µs | VALU instructions | scalar instructions | VALU per second | of the issue rate | LDS instructions | LDS bank conflicts / index cycles | waves |\n|---|---|---|---|---|---|---|---|---|\n"
      + "\n".join(rows) + "\n\n(First half of round 3, before the wave forward's per-batch checkpoint masks and the backward's reads-ahead: C3 forward 1098 µs, 5.01e8 VALU and "
      "3.60e8 scalar instructions, 53 %; C3 backward 888 µs with 5.2e7 LDS instructions and 4.2e7 conflict cycles; C2 backward 36 µs, 37 %.)\n\n")
new = sq + f'''**C2** (100 000 Gaussians, 1 view 512², R = 2.0e5; step {b["c2"]["ms_per_step"]:.3f} ms):

{table("c2")}

**C3** (8 subjects × 8 views 512² in one launch chain, 100 000 Gaussians each, R = 1.33e7; step {b["c3"]["ms_per_step"]:.2f} ms):

{table("c3")}

**C4** (200 000 Gaussians, 90 views 1024², forward only, R = 4.4e7; step {b["c4"]["ms_per_step"]:.2f} ms):

{table("c4")}

**C5** (1M Gaussians, 1 view 512², depth + alpha gradients, R = 2.2e6; step {b["c5"]["ms_per_step"]:.3f} ms; round 2: 0.493 ms):

{table("c5")}
'''
k = s.index("**C5** (1M Gaussians, 1 view 512², depth + alpha gradients", j)
# end of the C5 table = first blank line after its last row
m = re.search(r"\n\n", s[k + 10:].split("|---|", 1)[1])
end = k + 10 + len(s[k + 10:].split("|---|", 1)[0]) + len("|---|") + (m.start() if m else len(s))
s = s[:i] + new + s[end + 1:]
open(p, "w").write(s)
