"""Rewrites the headline numbers of README.md / DESIGN.md from profiles/<prefix>_bench_<cfg>.json.   usage: python tools/update_headline_numbers.py r02l"""
import json, re, sys
pfx = sys.argv[1]
b = {c: json.load(open(f"profiles/{pfx}_bench_{c}.json")) for c in ("c2", "c3", "c4", "c5")}
fmt = lambda x: f"{x:,.0f}".replace(",", " ")
hb = lambda c: (b[c]["step_hbm"]["achieved_GBps"] / 1000, 100 * b[c]["step_hbm"]["frac_of_peak"])
s = open("README.md").read()
ops = f"{b['c2']['config']['gaussian_pixel_visits_per_gpu'] * b['c2']['value']:.2e}".replace("e+11", "e11")
s = re.sub(r"\*\*[\d ]+ views/s\*\* \(0\.\d+ ms/step\), \d\.\d+e11", f"**{fmt(b['c2']['value'])} views/s** ({b['c2']['ms_per_step']:.3f} ms/step), {ops}", s)
s = re.sub(r"\*\*[\d ]+ views/s\*\* \(\d\.\d+ ms per 64-view step; \d\.\d+ TB/s of algorithmic bytes = \d+ % of the HBM roofline\)",
           f"**{fmt(b['c3']['value'])} views/s** ({b['c3']['ms_per_step']:.2f} ms per 64-view step; {hb('c3')[0]:.2f} TB/s of algorithmic bytes = {hb('c3')[1]:.0f} % of the HBM roofline)", s)
s = re.sub(r"\*\*[\d ]+ views/s\*\* \(\d\.\d+ ms; 4\.4e7 tile instances; \d+ % of the HBM roofline\)",
           f"**{fmt(b['c4']['value'])} views/s** ({b['c4']['ms_per_step']:.2f} ms; 4.4e7 tile instances; {hb('c4')[1]:.0f} % of the HBM roofline)", s)
s = re.sub(r"\| [\d ]+ views/s \(0\.\d+ ms; \d\.\d+ TB/s algorithmic = \d+ % of the HBM roofline\) \| 1 700 \|",
           f"| {fmt(b['c5']['value'])} views/s ({b['c5']['ms_per_step']:.3f} ms; {hb('c5')[0]:.2f} TB/s algorithmic = {hb('c5')[1]:.0f} % of the HBM roofline) | 1 700 |", s)
open("README.md", "w").write(s)
s = open("DESIGN.md").read()
i = s.index("Round-2 numbers (one MI355X, `profiles/"); j = s.index("At one view the step is the sum of its ten")
s = s[:i] + f"""Round-2 numbers (one MI355X, `profiles/{pfx}_bench_*.json`; round 1 in brackets): C2 {fmt(b['c2']['value'])} views/s, {b['c2']['ms_per_step']:.3f} ms/step [5 026; 0.199];
C3 {fmt(b['c3']['value'])} views/s, {b['c3']['ms_per_step']:.2f} ms per 64-view step [11 842; 5.40]; C4 {fmt(b['c4']['value'])} views/s, {b['c4']['ms_per_step']:.2f} ms per 90 views at 1024² [11 400; 7.9]; C5 {fmt(b['c5']['value'])}
views/s, {b['c5']['ms_per_step']:.3f} ms [1 813]; CPU oracle {b['c2']['cpu_baseline']['value']:.1f} views/s on 256 host cores (C2).  """ + s[j:]
open("DESIGN.md", "w").write(s)
