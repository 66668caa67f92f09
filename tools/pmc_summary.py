"""Dev tool: turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) + a --kernel-trace --stats pass of `bench.py --config <c>` into
profiles/r0N_pmc_<c>.json (what bench.py reads `roofline.traffic` from) and print a per-kernel table (markdown) with the algorithmic
bytes of SURVEY.md 8(d) next to the counters.
usage: python tools/pmc_summary.py <config> <dir_fetch> <dir_write> <kernel_stats.csv> <bench.json> <out.json>
Correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads, so read bytes
~= 2 x FETCH_SIZE (KB); WRITE_SIZE is taken as is (KB).  Calibration on this code: clamped_l1_kernel reads color + target and writes the
gradient image, 3 x 4 B per pixel and channel each way -- see the `calibration` entry of the output."""
import csv, glob, json, re, statistics, subprocess, sys

GROUPS = [  # (bench kernel id name, regex over rocprof kernel names)
    # (F1 + F2 are ONE row since round 5: SURVEY 8d's 8 B per Gaussian of the scan are the tile counts, which never leave preprocess_fwd's workgroups --
    # scan_block_sums only scans one sum per workgroup, a row of its own priced it at more than the memory peak)
    ("preprocess_fwd+scan", r"(preprocess_fwd_kernel|scan_block_sums_kernel)"), ("duplicate_keys", r"duplicate_keys_kernel"),
    ("radix_sort(all passes)", r"(wide_|radix_|vseg_|tile_sort|deep_tile|tile_collect)"), ("tile_ranges", r"tile_ranges_kernel"),
    ("render_fwd", r"(render_fwd|fwd_prepare)"), ("render_bwd", r"render_bwd"), ("preprocess_bwd", r"preprocess_bwd(_lanes)?_kernel"),
    ("clamped_l1", r"clamped_l1_kernel"),
]


def short(name):
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def load_pmc(d, counter):
    per = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    return per


def main():
    cfg, dfetch, dwrite, stats_csv, bench_json, out_path = sys.argv[1:7]
    ddram = sys.argv[7] if len(sys.argv) > 7 else None          # optional third pass: TCC_EA0_RDREQ_DRAM_sum / _32B_sum / TCC_EA0_WRREQ_DRAM_sum / TCC_EA0_WRREQ_64B_sum
    fetch, write = load_pmc(dfetch, "FETCH_SIZE"), load_pmc(dwrite, "WRITE_SIZE")
    dram = {c: load_pmc(ddram, c) for c in ("TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_RDREQ_DRAM_32B_sum", "TCC_EA0_WRREQ_DRAM_sum", "TCC_EA0_WRREQ_64B_sum",
                                            "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum")} if ddram else {}
    bench = json.loads([l for l in open(bench_json) if l.startswith("{")][-1])
    stats = {short(r["Name"]): (float(r["AverageNs"]) / 1e3, int(r["Calls"])) for r in csv.DictReader(open(stats_csv))}
    c = bench["config"]
    P, HW = int(re.search(r"(\d+) Gaussians/subject", c["workload"]).group(1)), None
    size = int(re.search(r"(\d+)x\d+,", c["workload"]).group(1))
    slots, Rn = c["view_slots_this_gpu"], c["num_rendered_per_gpu"]
    bwd = "fwd+bwd" in c["workload"]
    HW, tiles = size * size * slots, ((size + 15) // 16) ** 2 * slots
    nq = P * slots
    # B1's 88 B per tile instance are per instance the backward can VISIT (list entries up to the tile's last contributor: bench.py); the whole
    # list only where the line does not carry the count (lines of rounds 1-5)
    Rv = c.get("tile_instances_within_reach_of_the_backward_per_gpu", Rn)
    alg = {"preprocess_fwd+scan": 84 * nq, "duplicate_keys": 20 * nq + 12 * Rn, "radix_sort(all passes)": 24 * Rn,
           "tile_ranges": 8 * Rn + 8 * tiles, "render_fwd": 44 * Rn + 24 * HW, "render_bwd": 88 * Rv + 28 * HW, "preprocess_bwd": 108 * nq,
           "clamped_l1": (40 if "masked" in c["workload"] else 36) * HW}
    if c.get("fused_step"):
        # the fused single-view step: no loss launch -- the compositing kernel also reads the target (+ mask) and writes dL/dcolor (bench.py)
        alg["render_fwd"] += (28 if "masked" in c["workload"] else 24) * HW
    # launches per step: total calls / steps is unreliable under warm-up; use median KB per launch x launches of one step (= kernels that share a group)
    out = {"config": cfg, "P": P, "size": size, "view_slots": slots, "num_rendered": Rn,
           "command": f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --config {cfg} "
                      "--no-cpu-baseline --no-variants --steps 5 --warmup 3 (two separate passes); durations from a third pass with --kernel-trace --stats",
           "units": "bytes per launch group and step: sum over the group's kernels of (median counter KB per launch x 1024); reads corrected 2 x FETCH_SIZE "
                    "(MI355X_MICROARCH.md: FETCH_SIZE tallies 128-B requests at 64 B on gfx950); WRITE_SIZE as is",
           "kernels": {}, "per_kernel": {}}
    try:
        out["commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        pass
    calls = {k: v[1] for k, v in stats.items()}
    # forwards in the profiled run (timed steps + probe / gt renders): launches of the emission kernel (all instantiations)
    n_fwd = max(sum(v for k, v in calls.items() if k.startswith("duplicate_keys_kernel")), 1)
    for k in sorted(set(fetch) | set(write)):
        f, w = statistics.median(fetch.get(k, [0.0])), statistics.median(write.get(k, [0.0]))
        us, n = stats.get(k, (0.0, 0))
        fwd_chain = bool(re.search(r"(preprocess_fwd|scan_block|duplicate|wide_|radix_|vseg_|tile_sort|tile_collect|deep_tile|tile_ranges|fwd_prepare)", k))
        per_step = max(1, round(n / n_fwd)) if fwd_chain else 1                # launches of this kernel per forward (radix passes: several)
        if k.startswith("duplicate_keys_kernel") or k.startswith("scan_block"):
            per_step = 1 if n >= n_fwd / 2 else 0                              # (an instantiation only the untimed exact-mode probes use: not part of a step)
        out["per_kernel"][k] = {"fetch_size_kb": round(f, 1), "write_size_kb": round(w, 1), "hbm_bytes_corrected": int((2 * f + w) * 1024),
                                "avg_us": round(us, 2), "launches_per_step": per_step}
        if dram:
            # requests that leave the L2 for the memory side ("destined for DRAM": Infinity Cache hits included -- rocprofv3 exposes no MALL
            # hit counter on this stack).  TCC_EA0_RDREQ_DRAM_32B counts reads in 32-byte units (a 128-byte request counts 4); writes are 32-byte
            # requests unless counted in TCC_EA0_WRREQ_64B.  Calibration: clamped_l1 at C3 reads 28 B and writes 12 B per pixel: 469.8 / 201.8 MB
            # by these counters, 469.8 / 201.3 MB by arithmetic -- and 2 x FETCH_SIZE + WRITE_SIZE gives the same bytes: everything the L2 misses
            # is "DRAM traffic" to these counters, whether the Infinity Cache serves it or HBM does.
            m = lambda c: statistics.median(dram.get(c, {}).get(k, [0.0]))
            rd, rd32, wr, wr64 = m("TCC_EA0_RDREQ_DRAM_sum"), m("TCC_EA0_RDREQ_DRAM_32B_sum"), m("TCC_EA0_WRREQ_DRAM_sum"), m("TCC_EA0_WRREQ_64B_sum")
            out["per_kernel"][k].update({"ea_rdreq_dram": int(rd), "ea_rdreq_dram_32b": int(rd32), "ea_wrreq_dram": int(wr), "ea_wrreq_64b": int(wr64),
                                         "ea_dram_bytes": int(rd32 * 32 + min(wr, wr64) * 64 + max(wr - wr64, 0.0) * 32)})
    print(f"| kernel(s) | us / step | algorithmic MB | achieved GB/s | frac of 8 TB/s | counter MB | traffic / algorithmic |\n|---|---|---|---|---|---|---|")
    for g, rx in GROUPS:
        ks = [k for k in out["per_kernel"] if re.search(rx, k)]
        if g == "render_fwd" and bwd:      # the untimed gt / counter renders use the kernel without auxiliary outputs: keep the training variant only
            ks = [k for k in ks if "<0>" not in k] or ks
        if not ks:
            continue
        tot = sum(out["per_kernel"][k]["hbm_bytes_corrected"] * out["per_kernel"][k]["launches_per_step"] for k in ks)
        us = sum(out["per_kernel"][k]["avg_us"] * out["per_kernel"][k]["launches_per_step"] for k in ks)
        if us <= 0:
            continue
        out["kernels"][g] = {"members": ks, "hbm_bytes_corrected": tot, "us": round(us, 2), "algorithmic_bytes": alg.get(g)}
        if dram:
            out["kernels"][g]["ea_dram_bytes"] = sum(out["per_kernel"][k].get("ea_dram_bytes", 0) * out["per_kernel"][k]["launches_per_step"] for k in ks)
        a = alg.get(g, 0)
        if us > 0 and a:
            print(f"| {g} | {us:.1f} | {a / 1e6:.2f} | {a / us / 1e3:.0f} | {a / us / 1e3 / 8000:.3f} | {tot / 1e6:.2f} | {tot / a:.1f}x |")
    if "clamped_l1_kernel" in out["per_kernel"]:
        k = out["per_kernel"]["clamped_l1_kernel"]
        out["calibration"] = {"kernel": "clamped_l1_kernel", "known_read_bytes": (28 if "masked" in c["workload"] else 24) * HW, "known_write_bytes": 12 * HW,
                              "fetch_size_x2_bytes": int(2 * k["fetch_size_kb"] * 1024), "write_size_bytes": int(k["write_size_kb"] * 1024)}
    json.dump(out, open(out_path, "w"), indent=1)


main()
