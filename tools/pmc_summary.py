"""Dev tool: turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py` into profiles/<name>.json.
usage: python tools/pmc_summary.py <dir_fetch> <dir_write> <out.json> "<workload note>"
Correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): on gfx950 FETCH_SIZE counts 64 B per 128-B request for 16 B/lane loads,
so read bytes ~= 2 x FETCH_SIZE (KB); WRITE_SIZE is taken as is (KB)."""
import csv, glob, json, statistics, sys

SHORT = {"render_bwd_bucket": "render_bwd", "render_bwd_kernel": "render_bwd", "render_fwd": "render_fwd", "tile_sort": "tile_sort",
         "wide_downsweep": "radix_downsweep", "wide_upsweep": "radix_upsweep", "wide_rowscan": "radix_rowscan",
         "radix_downsweep": "radix_downsweep", "radix_upsweep": "radix_upsweep", "radix_rowscan": "radix_rowscan",
         "preprocess_bwd": "preprocess_bwd", "preprocess_fwd": "preprocess_fwd", "duplicate_keys": "duplicate_keys",
         "clamped_l1": "clamped_l1", "tile_ranges": "tile_ranges", "scan_block_sums": "scan_block_sums", "fwd_prepare": "fwd_prepare"}


def load(d, counter):
    per = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = next((v for k, v in SHORT.items() if k in r["Kernel_Name"]), None)
            if name:
                per.setdefault(name, []).append(float(r["Counter_Value"]))
    return per


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"command": "SIGMAN_GRAPHS=0 rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py "
                  "--no-cpu-baseline --steps 5 --warmup 3 (two separate passes)",
       "units": "KB per launch (median over launches), raw counter values; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts 64 B per "
                "128-B request for 16 B/lane loads -> read bytes ~= 2 x FETCH_SIZE; WRITE_SIZE uncalibrated",
       "workload": sys.argv[4], "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f = statistics.median(fetch.get(k, [0.0])); w = statistics.median(write.get(k, [0.0]))
    out["kernels"][k] = {"launches_per_run": len(fetch.get(k, [])), "fetch_size_kb": round(f, 1), "write_size_kb": round(w, 1),
                         "hbm_bytes_corrected": int((2 * f + w) * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
