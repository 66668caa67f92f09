# per-kernel average times (rocprofv3 kernel stats) of one bench config        usage: gpu_kstats.sh [config] [name substrings...]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; c=${1:-c2}; shift; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$c -o x -- python $R/bench.py --config $c --no-variants --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
python3 - "$@" <<PY
import csv, glob, sys
f = glob.glob("/tmp/ks_$c/**/*kernel_stats.csv", recursive=True)[0]
pats = sys.argv[1:] or ["preprocess", "duplicate", "collect", "deep_tile", "tile_sort", "render", "clamped", "scan", "vseg", "radix", "fill", "copy"]
for r in sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"])):
    if any(p in r["Name"] for p in pats): print("%-70s calls %5s avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
