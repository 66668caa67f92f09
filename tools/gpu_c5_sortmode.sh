# C5 with the view-segmented sort forced (SIGMAN_SORT_MODE=4) vs automatic: step time + per-kernel stats of the sort kernels
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for m in "" 4; do
  SIGMAN_SORT_MODE=$m python $R/bench.py --config c5 --no-variants --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('mode', '$m' or 'auto', d['ms_per_step'], d['kernel_ms_per_step']['radix_sort(all passes)'])"
  SIGMAN_SORT_MODE=$m rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5_$m -o x -- python $R/bench.py --config c5 --no-variants --no-cpu-baseline --steps 5 --warmup 3 > /dev/null 2>&1
  python3 - "$m" <<PY
import csv, glob, sys
f = glob.glob("/tmp/c5_%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)[0]
rows = {r["Name"][:50]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f))}
print({k: round(v, 1) for k, v in rows.items() if any(s in k for s in ("vseg", "tile_sort", "radix", "tile_ranges"))})
PY
done
