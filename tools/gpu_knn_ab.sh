# knn3 kernel time for several library builds under tools/ab/ (dev)      usage: LIBS="a b" gpu_knn_ab.sh [configs]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for v in $LIBS; do for c in ${@:-c2 c5}; do
  SIGMAN_PY_NODE=1 SIGMAN_GSPLAT_LIB=$R/tools/ab/$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/knn_${v}_$c -o x -- python $R/bench.py --config $c --no-variants --no-cpu-baseline --steps 3 --warmup 2 > /dev/null 2>&1
  python3 - $v $c <<PY
import csv, glob, sys
f = glob.glob("/tmp/knn_%s_%s/**/*kernel_stats.csv" % (sys.argv[1], sys.argv[2]), recursive=True)[0]
rows = {r["Name"][:40]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f))}
print(sys.argv[1], sys.argv[2], {k.split("::")[-1][:14]: round(v, 1) for k, v in rows.items() if any(s in k for s in ("knn3", "cell_count", "cell_scatter", "cell_scan_k"))})
PY
done; done
