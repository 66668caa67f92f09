# kernel time of the k-NN front end (knn3_kernel and friends) inside bench runs          usage: gpu_knn_time.sh [configs]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for c in ${@:-c2 c5}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/knn_$c -o x -- python $R/bench.py --config $c --no-variants --no-cpu-baseline --steps 3 --warmup 2 > /dev/null 2>&1
  python3 - $c <<PY
import csv, glob, sys
f = glob.glob("/tmp/knn_%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)[0]
rows = {r["Name"][:44]: (float(r["AverageNs"]) / 1e3, r["Calls"]) for r in csv.DictReader(open(f))}
print(sys.argv[1], {k: (round(v[0], 1), v[1]) for k, v in rows.items() if any(s in k for s in ("knn", "cell_", "bbox", "grid_setup", "cov3d"))})
PY
done
