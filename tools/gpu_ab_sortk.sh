# A/B of two library builds (tools/ab/old.so, new.so): per-kernel times of the sort kernels from rocprofv3 kernel stats     usage: gpu_ab_sortk.sh [config]
export TMPDIR=/tmp; c=${1:-c4}; R=$GRAFT_REPO_ROOT; cd /tmp
for rep in 1 2; do for v in old new; do
  SIGMAN_PY_NODE=1 SIGMAN_GSPLAT_LIB=$R/tools/ab/$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -o x -- python $R/bench.py --config $c --no-variants --no-cpu-baseline --steps 5 --warmup 3 > /dev/null 2>&1
  python3 - $v <<PY
import csv, glob, sys
f = glob.glob("/tmp/ab_%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)[0]
rows = {r["Name"][:44]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f))}
print(sys.argv[1], {k: round(v, 1) for k, v in rows.items() if "vseg" in k or "tile_sort" in k or "duplicate" in k})
PY
done; done
