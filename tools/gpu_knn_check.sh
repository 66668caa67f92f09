# 3-NN front end: exactness tests (cKDTree), end-to-end timing, per-kernel times          usage (GPU box): bash tools/gpu_knn_check.sh <tag>
T=${1:-knn}; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_callers.py -m gpu -q -x --timeout=120 -k "dist_cuda2 or renderer or covariance" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
timeout 120 python tools/time_knn.py 2>&1 | grep -v amdgpu.ids | tee $O/time_knn.txt
for w in humanoid volume; do
cd /tmp && rm -rf /tmp/knnprof && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/knnprof -o x -- python $GRAFT_REPO_ROOT/tools/time_knn.py $w > /dev/null 2>&1
python3 - $w <<PY | tee -a $O/kernels.txt
import csv, glob, sys
f = glob.glob("/tmp/knnprof/**/*kernel_stats.csv", recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        if any(s in r["Name"] for s in ("knn", "brick", "cov3d")): print(sys.argv[1], r["Name"][:60], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2), "us")
PY
done
