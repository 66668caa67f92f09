# final evidence of a round: full gpu test suite, smoke, all bench configs (+ 2-rank gloo plumbing runs), rocprofv3 stats + PMC for c2/c3/c5   usage: gpu_final.sh <tag>
set -x
T=$1; O=gpurun_out/$T; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --timeout=1200 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for c in c2 c3 c4 c5; do bash tools/gpu_profile.sh $c; done
bash tools/gpu_bench_all.sh $T
