# full -m gpu suite with the full-size excursion record refreshed + one c2 bench line      usage (GPU box): bash tools/gpu_suite.sh <tag>
T=${1:-suite}; O=gpurun_out/$T; mkdir -p $O
SIGMAN_RECORD_OBSERVED=1 timeout 2400 python -m pytest tests -m gpu -q -x --timeout=1200 --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -25 $O/pytest.log
timeout 600 python bench.py --config c2 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; cat $O/bench_c2.json
