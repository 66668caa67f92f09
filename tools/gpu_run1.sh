set -x
export SIGMAN_RECORD_OBSERVED=1
mkdir -p gpurun_out/r2a
timeout 2400 python -m pytest tests -m gpu -q -x --timeout=1200 > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
tail -15 gpurun_out/r2a/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2a/smoke.log 2>&1; tail -2 gpurun_out/r2a/smoke.log
timeout 900 python bench.py > gpurun_out/r2a/bench_c2.json 2> gpurun_out/r2a/bench_c2.err; tail -c 3000 gpurun_out/r2a/bench_c2.json; tail -5 gpurun_out/r2a/bench_c2.err
for c in c3 c4 c5; do timeout 900 python bench.py --config $c > gpurun_out/r2a/bench_$c.json 2> gpurun_out/r2a/bench_$c.err; tail -c 2500 gpurun_out/r2a/bench_$c.json; tail -5 gpurun_out/r2a/bench_$c.err; done
SIGMAN_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 50 --warmup 10 > gpurun_out/r2a/bench_c2_n2gloo.json 2> gpurun_out/r2a/bench_c2_n2gloo.err; tail -c 1500 gpurun_out/r2a/bench_c2_n2gloo.json; tail -5 gpurun_out/r2a/bench_c2_n2gloo.err
SIGMAN_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --config c3 --steps 5 --warmup 2 > gpurun_out/r2a/bench_c3_n2gloo.json 2> gpurun_out/r2a/bench_c3_n2gloo.err; tail -c 1500 gpurun_out/r2a/bench_c3_n2gloo.json; tail -5 gpurun_out/r2a/bench_c3_n2gloo.err
timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r2a/bench_refuse.out 2>&1; echo "refuse rc=$?"; tail -3 gpurun_out/r2a/bench_refuse.out
