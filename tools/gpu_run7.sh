set -x
O=gpurun_out/r2f; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x --timeout=1200 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
SIGMAN_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --config c4 --steps 5 --warmup 2 > $O/bench_c4_n2gloo.json 2> $O/bench_c4_n2gloo.err; tail -c 700 $O/bench_c4_n2gloo.json; grep -v "socket.cpp\|amdgpu.ids\|Gloo" $O/bench_c4_n2gloo.err | tail -5
