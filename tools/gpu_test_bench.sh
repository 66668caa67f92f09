# full gpu test suite + one bench line per config (+ C3 with the compact checkpoint layout)     usage (on the GPU box): bash tools/gpu_test_bench.sh
set -x
O=gpurun_out/r2c; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x --timeout=1200 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
for c in c2 c3 c4 c5; do timeout 600 python bench.py --config $c --no-variants --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; python -c "
import json; d=json.load(open('$O/bench_$c.json')); print('$c', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"; tail -3 $O/bench_$c.err; done
SIGMAN_AUX_ROWS_MAX_BYTES=1000000 timeout 600 python bench.py --config c3 --no-variants --no-cpu-baseline > $O/bench_c3_compact.json 2> $O/bench_c3_compact.err;  python -c "
import json; d=json.load(open('$O/bench_c3_compact.json')); print('c3-compact', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
