# full -m gpu suite with the full-size excursion record refreshed + bench lines      usage (GPU box): bash tools/gpu_suite.sh <tag> [configs...]
T=${1:-suite}; shift; O=gpurun_out/$T; mkdir -p $O
SIGMAN_RECORD_OBSERVED=1 timeout 2400 python -m pytest tests -m gpu -q -x --timeout=1200 --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -12 $O/pytest.log
for c in ${@:-c2}; do timeout 900 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; tail -2 $O/bench_$c.err | grep -v amdgpu.ids; python - <<PY
import json
try:
    d = json.load(open("$O/bench_$c.json"))
    print("$c", d["ms_per_step"], d["value"], d["kernel_ms_per_step"], d.get("variants"), d["config"].get("host_threads"), d.get("frontend_ms_per_subject"))
except Exception as e:
    print("$c: no line", e)
PY
done
