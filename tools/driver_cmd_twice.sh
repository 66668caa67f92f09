for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-variants --no-cpu-baseline > gpurun_out/drv_$i.json 2>/dev/null; python - <<PY
import json
for l in open("gpurun_out/drv_$i.json"):
    if l.startswith("{\"metric\""):
        d=json.loads(l); print(d["ms_per_step"], d["windows"]["wall_ms_per_step_min_median_max"], d["host_queue"]["queue_drain_steps"], d["sclk_mhz_probe"]["before_timed_region"])
PY
done; cat /proc/loadavg
