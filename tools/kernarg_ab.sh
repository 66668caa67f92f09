# Dev tool (GPU box): the headline step with the kernel arguments in host memory (HIP_FORCE_DEV_KERNARG=0) against device memory (=1), alternating
for rep in 1 2 3; do for v in 0 1; do
HIP_FORCE_DEV_KERNARG=$v python bench.py --gpus 1 --steps 20 --warmup 5 --no-variants --no-cpu-baseline --no-sclk > gpurun_out/ka_${v}_$rep.json 2>/dev/null; python - <<PY
import json
for l in open("gpurun_out/ka_${v}_$rep.json"):
    if l.startswith("{\"metric\""):
        d=json.loads(l); print("HIP_FORCE_DEV_KERNARG=$v", d["ms_per_step"], d["windows"]["wall_ms_per_step_min_median_max"], d["host_queue"]["queue_drain_steps"])
PY
done; done
