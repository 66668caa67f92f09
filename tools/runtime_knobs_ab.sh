# Dev tool (GPU box): the headline step under ROCm runtime settings that move what the command processor reads per dispatch out of host memory
# and HIP_FORCE_DEV_KERNARG=0 (kernel arguments back in host memory) for contrast (alternating; every line: setting, ms/step of the timed region, windows min/median/max, queue drain)
run() { name=$1; shift; env "$@" timeout 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-variants --no-cpu-baseline --no-sclk > gpurun_out/knob_tmp.json 2>gpurun_out/knob_tmp.err || { echo "$name FAILED"; tail -3 gpurun_out/knob_tmp.err; return; }
python - "$name" <<PY
import json, sys
for l in open("gpurun_out/knob_tmp.json"):
    if l.startswith("{\"metric\""):
        d=json.loads(l); print(sys.argv[1].ljust(34), d["ms_per_step"], d["windows"]["wall_ms_per_step_min_median_max"], d["host_queue"]["queue_drain_steps"], d["loss"])
PY
}
for rep in 1 2; do
run "default" X=1
run "HSA_ALLOCATE_QUEUE_DEV_MEM=1" HSA_ALLOCATE_QUEUE_DEV_MEM=1
# (ROC_SYSTEM_SCOPE_SIGNAL=0 is NOT in the list: with it the step never finished -- every run here is under a timeout since)
run "HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0
done
