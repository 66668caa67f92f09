#!/bin/bash
# Dev tool (GPU box): the headline step with the backward waiting for its own forward's count ("own", the default) against the lazy look
# (SIGMAN_COUNT_WAIT=lazy), on a quiet host and with one busy-loop process per host core next to it (what other tenants of a shared host do to
# the thread that feeds the GPU).   usage: bash tools/host_load_ab.sh   -> one line per run: ms/step, windows, queue drain, issue ms, drain, clock
N=$(nproc)
for rep in 1 2; do
for load in 0 1; do
  pids=""
  if [ $load = 1 ]; then for i in $(seq $N); do timeout 70 python -c "while True: pass" & pids="$pids $!"; done; sleep 1; fi
  for mode in own lazy lazy:4; do
    SIGMAN_COUNT_WAIT=$mode python bench.py --gpus 1 --steps 20 --warmup 5 --no-variants --no-cpu-baseline --no-sclk > gpurun_out/ab_${mode}_${load}_$rep.log 2>&1
    echo -n "load=$load mode=$mode: "; python - gpurun_out/ab_${mode}_${load}_$rep.log <<PY
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d = json.loads(l); w = d["windows"]; print(d["ms_per_step"], w["wall_ms_per_step_min_median_max"], w["queue_drain_steps_min_median_max"], d["host_queue"]["issue_ms_per_step"], d["host_queue"]["queue_drain_steps"], d["sclk_mhz_probe"]["before_timed_region"])
PY
  done
  for p in $pids; do kill $p 2>/dev/null; done; wait 2>/dev/null
done
done
cat /proc/loadavg
