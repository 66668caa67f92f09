# C5 / C2 forward kernel choice: segment-parallel (2) vs one wave per quadrant (3)
for rep in 1 2; do for m in 2 3; do for c in c5; do SIGMAN_FWD_MODE=$m SIGMAN_PY_NODE=1 timeout 600 python bench.py --config $c --no-variants --no-cpu-baseline > gpurun_out/g_$c_$m.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/g_$c_$m.json')); print('mode $m', '$c', d['ms_per_step'], d['kernel_ms_per_step'])"; done; done; done
