# the fused single-view step against the unfused chain on ONE box: bench.py lines (driver-style: first 100-ms region + eight repeat windows), alternating
#   usage (GPU box): bash tools/gpu_fused_ab.sh [configs...]   -> gpurun_out/fused_ab.txt
O=gpurun_out/fused_ab.txt; : > $O
for rep in 1 2 3; do for c in ${@:-c2 c1}; do for f in 0 1; do
  SIGMAN_FUSED_STEP=$f timeout 300 python bench.py --config $c --no-variants --no-cpu-baseline 2>/dev/null | python -c "
import json, sys; d = json.loads(sys.stdin.read()); w = d['windows']['wall_ms_per_step_min_median_max']
print('$c SIGMAN_FUSED_STEP=$f rep $rep: first region %.4f ms/step (%.0f views/s), windows min/median/max %.4f / %.4f / %.4f, fused_step=%s' % (d['ms_per_step'], d['value'], w[0], w[1], w[2], d['config']['fused_step']))" | tee -a $O
done; done; done
