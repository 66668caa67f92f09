# A/B of the forward kernels on one box: per-tile workgroups (1) vs one wave per quadrant (3)     usage: gpu_ab_fwd.sh [configs]
O=gpurun_out/ab; mkdir -p $O
for rep in 1 2; do for m in 1 3; do for c in ${@:-c3 c4}; do SIGMAN_FWD_MODE=$m timeout 600 python bench.py --config $c --no-variants --no-cpu-baseline > $O/b_${c}_$m.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/b_${c}_$m.json')); print('mode $m', '$c', d['ms_per_step'], d['kernel_ms_per_step']['render_fwd'], d['kernel_ms_per_step'].get('render_bwd'))"; done; done; done
