# fused single-view step: parity tests + A/B of the C2 / C1 step (SIGMAN_FUSED_STEP=0 / 1 / 2 on one box)
O=gpurun_out/fused; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_reference_calls.py -m gpu -q -x --timeout=300 -k "fused_single_view or cpp_batched_l1" 2>&1 | tail -15
for rep in 1 2; do for f in ${MODES:-0 1 2}; do for c in ${@:-c2}; do
  SIGMAN_FUSED_STEP=$f timeout 300 python bench.py --config $c --no-variants --no-cpu-baseline > $O/bench_${c}_f$f.json 2> $O/bench_${c}_f$f.err
  python -c "
import json; d=json.load(open('$O/bench_${c}_f$f.json')); print('$c fused=$f', d['value'], d['ms_per_step'], d.get('windows',{}).get('wall_ms_per_step_min_median_max'), d['kernel_ms_per_step'])"; tail -2 $O/bench_${c}_f$f.err | grep -v amdgpu
done; done; done
