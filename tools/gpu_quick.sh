# quick check: sort parity tests + bench c3/c4 kernel times   usage: gpu_quick.sh [configs...]
O=gpurun_out/quick; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout=600 -k "sort or multiview or segmented or full_size" 2>&1 | tail -2
for c in ${@:-c3 c4}; do timeout 600 python bench.py --config $c --no-variants --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; python -c "
import json; d=json.load(open('$O/bench_$c.json')); print('$c', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"; tail -2 $O/bench_$c.err | grep -v amdgpu; done
