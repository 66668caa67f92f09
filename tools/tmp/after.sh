O=gpurun_out/after; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bin.py -x -q 2>&1 | tail -3
for c in c5 c2 c1 c5 c2 c1; do timeout 300 python bench.py --config $c --no-cpu-baseline --no-variants > $O/b_$c.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/b_$c.json')); print('$c', d['ms_per_step'], d['windows']['wall_ms_per_step_min_median_max'], d['kernel_ms_per_step'])"; done | tee $O/b.txt
FUZZ_VIEWS=1,2 SIGMAN_SORT_DEEP=1 timeout 200 python tools/fuzz_bin_modes.py 40 2>&1 | tail -1 | tee $O/fuzzd.txt
timeout 200 python tools/fuzz_bin_modes.py 40 2>&1 | tail -1 | tee $O/fuzz.txt
