O=gpurun_out/c5big; mkdir -p $O
SIGMAN_GSPLAT_LIB=$PWD/tools/ab/stamps.so SIGMAN_PY_NODE=1 timeout 300 python tools/deep_stamps.py c5 2>&1 | grep -v amdgpu.ids | tee $O/stamps_c5.txt
timeout 600 python -m pytest tests/test_gpu_bin.py -x -q 2>&1 | tail -3
for m in 1 2 1 2; do SIGMAN_SORT_COLLECT=$m timeout 300 python bench.py --config c5 --no-cpu-baseline --no-variants > $O/b_$m.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/b_$m.json')); print('collect', $m, d['ms_per_step'], d['windows']['wall_ms_per_step_min_median_max'], d['kernel_ms_per_step'])"; done | tee $O/ab2.txt
SIGMAN_SORT_COLLECT=3 FUZZ_VIEWS=1,2 timeout 200 python tools/fuzz_bin_modes.py 40 2>&1 | tail -1 | tee $O/fuzz3.txt
