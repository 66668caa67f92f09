O=gpurun_out/c5big; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bin.py -x -q 2>&1 | tail -5 > $O/t_bin.txt; cat $O/t_bin.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "c5 or deep or full_size" --deselect tests/test_gpu_parity.py::test_full_size_record_matches_kernel_sources 2>&1 | tail -5 > $O/t_par.txt; cat $O/t_par.txt
for m in 1 2 1 2; do SIGMAN_SORT_COLLECT=$m timeout 300 python bench.py --config c5 --no-cpu-baseline --no-variants > $O/b_$m.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/b_$m.json')); print('collect', $m, d['ms_per_step'], d['windows']['wall_ms_per_step_min_median_max'], d['kernel_ms_per_step'])"; done | tee $O/ab.txt
SIGMAN_SORT_COLLECT=3 FUZZ_VIEWS=1,2 timeout 200 python tools/fuzz_bin_modes.py 60 2>&1 | tail -2 | tee $O/fuzz3.txt
timeout 200 python tools/fuzz_bin_modes.py 40 2>&1 | tail -2 | tee $O/fuzz.txt
bash tools/gpu_kstats.sh c5 | tee $O/ks.txt
