O=gpurun_out/gpw; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bin.py tests/test_gpu_parity.py -x -q -k "bin or gather or c1 or golden or small or capacity or sync_free" --deselect tests/test_gpu_parity.py::test_full_size_record_matches_kernel_sources 2>&1 | tail -4
timeout 200 python tools/fuzz_bin_modes.py 40 2>&1 | tail -1
for rep in 1 2 3; do for c in c1; do timeout 300 python bench.py --config $c --no-cpu-baseline --no-variants > $O/b_$c.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/b_$c.json')); print('$c', d['ms_per_step'], d['windows']['wall_ms_per_step_min_median_max'], d['kernel_ms_per_step'])"; done; done | tee $O/b.txt
bash tools/gpu_kstats.sh c1 preprocess_bwd deep_tile duplicate render preprocess_fwd | tee $O/ks_c1.txt
