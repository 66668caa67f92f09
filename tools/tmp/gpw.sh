O=gpurun_out/gpw; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "gather or c1 or golden or small" --deselect tests/test_gpu_parity.py::test_full_size_record_matches_kernel_sources 2>&1 | tail -4
timeout 200 python tools/fuzz_bwd_gather.py 40 2>&1 | tail -1
for m in 0 2 0 2; do SIGMAN_BWD_GATHER=$m timeout 300 python - $m <<'PY' 2>/dev/null
import sys, json, subprocess, os
sys.path.insert(0, ".")
from sigman_release_amd import _cabi
PY
done
for rep in 1 2; do for c in c1; do timeout 300 python bench.py --config $c --no-cpu-baseline --no-variants > $O/b_$c.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/b_$c.json')); print('$c', d['ms_per_step'], d['windows']['wall_ms_per_step_min_median_max'], d['kernel_ms_per_step'])"; done; done | tee $O/b.txt
bash tools/gpu_kstats.sh c1 preprocess_bwd deep_tile duplicate render preprocess_fwd | tee $O/ks_c1.txt
