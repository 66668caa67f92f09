# A/B of two builds of libsigman_gsplat.so on one box (dev: build the old tree into tools/ab/old.so, the new one into tools/ab/new.so)
# usage: gpu_ab_lib.sh [configs]
O=gpurun_out/ab; mkdir -p $O
for rep in 1 2 3; do for v in ${LIBS:-old new}; do for c in ${@:-c3 c4}; do SIGMAN_PY_NODE=1 SIGMAN_GSPLAT_LIB=$PWD/tools/ab/$v.so timeout 600 python bench.py --config $c --no-variants --no-cpu-baseline > $O/b_${c}_$v.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/b_${c}_$v.json')); print('$v', '$c', d['ms_per_step'], d['kernel_ms_per_step'])"; done; done; done
