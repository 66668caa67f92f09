# round-3 experiment: A/B (old/new libraries) + parity
LIBS="old new" bash tools/gpu_ab_lib.sh ${CFGS:-c3 c2 c5}
SIGMAN_RECORD_OBSERVED=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parallel.py -m gpu -x -q 2>&1 | tail -8
