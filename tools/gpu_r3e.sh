# round-3 experiment: A/B (old/new libraries) + the whole gpu suite (observed-record mode)
LIBS="old new" bash tools/gpu_ab_lib.sh ${CFGS:-c3 c2 c5}
SIGMAN_RECORD_OBSERVED=1 timeout 2200 python -m pytest tests -m gpu -x -q --timeout=1200 2>&1 | tail -8
