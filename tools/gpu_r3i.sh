# A/B of two library builds on c2 / c5 + the sort tests
LIBS="old new" bash tools/gpu_ab_lib.sh ${CFGS:-c2 c5}
timeout 1500 python -m pytest tests/test_gpu_bin.py tests/test_gpu_parity.py -m gpu -x -q -k "bin or sort or flavour or oversize or deep" 2>&1 | tail -3
