# k-NN front end: parity tests, per-kernel times, end-to-end front-end time old vs new library
timeout 900 python -m pytest tests/test_gpu_callers.py tests/test_gpu_reference_calls.py -m gpu -x -q 2>&1 | tail -3
bash tools/gpu_knn_time.sh c2 c3
for v in old new; do SIGMAN_GSPLAT_LIB=$PWD/tools/ab/$v.so python tools/time_knn.py 2>&1 | tail -4; done
