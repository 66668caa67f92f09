O=gpurun_out/qm; mkdir -p $O
SIGMAN_QMASK_EXP=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or c2 or c1 or c5" --deselect tests/test_gpu_parity.py::test_full_size_record_matches_kernel_sources 2>&1 | tail -3
for e in 0 1 0 1; do for c in c2 c5 c1; do echo "exp $e $c"; SIGMAN_QMASK_EXP=$e bash tools/gpu_kstats.sh $c render_fwd_seg qmask deep_tile 2>&1 | grep -v "seg_kernel<0>"; done; done | tee $O/ks.txt
