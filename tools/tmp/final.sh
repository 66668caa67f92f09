bash tools/gpu_final.sh r06 2>&1 | grep -v amdgpu.ids | tail -60
for c in c2 c3; do bash tools/pmc_sq.sh $c render_fwd render_bwd > gpurun_out/sq_$c.log 2>&1; tail -3 gpurun_out/sq_$c.log; done
bash tools/gpu_profile.sh c1 5 > gpurun_out/prof_c1.log 2>&1; tail -9 gpurun_out/prof_c1.log
O=gpurun_out/r06_fuzz; mkdir -p $O
timeout 200 python tools/fuzz_parity.py 120 40000 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|np.transpose" > $O/parity.txt
timeout 120 python tools/fuzz_bin_modes.py 60 2>&1 | tail -1 > $O/bin_modes.txt
SIGMAN_SORT_DEEP=1 FUZZ_VIEWS=1,2 timeout 120 python tools/fuzz_bin_modes.py 60 2>&1 | tail -1 > $O/bin_modes_deep.txt
timeout 150 python tools/fuzz_determinism.py 90 11 2>&1 | tail -1 > $O/determinism.txt
FUZZ_BIG=1 timeout 150 python tools/fuzz_determinism.py 90 12 2>&1 | tail -1 > $O/determinism_big.txt
timeout 120 python tools/fuzz_fused_step.py 60 101 2>&1 | tail -1 > $O/fused.txt
timeout 120 python tools/fuzz_bwd_gather.py 60 2>&1 | tail -1 > $O/bwd_gather.txt
timeout 120 python tools/fuzz_render.py 45 2>&1 | tail -1 > $O/render.txt
tail -n 3 $O/*.txt
