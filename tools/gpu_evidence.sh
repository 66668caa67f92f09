# round evidence without the test suite: rocprofv3 kernel stats + PMC (FETCH / WRITE / DRAM-side) for c2..c5, SQ counters of the compositing
# kernels for c2..c4, every bench line (+ the 2-rank gloo plumbing lines)        usage (GPU box): bash tools/gpu_evidence.sh <tag>
T=${1:-ev}; cd $GRAFT_REPO_ROOT
for c in c2 c3 c4 c5; do bash tools/gpu_profile.sh $c 5 > gpurun_out/prof_$c.log 2>&1; tail -9 gpurun_out/prof_$c.log; done
for c in c2 c3 c4; do bash tools/pmc_sq.sh $c render_fwd render_bwd > gpurun_out/sq_$c.log 2>&1; done
bash tools/gpu_bench_all.sh $T > gpurun_out/bench_all_$T.log 2>&1; grep -E "^c[1-5] " gpurun_out/bench_all_$T.log | cut -c1-300
