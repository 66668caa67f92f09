# rocprofv3 evidence for one bench config: kernel stats + FETCH_SIZE + WRITE_SIZE passes (separate runs, no graphs)   usage: gpu_profile.sh <config> [steps]
set -x
c=$1; steps=${2:-5}
O=$GRAFT_REPO_ROOT/gpurun_out/prof_$c; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --config $c --no-cpu-baseline --no-variants --steps $steps --warmup 3"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$c -o $c -- $B > $O/bench_under_rocprof.json 2> $O/ks.err
cp $(find /tmp/ks_$c -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- $B > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- $B > /dev/null 2> $O/pmc_write.err
# what leaves the L2 for the memory side (round 5; there is no Infinity-Cache hit counter in rocprofv3's list on this stack: gpurun_out/counters_avail.txt)
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $O/pmc_dram -o d -- $B > /dev/null 2> $O/pmc_dram.err
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
cd $GRAFT_REPO_ROOT
$B > $O/bench_plain.json 2>/dev/null
python tools/pmc_summary.py $c $O/pmc_fetch $O/pmc_write $O/kernel_stats.csv $O/bench_plain.json $O/pmc_$c.json $O/pmc_dram | tee $O/table.md
# (the per-dispatch counter tables are tens of MB per config: gpurun copies gpurun_out/ back only below 64 MiB)
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_dram /tmp/ks_$c
