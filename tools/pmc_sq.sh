# SQ counters of one kernel inside a bench config (separate rocprofv3 --pmc passes, no stats/trace domains besides kernel-trace)
#   usage: pmc_sq.sh <config> <kernel name substring> [more substrings...]
export TMPDIR=/tmp
c=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/sq_$c; mkdir -p $O; cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --config $c --no-cpu-baseline --no-variants --steps 3 --warmup 2"
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_IFETCH" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_LEVEL_WAVES SQ_INSTS_SMEM"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/sq_$n -o x -- $B > /dev/null 2> $O/err_$n.log
  python3 - "$@" <<PY
import csv, glob, collections, sys
f = glob.glob("/tmp/sq_$n/**/*counter_collection.csv", recursive=True)
for pat in sys.argv[1:]:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if pat in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): print(pat, k, "launches", len(v), "median %.4g" % sorted(v)[len(v)//2])
PY
done
