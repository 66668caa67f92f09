# SQ counters of the per-tile sort micro-benchmark   usage: pmc_bts.sh <case e.g. 12000x1000>
export TMPDIR=/tmp BTS_ONLY=$1
O=$GRAFT_REPO_ROOT/gpurun_out/bts_pmc; mkdir -p $O; cd /tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM" "SQ_IFETCH SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/bp_$n -o x -- $GRAFT_REPO_ROOT/tools/micro/bts > /dev/null 2> $O/err_$n.log
  python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/bp_$n/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "tile_sort_regs" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print(k, "launches", len(v), "median %.4g" % sorted(v)[len(v)//2])
PY
done
