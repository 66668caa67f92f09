# SQ counters of the compositing kernels inside a bench config (separate rocprofv3 --pmc passes next to --kernel-trace only) -> gpurun_out/sq_<config>/r06_sq_<config>.json
#   usage: pmc_sq.sh <config> <kernel name substring> [more substrings...]
export TMPDIR=/tmp
c=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/sq_$c; mkdir -p $O; cd /tmp; rm -f $O/lines.txt
B="python $GRAFT_REPO_ROOT/bench.py --config $c --no-cpu-baseline --no-variants --steps 3 --warmup 2"
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_IFETCH" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_LEVEL_WAVES SQ_INSTS_SMEM"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/sq_$n -o x -- $B > /dev/null 2> $O/err_$n.log
  python3 - "$@" >> $O/lines.txt <<PY
import csv, glob, collections, sys
f = glob.glob("/tmp/sq_$n/**/*counter_collection.csv", recursive=True)
for pat in sys.argv[1:]:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if pat in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): print(pat, k, len(v), "%.6g" % sorted(v)[len(v)//2])
PY
done
# kernel durations of a plain kernel trace of the same command
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sq_ks -o x -- $B > $O/bench.json 2> $O/err_ks.log
python3 - "$c" "$O" "$@" <<'PY'
import csv, glob, json, sys
c, O, pats = sys.argv[1], sys.argv[2], sys.argv[3:]
out = {"config": c, "command": "rocprofv3 --pmc <4 SQ counters per pass> --kernel-trace -- python bench.py --config %s --no-cpu-baseline --no-variants --steps 3 --warmup 2" % c,
       "note": "median over the launches of the kernel in the run; SQ counters are summed over all SEs/XCDs by rocprofv3; *_CYCLES in units of 4 clocks per the counter definitions", "kernels": {}}
for line in open(O + "/lines.txt"):
    pat, name, n, val = line.split()
    out["kernels"].setdefault(pat, {"launches": int(n)})[name] = float(val)
ks = glob.glob("/tmp/sq_ks/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(ks[0])):
    for pat in pats:
        if pat in r["Name"] and "avg_us" not in out["kernels"].get(pat, {}): out["kernels"].setdefault(pat, {})["avg_us"] = float(r["AverageNs"]) / 1e3
try: out["bench"] = {k: json.load(open(O + "/bench.json"))[k] for k in ("ms_per_step", "kernel_ms_per_step")}
except Exception: pass
json.dump(out, open(O + "/r06_sq_%s.json" % c, "w"), indent=1)
print(json.dumps(out, indent=1))
PY
