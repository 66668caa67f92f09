export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/sq1 -o s -- $GRAFT_REPO_ROOT/tools/micro/bts > /tmp/sq1.log 2>&1
python3 - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/sq1/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'tile_sort_regs' in r['Kernel_Name'] or 'tile_sort_dyn' in r['Kernel_Name']:
            k = ('regs' if 'regs' in r['Kernel_Name'] else 'lds-block') + ' grid=' + r['Grid_Size']
            rows[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in rows.items():
    print(k, {c: f"{sum(v)/len(v):.3g}" for c, v in d.items()})
PY
