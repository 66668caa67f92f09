export TMPDIR=/tmp AFF=0,1,2,3; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -o pv -- python $GRAFT_REPO_ROOT/tools/profile_per_view.py > /tmp/pv.log 2>&1
grep "V=8" /tmp/pv.log
python3 - <<'PY'
import csv, re
rows = list(csv.DictReader(open('/tmp/pv/pv_kernel_stats.csv')))
for r in rows[:16]:
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", r["Name"]); print(f"{(m.group(1) if m else r['Name'][:60]):60s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={r['Percentage']}")
# timeline gaps for one step: take kernel trace, consecutive kernels on the stream
tr = list(csv.DictReader(open('/tmp/pv/pv_kernel_trace.csv')))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(tr); seg = tr[n//2:n//2+60]
t0 = int(seg[0]['Start_Timestamp'])
for a, b in zip(seg, seg[1:]):
    m = re.search(r"(\w+_kernel|\w+)", a['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::',''))
    print(f"{(int(a['Start_Timestamp'])-t0)/1e3:9.1f} us  dur {(int(a['End_Timestamp'])-int(a['Start_Timestamp']))/1e3:7.1f}  gap-to-next {(int(b['Start_Timestamp'])-int(a['End_Timestamp']))/1e3:6.1f}  {a['Kernel_Name'][:70]}")
PY
