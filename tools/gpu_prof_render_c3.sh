R=$PWD; cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r3 -o c3 -- python $R/tools/time_render.py --subjects 8 --views 8 --iters 6 > /tmp/prof_r3.log 2>&1; cd $R
python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/prof_r3/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:26]:
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"]) / 1e3:9.2f} pct {r["Percentage"]}')
PY
