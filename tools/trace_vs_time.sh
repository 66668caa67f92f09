# Do the kernels of a long single-view run slow down over time, or does the host fall behind?  kernel trace of bench.py --steps N: per time bin the
# median duration of the dominant kernel and the period between its launches.       usage: trace_vs_time.sh [steps]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp; rm -rf /tmp/tvt
rocprofv3 --kernel-trace --output-format csv -d /tmp/tvt -o x -- python $R/bench.py --no-cpu-baseline --no-variants --no-settle --steps ${1:-6000} > /tmp/tvt_bench.json 2>/dev/null
python3 - <<'PY'
import csv, glob, statistics
f = glob.glob("/tmp/tvt/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
for pat in ("render_fwd_seg", "render_bwd_bucket", "preprocess_fwd_kernel"):
    k = sorted((s, e) for s, e, n in rows if pat in n)
    t0 = k[0][0]
    nb = 24; span = (k[-1][0] - t0) / nb
    print(pat, len(k), "launches over %.2f s" % ((k[-1][0] - t0) / 1e9))
    for b in range(nb):
        sel = [(s, e) for s, e in k if t0 + b * span <= s < t0 + (b + 1) * span]
        if len(sel) < 3: continue
        dur = statistics.median(e - s for s, e in sel) / 1e3
        per = statistics.median(sel[i + 1][0] - sel[i][0] for i in range(len(sel) - 1)) / 1e3
        print("  t=%.2fs  n=%5d  kernel %.1f us  period %.1f us" % (b * span / 1e9, len(sel), dur, per))
PY
python3 -c "
import json; d=json.loads([l for l in open('/tmp/tvt_bench.json') if l.startswith('{')][-1]); print('bench ms_per_step', d['ms_per_step'], 'steps', d['steps'])"
