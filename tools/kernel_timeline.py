"""Dev tool (GPU box): kernel timeline of one bench step from a rocprofv3 --kernel-trace CSV: start offset, duration and gap to the
previous kernel for every dispatch of the LAST complete step.   usage: python tools/kernel_timeline.py <kernel_trace.csv> [first kernel substring]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "preprocess_fwd_kernel"
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
k = int(sys.argv[3]) if len(sys.argv) > 3 else 3
a, b = starts[-k], starts[-k + 1]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = None
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if prev_end is None else f"gap {(s - prev_end) / 1e3:6.1f}"
    print(f"{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  {gap:12s} {r['Kernel_Name'][:70]}")
    prev_end = e
print(f"step span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
