#!/usr/bin/env python3
"""print a rocprofv3 kernel_stats.csv compactly: kernel name (template args kept), calls, average us      usage: kstats.py file.csv [min_calls]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
mc = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for r in rows:
    nm = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'([\w:]+(<[^()]*?>)?)\(', nm)
    n = m.group(1) if m else nm[:50]
    if int(r['Calls']) >= mc:
        print(f"{n[:52]:52s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} us  tot {float(r['TotalDurationNs'])/1e6:9.3f} ms")
