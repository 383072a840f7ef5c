#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-dispatch durations of the last frame, in order.

usage: trace_summary.py <dir with *_kernel_trace.csv> [n_last]
"""
import csv, glob, sys, re
d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 400
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n_last:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"])
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:9.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size','?')):>8}  {name}")
