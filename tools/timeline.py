#!/usr/bin/env python3
"""Timeline of the last frames of a rocprofv3 --kernel-trace run of bench.py: per queue (= HIP stream) the kernels in
order with start offsets, durations and the idle gap before each, then busy time per queue per frame.

usage: timeline.py <dir with *_kernel_trace.csv> [frames, default 2] [k_finish3d occurrences to skip at the end, default 3; or +k: start
       at the k-th frame from the beginning of the run (the queued frames of bench.py come first)]
"""
import csv, glob, sys, re, collections
d = sys.argv[1]
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 2
from_start = len(sys.argv) > 3 and sys.argv[3].startswith("+")
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 3
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").split("<")[0]
rows.sort(key=lambda r: r["s"])
fin = [i for i, r in enumerate(rows) if r["n"] == "k_finish3d"]
fin = fin[:skip + nfr + 1] if from_start else (fin[:len(fin) - skip] if skip else fin)
lo = rows[fin[-nfr - 1]]["e"]
hi = rows[fin[-1]]["e"]
sel = [r for r in rows if r["e"] > lo and r["s"] <= hi]
qs = sorted({r["Queue_Id"] for r in sel})
print(f"{nfr} frames: {(hi - lo) / 1e3 / nfr:.1f} us per frame; queues {qs}")
last = {}
for r in sel:
    q = qs.index(r["Queue_Id"])
    gap = (r["s"] - last[q]) / 1e3 if q in last else 0.0
    last[q] = r["e"]
    print(f"{(r['s'] - lo) / 1e3:9.1f} us  q{q} {'    ' * q}+{(r['e'] - r['s']) / 1e3:7.1f}  (gap {gap:7.1f})  {r['n']}")
busy = collections.defaultdict(float)
for r in sel:
    busy[r["Queue_Id"]] += (min(r["e"], hi) - max(r["s"], lo)) / 1e3
for q in qs:
    print(f"queue {q}: busy {busy[q] / nfr:.1f} us per frame")
