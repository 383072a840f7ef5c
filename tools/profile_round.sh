#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel stats of the bench command, then two separate
# PMC passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass) with --kernel-trace only.
# usage: tools/profile_round.sh <tag>      (writes gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r01}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 5 --warmup 1 --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o s -- $CMD > $OUT/stats_run.log 2>&1
cp /tmp/p_stats/s_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || find /tmp/p_stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
grep '^{"metric"' $OUT/stats_run.log > $OUT/bench_under_rocprof.json
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/p_$C -o c -- $CMD > $OUT/pmc_$C.log 2>&1
  F=$(find /tmp/p_$C -name "*counter_collection.csv" | head -1)
  python - "$F" "$C" > $OUT/pmc_$C.txt <<'PY'
import csv, sys, collections, re
f, c = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != c: continue
    k = re.sub(r"\(.*", "", r["Kernel_Name"])
    tot[k] += float(r["Counter_Value"]); n[k] += 1
print(f"# {c}: sum over dispatches of the whole run (1 warm-up + 5 timed + 3 profiled frames = 9), KB as reported by rocprofv3 (no correction applied)")
for k in sorted(tot, key=lambda k: -tot[k]):
    print(f"{tot[k]:16.0f}  dispatches {n[k]:6d}  per-dispatch {tot[k]/n[k]:14.1f}  {k}")
PY
done
cd $R
ls -la $OUT
