#!/bin/bash
# GPU box: where the waves of the dominant kernels wait - SQ activity / wait / instruction-cache counters of the bench frames,
# one rocprofv3 --pmc pass per group, per-kernel sums -> gpurun_out/pmc_stalls.txt
set -u
R=$PWD
OUT=$R/gpurun_out/pmc_stalls
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# usage: tools/pmc_stalls.sh [default|general]
if [ "${1:-default}" = general ]; then FLAG="--only-general"; else FLAG="--no-general"; fi
CMD="python $R/bench.py --steps 5 --warmup 1 --no-cpu $FLAG"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/ps_$i -o c -- $CMD > $OUT/run_$i.log 2>&1
  F=$(find /tmp/ps_$i -name "*counter_collection.csv" | head -1)
  cp "$F" $OUT/pmc_$i.csv 2>/dev/null
done
cd $R
python - "$OUT" "${1:-default}" <<'PY'
import csv, sys, collections, re, glob, os
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(out, "pmc_*.csv")):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").split("<")[0]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
lines = []
for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", 0))[:8]:
    d = tot[k]
    lines.append(k + "  (per launch)")
    for c in sorted(d):
        lines.append(f"    {c:28s} {d[c] / max(n[k][c], 1):14.4g}")
open(os.path.join(out, "..", "pmc_stalls_%s.txt" % (sys.argv[2] if len(sys.argv) > 2 else "default")), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
