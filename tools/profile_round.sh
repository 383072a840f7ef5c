#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel stats of the bench command, then separate PMC passes
# (FETCH_SIZE, WRITE_SIZE, SQ instruction counters; they do not fit one pass) with --kernel-trace only -
# once for the default path and once for the general path (bench.py --only-general: column-invariance short cuts off).
# usage: tools/profile_round.sh <tag>      (writes gpurun_out/prof_<tag>/, incl. traffic.json for bench.py's roofline)
set -u
TAG=${1:-r03}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for PATHNAME in default general; do
  if [ $PATHNAME = default ]; then FLAG="--no-general"; else FLAG="--only-general"; fi
  CMD="python $R/bench.py --steps 5 --warmup 1 --no-cpu $FLAG"
  if [ -z "${PROFILE_ROUND_QUICK:-}" ]; then
  rm -rf /tmp/p_stats
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o s -- $CMD > $OUT/stats_run_$PATHNAME.log 2>&1
  cp /tmp/p_stats/s_kernel_stats.csv $OUT/kernel_stats_$PATHNAME.csv 2>/dev/null || find /tmp/p_stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$PATHNAME.csv \;
  grep '^{"metric"' $OUT/stats_run_$PATHNAME.log > $OUT/bench_under_rocprof_$PATHNAME.json
  fi
  # (PROFILE_ROUND_QUICK=1: only the passes bench.py's roofline reads - traffic and instruction counts)
  for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" ${PROFILE_ROUND_QUICK:+SKIP} "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM"; do
    if [ "$C" = SKIP ]; then break; fi
    N=$(echo $C | tr ' ' '_')
    rm -rf /tmp/p_$N
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/p_$N -o c -- $CMD > $OUT/pmc_${PATHNAME}_$N.log 2>&1
    F=$(find /tmp/p_$N -name "*counter_collection.csv" | head -1)
    cp "$F" $OUT/pmc_${PATHNAME}_$N.csv 2>/dev/null
  done
done
cd $R
python - "$OUT" <<'PY'
import csv, sys, collections, re, json, glob, os
out = sys.argv[1]
sys.path.insert(0, "tools")
from src_hash import source_hash
res = {"_comment": "per-kernel PMC totals of `bench.py --steps 5 --warmup 1 --no-cpu [--no-general | --only-general]` under rocprofv3 --pmc, one pass per counter "
                   "group and path; frames = launches of k_finish3d in the pass; FETCH_SIZE / WRITE_SIZE in KB as reported (bench.py doubles FETCH_SIZE as "
                   "MI355X_MICROARCH.md prescribes for gfx950)",
       "source_hash": source_hash()}
lines = []
for path in ("default", "general"):
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
    # (the LAST frames of each pass only: the first frames of a context may run out of tape arena - it starts small and grows, the frames
    # are right but their tiles keep their parents' tapes - and would count 2-4 x the instructions of a steady frame into the mean)
    LAST = 16
    for f in glob.glob(os.path.join(out, f"pmc_{path}_*.csv")):
        rows = list(csv.DictReader(open(f)))
        name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").split("<")[0]
        ends = sorted({int(r["End_Timestamp"]) for r in rows if name(r) == "k_finish3d"})
        after = ends[-LAST - 1] if len(ends) > LAST else 0
        for r in rows:
            if int(r["Start_Timestamp"]) <= after:
                continue
            k = name(r)
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    frames = max(n["k_finish3d"].values()) if n["k_finish3d"] else 1
    res[path] = {"frames": frames}
    lines.append(f"---- {path} path, {frames} frames per pass")
    for k in sorted(tot, key=lambda k: -tot[k].get("SQ_INSTS_VALU", 0)):
        d = tot[k]; c = n[k]
        launches = max(c.values())
        e = {"launches_per_frame": launches / frames}
        if "FETCH_SIZE" in d: e["fetch_kb_per_frame"] = d["FETCH_SIZE"] / frames
        if "WRITE_SIZE" in d: e["write_kb_per_frame"] = d["WRITE_SIZE"] / frames
        for key, name in (("SQ_INSTS_VALU", "valu_per_launch"), ("SQ_INSTS_SALU", "salu_per_launch"), ("SQ_INSTS_SMEM", "smem_per_launch"),
                          ("SQ_WAVES", "waves_per_launch"), ("SQ_WAVE_CYCLES", "wave_cycles_per_launch"), ("SQ_BUSY_CYCLES", "busy_cycles_per_launch")):
            if key in d: e[name] = d[key] / max(c[key], 1)
        res[path][k] = e
        lines.append(f"{k:40s} " + "  ".join(f"{a}={b:.4g}" for a, b in e.items()))
json.dump(res, open(os.path.join(out, "traffic.json"), "w"), indent=1)
open(os.path.join(out, "pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
ls -la $OUT
