#!/bin/bash
# GPU box: tools/_bin/fetch_calib (build line in tools/fetch_calib.cpp) under rocprofv3, one pass per counter -> gpurun_out/fetch_calib.txt:
# what FETCH_SIZE / WRITE_SIZE report per kernel launch against the bytes the kernel is known to move
R=$PWD
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/fc_$C
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/fc_$C -o c -- $R/tools/_bin/fetch_calib > /tmp/fc_$C.log 2>&1
done
python - <<'PY' > $R/gpurun_out/fetch_calib.txt
import csv, glob, collections
print(open("/tmp/fc_FETCH_SIZE.log").read().strip().splitlines()[-1])
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/fc_{C}/**/*counter_collection.csv", recursive=True)[0]
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == C:
            k = r["Kernel_Name"].split("(")[0]
            tot[k] += float(r["Counter_Value"]); n[k] += 1
    for k in sorted(tot):
        if k.startswith("calib"):
            print(f"{C:10s} {k:14s} launches {n[k]}  counter per launch {tot[k] / n[k]:14.1f}  (x 1024 = {tot[k] / n[k] * 1024 / 2**30:.3f} GiB; known: 2 GiB moved)")
PY
cat $R/gpurun_out/fetch_calib.txt
