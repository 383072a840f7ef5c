R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr; rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
python - <<'PY'
import csv, glob, re
f = sorted(glob.glob("/tmp/tr/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fin = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_finish3d")]
a, b = fin[1] + 1, fin[2] + 1
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if (s - t0) / 1e3 > 1000 and (s - t0) / 1e3 < 2400:
        print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f}  q{r.get('Queue_Id','?'):>3} {name}")
print("frame", (int(rows[b-1]["End_Timestamp"]) - t0) / 1e3)
PY
