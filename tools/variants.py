#!/usr/bin/env python3
"""GPU box: A/B of assembly variants built by tools/build_variant.py - one process each (tools/variant_run.py), results side by side.
usage: tools/variants.py [--model m.vm] [--size n] [--path general|default] [--stats] name[@K=V,...] ...   ("embedded" = the library's own; K=V: environment)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
model, size, path, stats = "prospero.vm", "1024", "general", False
names = []
while args:
    a = args.pop(0)
    if a == "--model": model = args.pop(0)
    elif a == "--size": size = args.pop(0)
    elif a == "--path": path = args.pop(0)
    elif a == "--stats": stats = True
    else: names.append(a)
rows = []
for i, nm in enumerate(names):
    env = dict(os.environ)
    nm, _, kv = nm.partition("@")           # name@K=V,K=V: environment of the run (context options: FHIP_<NAME>)
    for e in filter(None, kv.split(",")):
        env[e.split("=")[0]] = e.split("=")[1]
    if nm != "embedded":
        env["FHIP_INTERP_CO"] = os.path.join(ROOT, "fidget_amd", "csrc", "_gen", "variants", nm + ".co")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "variant_run.py"), model, size, path] + (["stats"] if stats and i == 0 else [])
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode or not line:
        rows.append({"name": nm, "error": (p.stderr or p.stdout)[-2000:]})
        print(nm, "FAILED", rows[-1]["error"], flush=True)
        continue
    r = json.loads(line[-1]); r["name"] = nm + ("@" + kv if kv else "")
    nm = r["name"]
    rows.append(r)
    print(nm, model, path, "columns ms/launch", r["kernel_ms_per_launch"].get("fh_columns"), "queued ms/frame", r["queued_ms_per_frame"], "sha", r["sha"],
          {k: v for k, v in r["kernel_ms_per_launch"].items() if k != "fh_columns"}, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
tag = f"{model.split('.')[0]}_{size}_{path}"
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", f"variants_{tag}.json"), "w"), indent=1)
