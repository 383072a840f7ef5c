#!/usr/bin/env python3
"""GPU box: exhaustive (all 2^32 inputs) accuracy sweep of the device's transcendental opcodes against glibc's f32 libm."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fidget_amd as F
import oracle as O
from test_gpu_math import sweep, OPS
hip = F.HipContext(0)
n = 1 << 26
res = {}
for op in (sys.argv[1:] or OPS):
    t0 = time.time()
    acc = {"max_ulp": 0, "differ": 0, "over_1_ulp": 0, "inputs": 1 << 32}
    for c in range(64):
        r = sweep(F, O, hip, op, (c * n) & 0xFFFFFFFF, 1, n)
        acc["differ"] += r["differ"]; acc["over_1_ulp"] += r["over_1_ulp"]
        if r["max_ulp"] > acc["max_ulp"]:
            acc["max_ulp"], acc["worst_input_bits"] = r["max_ulp"], hex(r["worst_input_bits"])
    acc["seconds"] = round(time.time() - t0, 1)
    res[op] = acc
    print(op, json.dumps(acc), flush=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "math_sweep.json"), "w"), indent=1)
