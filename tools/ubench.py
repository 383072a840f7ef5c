#!/usr/bin/env python3
"""GPU box: instruction-cost micro-benchmarks (fidget_amd/csrc/gen_ubench.py): shader clocks per pattern at 1, 2, 3 and 4 waves per SIMD.
usage: tools/ubench.py [first test [last test]]   (FHIP_INTERP_CO may name a variant's code object)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fidget_amd", "csrc"))
import numpy as np
import fidget_amd as F
from gen_ubench import TESTS
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
last = int(sys.argv[2]) if len(sys.argv) > 2 else len(TESTS) - 1
hip = F.HipContext(0)
res = []
for t, (desc, _) in enumerate(TESTS):
    if not first <= t <= last:
        continue
    row = {"test": t, "pattern": desc}
    for waves in (256, 1024, 2048, 3072, 4096):
        out = np.zeros(waves, np.float32)
        for _ in range(2):
            hip.check(F.lib().fhip_debug_ubench(hip._h, t, 500, waves, out.ctypes.data_as(F.C.c_void_p)))
        row[f"waves{waves}"] = [round(float(out.mean()), 2), round(float(out.min()), 2), round(float(out.max()), 2)]
    res.append(row)
    print(json.dumps(row), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "ubench.json"), "w"), indent=1)
