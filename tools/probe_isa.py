#!/usr/bin/env python3
"""GPU box: run the ISA probe kernel (gen_interp.py gen_probe) and print its rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fidget_amd as F
hip = F.HipContext(0)
out = np.zeros((16, 64), np.float32)
hip.check(F.lib().fhip_debug_probe(hip._h, out.ctypes.data_as(F.C.c_void_p)))
names = ["pk_add src0 rel lo (20)", "hi (40)", "pk_mul dst rel: v32 (0)", "v34 (16)", "v35 (64)", "in place lo (12)", "hi (24)", "pk_mov lo", "hi",
         "readlane under SRC0 rel (64 = immune, 128 = indexed)", "readlane under DST rel (64)", "s17 after DST rel readlane (0)",
         "readfirstlane under SRC0 rel (64 = immune)"]
for i, n in enumerate(names):
    print(f"row {i:2d} {out[i, 0]:8.1f}  {n}")
