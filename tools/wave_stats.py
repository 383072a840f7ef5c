#!/usr/bin/env python3
"""Render one frame and print the per-kernel-kind wave busy statistics (GPU box)."""
import os, sys, json
os.environ.setdefault("FHIP_PROBE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fidget_amd as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = sys.argv[2] if len(sys.argv) > 2 else "prospero.vm"
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
shape = F.Shape.from_vm(os.path.join(ROOT, "models", model), hip=hip)
out = torch.zeros((n, n, 4), dtype=torch.int32, device="cuda")
for _ in range(2):
    F.render3d(shape, n, out=out)
hip.sync()
for k, v in hip.wave_stats().items():
    v["busy_us_mean"] = v["busy_us_sum"] / max(v["waves"], 1)
    print(k, json.dumps(v))
print(hip.tile_phases)
print(hip.counters())
lv = hip.last_leaves()
import numpy as np
print("leaves of the last slab:", len(lv), "mean len", lv["len"].mean(), "mean regs", lv["regs"].mean())
for q in (4, 6, 8, 12, 16, 24, 32):
    print(f"  regs <= {q}: {(lv['regs'] <= q).mean() * 100:.1f} %   len <= {q}: {(lv['len'] <= q).mean() * 100:.1f} %")
print("  len percentiles 50/90/99/max:", np.percentile(lv["len"], [50, 90, 99, 100]))
