#!/usr/bin/env python3
"""GPU box: ms per frame of a sequence queued back to back on one context (what bench.py's c3_bear leg times), and whether the last
image equals a frame rendered alone.  usage: tools/queued_frames.py [size] [model] [frames]   (options through FHIP_* as usual)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fidget_amd as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
model = sys.argv[2] if len(sys.argv) > 2 else "bear.vm"
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 40
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
shape = F.Shape.from_vm(os.path.join(ROOT, "models", model), hip=hip)
out = torch.zeros((n, n, 4), dtype=torch.int32, device="cuda")
F.render3d(shape, n, out=out)
hip.sync()
alone = out.clone()
for _ in range(6):
    F.render3d(shape, n, out=out)
hip.sync()
res = []
for rep in range(3):
    out.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        F.render3d(shape, n, out=out)
    hip.sync()
    res.append((time.perf_counter() - t0) / frames * 1e3)
print(f"{model} {n}^3: {min(res):.3f} ms per queued frame (runs: {' '.join(f'{r:.3f}' for r in res)}), last image equals a frame alone: {bool(torch.equal(out, alone))}", flush=True)
ms = (F.C.c_float * 3)()
ln = F.C.c_int(0)
ph = F.lib().fhip_debug_lane_tune(hip._h, ms, F.C.byref(ln))
print(f"  arrangement tuner: phase {ph}, stage pipeline {ms[0]:.3f} ms per frame, lanes {ms[1]:.3f}, stage pipeline again {ms[2]:.3f}, decision: {'lanes' if ln.value else 'stage pipeline'}; {F.lib().fhip_debug_lane_frames(hip._h)} frames went to lanes", flush=True)
