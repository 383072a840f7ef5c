#!/usr/bin/env python3
"""GPU box, under rocprofv3 --kernel-trace --stats: frames of prospero.vm in the FEW-TILES regime, one at a time (waited for each, no
frame lanes): where the time of a small image of a large tape goes, kernel by kernel.
usage: small_frames.py 3d:512 | 3d:256 | 3d:128 | 2d:256 | 2d:64 | octant [frames] [model]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fidget_amd as F
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
hip.set_option("frame_lanes", 0)
what = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 5
model = sys.argv[3] if len(sys.argv) > 3 else "prospero.vm"
shape = F.Shape.from_vm(os.path.join(ROOT, "models", model), hip=hip)
if what == "octant":
    n, kw = 1024, {"block": (7, (2, 2, 2))}
    out = torch.zeros((n, n, 4), dtype=torch.int32, device="cuda")
    call = lambda: F.render3d(shape, n, out=out, **kw)
else:
    kind, n = what.split(":")
    n = int(n)
    if kind == "3d":
        out = torch.zeros((n, n, 4), dtype=torch.int32, device="cuda")
        call = lambda: F.render3d(shape, n, out=out)
    else:
        out = torch.zeros((n, n), dtype=torch.float32, device="cuda")
        call = lambda: F.render2d(shape, n, out=out)
call(); torch.cuda.synchronize()
ts = []
for _ in range(frames):
    t0 = time.perf_counter(); call(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(what, model, "ms per lone frame:", " ".join(f"{t:.3f}" for t in ts), flush=True)
