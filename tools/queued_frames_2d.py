#!/usr/bin/env python3
"""GPU box: ms per 2D frame of a sequence queued back to back on one context.  usage: tools/queued_frames_2d.py model size [size ...]   (FHIP_* options as usual)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fidget_amd as F
model = sys.argv[1]
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
shape = F.Shape.from_vm(os.path.join(ROOT, "models", model), hip=hip)
for n in [int(a) for a in sys.argv[2:]]:
    out = torch.zeros((n, n), dtype=torch.float32, device="cuda")
    F.render2d(shape, n, out=out); hip.sync(); alone = out.clone()
    for _ in range(12): F.render2d(shape, n, out=out)
    hip.sync()
    res = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): F.render2d(shape, n, out=out)
        hip.sync(); res.append((time.perf_counter() - t0) / 200 * 1e3)
    print(f"2D {model} {n}^2: {min(res):.4f} ms per queued frame; equal {bool(torch.equal(out.view(torch.int32), alone.view(torch.int32)))}; lane frames {hip.lane_frames()}", flush=True)
