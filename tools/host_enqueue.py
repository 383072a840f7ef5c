#!/usr/bin/env python3
"""GPU box: host time to queue one asynchronous 1024^3 frame (the call returns when everything is queued) next to the frame period."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fidget_amd as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
shape = F.Shape.from_vm(os.path.join(ROOT, "models", "prospero.vm"), hip=hip)
out = torch.zeros((n, n, 4), dtype=torch.int32, device="cuda")
for _ in range(5):
    F.render3d(shape, n, out=out)
torch.cuda.synchronize()
K = 50
ts = []
t0 = time.perf_counter()
for _ in range(K):
    a = time.perf_counter()
    F.render3d(shape, n, out=out)
    ts.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
ts.sort()
print(f"host enqueue per frame: median {ts[K // 2] * 1e3:.3f} ms, min {ts[0] * 1e3:.3f}, max {ts[-1] * 1e3:.3f}; loop {(t1 - t0) / K * 1e3:.3f} ms per frame; with final sync {(t2 - t0) / K * 1e3:.3f} ms per frame")
