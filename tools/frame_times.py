#!/usr/bin/env python3
"""GPU box: the distribution of per-frame times (HIP events on the caller's stream) of K queued frames of the default path: percentiles,
the first frames in order, and where the slow ones sit.  usage: tools/frame_times.py [K]"""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import fidget_amd as F
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
shape = F.Shape.from_vm(os.path.join(ROOT, "models", "prospero.vm"), hip=hip)
n = 1024
out = torch.zeros((n, n, 4), dtype=torch.int32, device="cuda")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for _ in range(8):
    F.render3d(shape, n, out=out)
torch.cuda.synchronize()
gc.collect(); gc.disable()
marks = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
host = []
t0 = time.perf_counter()
marks[0].record()
for i in range(K):
    h0 = time.perf_counter()
    F.render3d(shape, n, out=out)
    marks[i + 1].record()
    host.append((time.perf_counter() - h0) * 1e3)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) * 1e3
ms = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(K)])
print(os.environ.get("TAG", ""), "wall/K", round(wall / K, 4), "mean", round(ms.mean(), 4), "percentiles 5/25/50/75/95/99", [round(float(np.percentile(ms, p)), 3) for p in (5, 25, 50, 75, 95, 99)], "max", round(ms.max(), 3))
print(" first 40 frames:", [round(float(x), 2) for x in ms[:40]])
print(" host enqueue ms, first 40:", [round(x, 2) for x in host[:40]])
