#!/usr/bin/env python3
"""GPU box: per-frame times (HIP events on the caller's stream) of queued frames, default path then general path in one process -
where do frames that take many times the median come from?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import fidget_amd as F
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
shape = F.Shape.from_vm(os.path.join(ROOT, "models", "prospero.vm"), hip=hip)
n = 1024
out = torch.zeros((n, n, 4), dtype=torch.int32, device="cuda")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100


def run(tag, warm=5):
    for _ in range(warm):
        F.render3d(shape, n, out=out)
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    host = []
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(K):
        h0 = time.perf_counter()
        F.render3d(shape, n, out=out)
        host.append((time.perf_counter() - h0) * 1e3)
        marks[i + 1].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    ms = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(K)])
    slow = np.nonzero(ms > 2 * np.median(ms))[0]
    print(tag, "wall/K", round(wall / K, 4), "mean", round(ms.mean(), 4), "median", round(float(np.median(ms)), 4), "max", round(ms.max(), 3), "slow frames", [(int(i), round(float(ms[i]), 2), round(host[i], 2)) for i in slow][:10],
          "host enqueue max", round(max(host), 2), flush=True)


run("default")
hip.set_option("no_column_inv", 1)
run("general (first loop after the switch)")
run("general (second loop)")
hip.set_option("no_column_inv", 0)
run("default again")
