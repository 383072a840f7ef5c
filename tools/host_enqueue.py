#!/usr/bin/env python3
"""GPU box: what ONE frame costs the host thread that queues it (fhip_render3d with a device output returns when the frame's kernels are
queued): the duration of every call in a run of frames queued back to back, against the rate the queue drains at.
usage: host_enqueue.py [size] [frames]       (FHIP_STATS=2: the library prints where the time inside the call goes, every 200 frames)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import fidget_amd as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 400
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
shape = F.Shape.from_vm(os.path.join(ROOT, "models", "prospero.vm"), hip=hip)
out = torch.zeros((n, n, 4), dtype=torch.int32, device="cuda")
for _ in range(150):
    F.render3d(shape, n, out=out)
hip.sync()
ts = []
t0 = time.perf_counter()
for _ in range(frames):
    a = time.perf_counter(); F.render3d(shape, n, out=out); ts.append(time.perf_counter() - a)
t1 = time.perf_counter()
hip.sync()
t2 = time.perf_counter()
ts = np.array(ts) * 1e3
print(f"{n}^3, {frames} frames: queued in {(t1 - t0) / frames * 1e3:.3f} ms per frame (the host), drained after {(t2 - t0) / frames * 1e3:.3f} ms per frame; "
      f"call durations ms: min {ts.min():.3f} median {np.median(ts):.3f} p90 {np.percentile(ts, 90):.3f} max {ts.max():.3f}; counters {hip.counters()}", flush=True)
