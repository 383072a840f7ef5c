#!/usr/bin/env python3
"""GPU box (run by hand, not a test): throughput with N frames in flight (one fhip context + stream per frame slot)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fidget_amd as F
n = 1024
res = {}
for inflight in (1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(inflight)]
    ctxs = [F.HipContext(0, s.cuda_stream) for s in streams]
    shapes = [F.Shape.from_vm(os.path.join(ROOT, "models", "prospero.vm"), hip=c) for c in ctxs]
    outs = [torch.zeros((n, n, 4), dtype=torch.int32, device="cuda") for _ in ctxs]
    for i in range(2 * inflight):
        F.render3d(shapes[i % inflight], n, out=outs[i % inflight])
    torch.cuda.synchronize()
    K = 30
    t0 = time.perf_counter()
    for i in range(K):
        F.render3d(shapes[i % inflight], n, out=outs[i % inflight])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    same = all(bool((outs[0] == o).all()) for o in outs)
    res[inflight] = {"ms_per_frame": dt, "identical_images": same}
    print(inflight, dt, same, flush=True)
    del ctxs, shapes, outs
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "inflight.json"), "w"))
