#!/usr/bin/env python3
"""GPU box: frames of a sequence rendered by K contexts in turn (each context its own streams and frame sets, frame i on context
i % K), against one context.  The question it answers: the pipelined frame rate of one context is set by its busiest stream
(the stage of frame n + 1 cannot start before the same stage of frame n: 0.43 ms of level 1 per frame, profiles/r03x/timeline_one_frame.txt),
while the kernels of those stages are bound by latency, not by the machine (one wave per SIMD, 0.5-4 k waves) - so does a second
context's pipeline fit beside the first's?  usage: tools/two_contexts.py [size] [model] [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fidget_amd as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = sys.argv[2] if len(sys.argv) > 2 else "prospero.vm"
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 120
KS = [int(k) for k in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 2, 3, 4]
for K in KS:
    streams = [torch.cuda.Stream() for _ in range(K)]
    hips = [F.HipContext(0, s.cuda_stream) for s in streams]
    shapes = [F.Shape.from_vm(os.path.join(ROOT, "models", model), hip=h) for h in hips]
    outs = [torch.zeros((n, n, 4), dtype=torch.int32, device="cuda") for _ in range(K)]
    for i in range(3 * K):
        F.render3d(shapes[i % K], n, out=outs[i % K])
    torch.cuda.synchronize()
    ref = outs[0].clone()
    import gc
    gc.collect(); gc.disable()
    t0 = time.perf_counter()
    for i in range(frames):
        F.render3d(shapes[i % K], n, out=outs[i % K])
    for h in hips:
        h.sync()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / frames * 1e3
    gc.enable()
    same = all(bool((o == ref).all()) for o in outs)
    print(f"{K} context(s): {dt:.3f} ms per {n}^3 frame of {model}, images equal: {same}", flush=True)
    del shapes, hips
