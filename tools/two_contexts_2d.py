#!/usr/bin/env python3
"""GPU box: queued 2D frames rendered by K contexts in turn (frame i on context i % K), against one context - the measurement behind the
frame lanes for 2D frames (profiles/r04r/frame_major3.txt).  usage: tools/two_contexts_2d.py model size frames K,K,...
(the images are compared bit for bit; a fill pixel is a NaN pattern, which torch.equal would call unequal to itself)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fidget_amd as F
model, n, frames = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
for K in [int(k) for k in sys.argv[4].split(",")]:
    streams = [torch.cuda.Stream() for _ in range(K)]
    hips = [F.HipContext(0, s.cuda_stream) for s in streams]
    shapes = [F.Shape.from_vm(os.path.join(ROOT, "models", model), hip=h) for h in hips]
    outs = [torch.zeros((n, n), dtype=torch.float32, device="cuda") for _ in range(K)]
    for i in range(3 * K):
        F.render2d(shapes[i % K], n, out=outs[i % K])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(frames):
        F.render2d(shapes[i % K], n, out=outs[i % K])
    for h in hips:
        h.sync()
    torch.cuda.synchronize()
    same = all(bool(torch.equal(o.view(torch.int32), outs[0].view(torch.int32))) for o in outs)
    print(f"2D {model} {n}^2, {K} context(s): {(time.perf_counter() - t0) / frames * 1e3:.3f} ms per frame, images equal: {same}", flush=True)
    del shapes, hips
