#!/usr/bin/env python3
"""GPU box, experiment: frames of a queue shared out between a context under the stage pipeline ("A") and one-stream contexts ("B", "C":
no_pipeline), by a repeating pattern such as AAB - does a lane beside the stage pipeline add to it?
usage: tools/mixed_arrangements.py size model frames PATTERN [PATTERN ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fidget_amd as F
n, model, frames = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
for pattern in sys.argv[4:]:
    names = sorted(set(pattern))
    ctx = {}
    for c in names:
        s = torch.cuda.Stream()
        h = F.HipContext(0, s.cuda_stream)
        h.set_option("frame_lanes", 0)
        if c != "A":
            h.set_option("no_pipeline", 1)
        ctx[c] = (s, h, F.Shape.from_vm(os.path.join(ROOT, "models", model), hip=h), torch.zeros((n, n, 4), dtype=torch.int32, device="cuda"))
    def run(k):
        for i in range(k):
            s, h, shape, out = ctx[pattern[i % len(pattern)]]
            F.render3d(shape, n, out=out)
        for c in names:
            ctx[c][1].sync()
        torch.cuda.synchronize()
    run(3 * len(pattern))
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        run(frames)
        best = min(best, (time.perf_counter() - t0) / frames * 1e3)
    same = all(bool(torch.equal(ctx[c][3], ctx[names[0]][3])) for c in names)
    print(f"{model} {n}^3 pattern {pattern}: {best:.3f} ms per frame, images equal: {same}", flush=True)
    del ctx
