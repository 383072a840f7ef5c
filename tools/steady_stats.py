#!/usr/bin/env python3
"""GPU box, under `rocprofv3 --kernel-trace --stats`: N frames of the general path (or a model's default path) rendered ONE AT A TIME in the
library's profiling mode - every kernel between its own pair of events, nothing of another frame beside it, the arena at its final size
from the first frame - so that the profiler's per-kernel AVERAGE is the same quantity as bench.py's `roofline.avg_launch_ms`
(round 5's committed average mixed queued frames, overflowing first frames and profiled ones: 1.27 ms against the line's 0.57).
Prints the HIP-event average per launch of the leaf kernel for comparison.   usage: steady_stats.py [model.vm] [size] [general|default] [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fidget_amd as F
model = sys.argv[1] if len(sys.argv) > 1 else "prospero.vm"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
general = (sys.argv[3] if len(sys.argv) > 3 else "general") == "general"
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 30
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
if general:
    hip.set_option("no_column_inv", 1)
shape = F.Shape.from_vm(os.path.join(ROOT, "models", model), hip=hip)
out = torch.zeros((n, n, 4), dtype=torch.int32, device="cuda")
hip.profile(True)
tot = {}
for i in range(frames + 3):
    F.render3d(shape, n, out=out)
    k = hip.profile_read_kernels()
    if i >= 3:
        for name, (ms, cnt) in k.items():
            if cnt:
                t = tot.setdefault(name, [0.0, 0])
                t[0] += ms; t[1] += cnt
hip.profile(False)
print({name: round(v[0] / v[1], 4) for name, v in tot.items()}, "HIP-event ms per launch over", frames, "frames (the first 3 of the process left out; rocprofv3 counts them)")
