#!/usr/bin/env python3
"""GPU box, under rocprofv3 --kernel-trace --stats: N frames of one of the other BASELINE configurations.
usage: config_profile.py bear3d|prospero2d [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fidget_amd as F
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
what = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10
if what == "bear3d":
    shape = F.Shape.from_vm(os.path.join(ROOT, "models", "bear.vm"), hip=hip)
    out = torch.zeros((512, 512, 4), dtype=torch.int32, device="cuda")
    for _ in range(frames):
        F.render3d(shape, 512, out=out)
else:
    shape = F.Shape.from_vm(os.path.join(ROOT, "models", "prospero.vm"), hip=hip)
    out = torch.zeros((4096, 4096), dtype=torch.float32, device="cuda")
    for _ in range(frames):
        F.render2d(shape, 4096, out=out)
torch.cuda.synchronize()
