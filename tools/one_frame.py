#!/usr/bin/env python3
"""GPU box, under rocprofv3 --kernel-trace: frames of prospero.vm 1024^3 rendered one at a time (waited for each) - the timeline of one
frame alone, memory operations included (tools/timeline.py <dir> 1 1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fidget_amd as F
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
shape = F.Shape.from_vm(os.path.join(ROOT, "models", "prospero.vm"), hip=hip)
out = torch.zeros((1024, 1024, 4), dtype=torch.int32, device="cuda")
if len(sys.argv) > 1 and sys.argv[1] == "general":
    hip.set_option("no_column_inv", 1)
for _ in range(8):
    F.render3d(shape, 1024, out=out)
    torch.cuda.synchronize()
