#!/usr/bin/env python3
"""GPU box: the frame as a caller of the reference's blocking API gets it - fhip_render3d with a HOST output pointer returns when
the image is in host memory (nothing pipelined, the device-to-host copy included) - next to bench.py's device-resident numbers.
prospero.vm 1024^3; pageable and pinned landing buffers."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fidget_amd as F

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
shape = F.Shape.from_vm(os.path.join(ROOT, "models", "prospero.vm"), hip=hip)
res = {}
pinned = torch.zeros((n, n, 16), dtype=torch.uint8).pin_memory().numpy().view(F.GEOMETRY_PIXEL).reshape(n, n)
for name, buf in (("pageable", np.zeros((n, n), F.GEOMETRY_PIXEL)), ("pinned", pinned)):
    for _ in range(3):
        F.render3d(shape, n, host_out=buf)
    ms = sorted(F.render3d(shape, n, host_out=buf)[2] * 1e3 for _ in range(20))
    res[name] = {"median_ms": ms[10], "min_ms": ms[0], "Mvoxel_per_s": n ** 3 / (ms[10] * 1e-3) / 1e6}
    print(name, res[name], flush=True)
dev = torch.zeros((n, n, 4), dtype=torch.int32, device="cuda")
F.render3d(shape, n, out=dev)
full = np.asarray(dev.cpu().numpy()).view(np.uint32)
assert np.array_equal(pinned.view(np.uint32).reshape(n, n, 4), full)
res["same_image_as_the_device_resident_render"] = True
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "host_frames.json"), "w"), indent=1)
