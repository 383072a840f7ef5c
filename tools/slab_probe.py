#!/usr/bin/env python3
"""GPU box: pipelined frame time of models that DO have occluded geometry (colonnade.vm under the reference bench camera and the
identity camera, bear.vm) by slab thickness (context option slab_layers): thicker z-slabs shorten the tile chain but let the
tile stage see less of what the slabs in front already hide."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import fidget_amd as F
from test_gpu_parity import bench_camera
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
res = {}
for model, n, cam in (("colonnade.vm", 1024, bench_camera(0.3)), ("colonnade.vm", 1024, None), ("bear.vm", 512, None), ("prospero.vm", 2048, None)):
    shape = F.Shape.from_vm(os.path.join(ROOT, "models", model), hip=hip)
    out = torch.zeros((n, n, 4), dtype=torch.int32, device="cuda")
    for sl in (1, 2, 4):
        hip.set_option("slab_layers", sl)
        for _ in range(3):
            F.render3d(shape, n, out=out, world_to_model=cam)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            F.render3d(shape, n, out=out, world_to_model=cam)
        torch.cuda.synchronize()
        res[f"{model} {n}^3 {'bench camera' if cam is not None else 'identity'} slab_layers={sl}"] = (time.perf_counter() - t0) / 20 * 1e3
        print(list(res.items())[-1], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out", "r03f"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r03f", "slab_probe.json"), "w"), indent=1)
