#!/usr/bin/env python3
"""GPU box: frame times of the other BASELINE.json configurations (parity cases, not bench lines)."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import numpy as np
import fidget_amd as F
import oracle as O
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
res = {}


def t3(model, n, reps=10, **kw):
    shape = F.Shape.from_vm(os.path.join(ROOT, "models", model), hip=hip)
    out = torch.zeros((n, n, 4), dtype=torch.int32, device="cuda")
    for _ in range(3):
        F.render3d(shape, n, out=out, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        F.render3d(shape, n, out=out, **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def t2(model, n, reps=10):
    shape = F.Shape.from_vm(os.path.join(ROOT, "models", model), hip=hip)
    out = torch.zeros((n, n), dtype=torch.float32, device="cuda")
    for _ in range(3):
        F.render2d(shape, n, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        F.render2d(shape, n, out=out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


res["C2 prospero.vm 2D 4096^2 ms"] = t2("prospero.vm", 4096)
res["C3 bear.vm 3D 512^3 ms"] = t3("bear.vm", 512)
res["prospero.vm 3D 2048^3 ms"] = t3("prospero.vm", 2048, reps=5)
sys.path.insert(0, os.path.join(ROOT, "tests"))
try:
    from test_gpu_parity import bench_camera
    res["colonnade.vm 3D 1024^3, reference bench camera (perspective 0.3) ms"] = t3("colonnade.vm", 1024, world_to_model=bench_camera(0.3))
    res["colonnade.vm 3D 1024^3, identity ms"] = t3("colonnade.vm", 1024)
except Exception as e:
    res["bench camera"] = repr(e)
# C5 (mesh): the evaluation side of Octree::build for gyroid-sphere (leaf sampling; no dual walk on the device yet)
for depth in (6, 7, 8):
    shape = F.Shape.from_vm(os.path.join(ROOT, "models", "gyroid-sphere.vm"), hip=hip)
    F.mesh_sample(shape, 4)
    t0 = time.perf_counter()
    leaves, counts = F.mesh_sample(shape, depth)
    dt = (time.perf_counter() - t0) * 1e3
    res[f"C5 gyroid-sphere.vm octree depth {depth} ({2 ** depth}^3): cells / leaf cells / ms (incl. {leaves.nbytes >> 20} MiB of leaf records to the host)"] = [counts["cells"], counts["leaf_cells"], dt]
print(json.dumps(res, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "other_configs.json"), "w"), indent=1)
