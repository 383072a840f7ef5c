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


def t3(model, n, reps=20, **kw):
    shape = F.Shape.from_vm(os.path.join(ROOT, "models", model), hip=hip)
    out = torch.zeros((n, n, 4), dtype=torch.int32, device="cuda")
    for _ in range(60):      # (the library's arrangement tuner takes ~50 queued frames of a kind, capi_render.hpp lane_mode)
        F.render3d(shape, n, out=out, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        F.render3d(shape, n, out=out, **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def t2(model, n, reps=20):
    shape = F.Shape.from_vm(os.path.join(ROOT, "models", model), hip=hip)
    out = torch.zeros((n, n), dtype=torch.float32, device="cuda")
    for _ in range(8):
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
# C5 (mesh): Octree::build + walk_dual of gyroid-sphere (device: cell recursion, leaf sampling, QEF; host threads: octree assembly with
# cell collapse, dual walk).  Depth 10 = BASELINE.json's 1024^3; the second depth-10 call finds the pinned landing area of the leaf
# records already there
shape = F.Shape.from_vm(os.path.join(ROOT, "models", "gyroid-sphere.vm"), hip=hip)
F.mesh(shape, 4)
for tag, depth in (("", 8), ("", 9), (" (first call)", 10), (" (second call)", 10)):
    t0 = time.perf_counter()
    tris, verts, counts = F.mesh(shape, depth)
    dt = time.perf_counter() - t0
    res[f"C5 gyroid-sphere.vm mesh, octree depth {depth} ({2 ** depth}^3){tag}: s / triangles / vertices / cells evaluated / leaf cells"] = [
        dt, len(tris), len(verts), counts["cells"], counts["leaf_cells"]]
    del tris, verts
print(json.dumps(res, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "other_configs.json"), "w"), indent=1)
