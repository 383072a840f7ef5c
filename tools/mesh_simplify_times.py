#!/usr/bin/env python3
"""GPU box: mesh builds of the large-tape models with and without the tape simplification down the octree (context option
mesh_simplify_min_ops: 256 by default, 0 = every cell and leaf sample evaluated with the root tape, as until round 3): seconds per
build inside the wrapper (best of 3 after a warm-up build), and that both give the same mesh.
usage: tools/mesh_simplify_times.py [depth, default 8]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.pop("FHIP_MESH_TIMES", None)
import numpy as np
import torch
import fidget_amd as F
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
res = {}
for model in ("prospero.vm", "colonnade.vm", "bear.vm"):
    shape = F.Shape.from_vm(os.path.join(ROOT, "models", model), hip=hip)
    out = {}
    for name, min_ops in (("simplified", 256), ("root tape", 0)):
        with hip.options(mesh_simplify_min_ops=min_ops):
            F.mesh(shape, depth)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                tris, verts, counts = F.mesh(shape, depth)
                best = min(best, time.perf_counter() - t0)
        out[name] = {"s_per_build": round(best, 4), "triangles": int(len(tris)), "vertices": int(len(verts)), "cells_evaluated": int(counts["cells"])}
        out[name + " mesh"] = (tris, verts)
    a, b = out.pop("simplified mesh"), out.pop("root tape mesh")
    out["meshes_identical"] = bool(a[0].shape == b[0].shape and (a[0] == b[0]).all() and a[1].shape == b[1].shape and (a[1].view(np.uint32) == b[1].view(np.uint32)).all())
    out["speedup"] = round(out["root tape"]["s_per_build"] / out["simplified"]["s_per_build"], 2)
    res[f"{model} depth {depth}"] = out
    print(model, json.dumps(out), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"mesh_simplify_times_depth{depth}.json"), "w"), indent=1)
