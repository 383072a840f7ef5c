#!/usr/bin/env python3
"""GPU box: BASELINE config 5 (gyroid-sphere Manifold Dual Contouring) - wall time of fidget_amd.mesh per octree depth, with the
phases the library reports (FHIP_MESH_TIMES=1), next to the CPU oracle where it finishes in seconds."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FHIP_MESH_TIMES"] = "1"
import torch
import numpy as np
import fidget_amd as F
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
res = {}
depths = [int(a) for a in sys.argv[1:]] or [6, 7, 8, 9]
shape = F.Shape.from_vm(os.path.join(ROOT, "models", "gyroid-sphere.vm"), hip=hip)
F.mesh(shape, 4)
for depth in depths:
    for rep in range(int(os.environ.get("MESH_TIMES_REPS", "1"))):      # (the first build of a size also pins its landing area and makes room)
        t0 = time.perf_counter()
        tris, verts, counts = F.mesh(shape, depth)
        dt = time.perf_counter() - t0
        print(depth, "build", rep, dt, flush=True)
    res[f"depth {depth}"] = {"s": dt, "triangles": len(tris), "vertices": len(verts), **counts}
    print(depth, res[f"depth {depth}"], flush=True)
    if os.environ.get("MESH_TIMES_PARITY"):
        # the same mesh from the CPU oracle's multithreaded constructor (Octree::build_inner_mt restated, oracle/src/mesh.hpp build_mt) on
        # every host thread + its walk_dual: the CPU baseline of this configuration, and the device mesh compared with it element for element
        import hashlib
        import oracle as O
        oshape = O.Shape.from_vm(os.path.join(ROOT, "models", "gyroid-sphere.vm"))
        th = O.max_threads()
        t0 = time.perf_counter()
        oc = O.Octree(oshape, depth, threads=th, keep_samples=False)
        t_build = time.perf_counter() - t0
        ot, ov = oc.walk_dual()
        t_cpu = time.perf_counter() - t0
        te = bool(ot.shape == tris.shape and (ot == tris).all())
        ve = bool(ov.shape == verts.shape and (ov.view(np.uint32) == verts.view(np.uint32)).all())
        sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
        print(f"parity depth {depth} triangles_equal {te} vertices_equal {ve} cpu_s {t_cpu:.3f} cpu_build_s {t_build:.3f} threads {th} "
              f"sha_device {sha(tris)}{sha(verts)} sha_oracle {sha(ot)}{sha(ov)}", flush=True)
        res[f"depth {depth}"].update(parity={"triangles_equal": te, "vertices_equal": ve}, cpu_s_per_build=t_cpu, cpu_build_s=t_build, cpu_threads=th)
        del oc, ot, ov
        # ... and the vertices against a solve that is NOT the product's own arithmetic (tests/qef_independent.py: the QEFs accumulated
        # from the device's leaf records as qef.rs does, solved by LAPACK's f64 SVD under qef.rs's rank rule), at depth 8 - the leaf
        # records of a depth-10 build are 17 GB; the solve per vertex is the same code at any depth
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import qef_independent as Q
        recs, _ = F.mesh_sample(shape, 8)
        q = Q.check(recs, Q.per_vertex_counts(O.mdc_table))
        print("qef_independent depth 8 " + json.dumps(q), flush=True)
        res[f"depth {depth}"]["qef_independent_check"] = q
        del recs
    del tris, verts
    if os.environ.get("MESH_TIMES_HOST_ASM"):      # the same build with the octree assembled on the host's threads (the round-2 path)
        with hip.options(mesh_device_assembly=0):
            for rep in range(2):
                t0 = time.perf_counter()
                tris, verts, counts = F.mesh(shape, depth)
                dt = time.perf_counter() - t0
                del tris, verts
        res[f"depth {depth}, host assembly"] = {"s": dt}
        print(depth, "host assembly", dt, flush=True)
        t0 = time.perf_counter()
        tris, verts, counts = F.mesh(shape, depth)
        res[f"depth {depth}"]["s_again"] = time.perf_counter() - t0
        print(depth, "device assembly again", res[f"depth {depth}"]["s_again"], flush=True)
        del tris, verts
# the build sharded by the root's octants (fidget_amd.mesh_part / mesh_merge), its parts one after the other on this one GPU:
# what each rank of an N-GPU build would spend on its part, and what the merging rank spends afterwards
n_parts = int(os.environ.get("MESH_TIMES_PARTS", "0"))
if n_parts > 1:
    for depth in depths:
        part_s, blobs = [], []
        for k in range(n_parts):
            t0 = time.perf_counter()
            blobs.append(F.mesh_part(shape, depth, k, n_parts))
            part_s.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        tris, verts, counts = F.mesh_merge(blobs, hip=hip)
        merge_s = time.perf_counter() - t0
        res[f"depth {depth}, {n_parts} parts"] = {"part_s": part_s, "part_bytes": [int(b.size) for b in blobs], "merge_s": merge_s,
                                                "slowest_part_plus_merge_s": max(part_s) + merge_s, "triangles": len(tris), "vertices": len(verts), **counts}
        print(depth, res[f"depth {depth}, {n_parts} parts"], flush=True)
        del tris, verts, blobs
if "--oracle" in os.environ.get("MESH_TIMES_FLAGS", ""):
    import oracle as O
    for depth in [d for d in depths if d <= 8]:
        t0 = time.perf_counter()
        m = O.mesh(os.path.join(ROOT, "models", "gyroid-sphere.vm"), depth)
        res[f"oracle depth {depth}"] = {"s": time.perf_counter() - t0}
        print("oracle", depth, res[f"oracle depth {depth}"], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "mesh_times.json"), "w"), indent=1)
