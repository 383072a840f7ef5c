#!/usr/bin/env python3
"""BASELINE configuration 5 on the GPUs of one node: gyroid-sphere Manifold Dual Contouring with the build sharded by the
root's octants (fidget_amd.dist.mesh_sharded: one process per GPU, fhip_mesh_sample_part on each, the parts handed to rank 0
through shared memory, fhip_mesh_merge there).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/mesh_sharded.py [depth] [repeats]

(also runs as one plain process: N = 1).  Rank 0 prints one JSON line: wall time per build (max over ranks, barrier on both
sides), the mesh's size, and whether it equals the single-GPU build of the same model on rank 0 when CHECK=1."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
import fidget_amd as F
from fidget_amd.dist import mesh_sharded

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 10
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
hip = F.HipContext(local, torch.cuda.current_stream(dev).cuda_stream)
shape = F.Shape.from_vm(os.path.join(ROOT, "models", "gyroid-sphere.vm"), hip=hip)


def fence():
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)


def build():
    return mesh_sharded(lambda r, n, alloc: F.mesh_part(shape, depth, r, n, alloc=alloc), lambda parts: F.mesh_merge(parts, hip=hip), dst=0, device=dev)


build()                                   # warm-up: pinned landing areas, host-side caches
times, result = [], None
for _ in range(repeats):
    fence()
    t0 = time.perf_counter()
    result = build()
    fence()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    times.append(float(dt.item()))
if rank == 0:
    tris, verts, counts = result
    out = {"model": "gyroid-sphere.vm", "depth": depth, "n_gpus": world, "parts": min(world, 8), "seconds": sorted(times), "median_s": float(np.median(times)),
           "triangles": int(len(tris)), "vertices": int(len(verts)), **counts}
    if os.environ.get("CHECK"):
        t1, v1, c1 = F.mesh(shape, depth)
        out["equals_single_gpu_build"] = bool(np.array_equal(t1, tris) and np.array_equal(v1.view(np.uint32), verts.view(np.uint32)) and c1 == counts)
    print(json.dumps(out), flush=True)
if world > 1:
    dist.destroy_process_group()
