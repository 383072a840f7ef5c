#!/usr/bin/env python3
"""bench.py - headline benchmark of fidget-hip on MI355X.

Metric (BASELINE.json): Mvoxel/s of the heightmap+normals render (interval + point
evaluation) of prospero.vm at 1024^3, nominal volume / wall time
(W*H*D / t / 1e6, the convention of the reference's README.md:152-156).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (the driver launches N>1 through torch.distributed.run).  A step is one full frame,
inputs resident in HBM, output left in HBM; frames are queued back to back and the library pipelines them
(the coarse levels of frame n + 1 beside the slabs of frame n).

What the ONE JSON line says, and how to read it:

  value / ms_per_step        the timed K frames of the DEFAULT path.  prospero.vm has no z: every one of its tapes takes
                             the column-invariance short cuts, so next to it stand
  general                    the same K frames with the short cuts off (context option no_column_inv): what a model with z
                             in every tape gets from the same kernels; its image is compared with the default path's;
  frame_latency_ms           one frame alone, nothing in flight before it (the reference's -N loop is blocking frames),
                             for both paths; host_output_frame_ms: the blocking call with a HOST output buffer, device to
                             host copy included (what the Rust trait's render returns; never `value`);
  roofline                   PRIMARY: the dominant kernel of the GENERAL path (the leaf interpreter fh_columns), numerator
                             from the device's own counters of the very frames whose launches are timed (leaf tape words x
                             passes, counted where the leaves are queued), every launch between its own pair of HIP events
                             on the stream it is launched on; `alu` beside the HBM figure: f32 lane-operations per second
                             against the plain-f32 VALU rate - the resource that actually binds an interpreter;
                             `path_frame` inside it: that path's ms_per_step / value / single-frame latency;
  roofline_timed_path        the dominant kernel of the path `value` times (prospero.vm: level 1's fh_tiles_v64), same fields;
  roofline_default_path      the leaf kernel on the default path; roofline_tiles / roofline_tiles_l1 / roofline_prune (and their
                             *_default_path twins): the per-slab tile kernel fh_tiles_v32, level 1's fh_tiles_v64 and the root
                             level's prune likewise (tape ops read + written at their level in the timed frames);
  cpu_baseline               the C++ oracle (restatement of the reference's VmShape path, OpenMP over root tiles like
                             render_tiles' rayon pool) on this box's host cores, same frame; parity of the device image
                             against it at full size; `c3_bear`: BASELINE configuration 3 with its measured normal error; `c5_mesh`:
                             BASELINE configuration 5, seconds per mesh build.

N > 1: ONE frame sharded over the ranks (total work fixed: "strong") - "columns" (root-tile column index % N == rank at
full depth; one RCCL SUM reduce of the partial images) and "blocks" (the north star's octants, 2 x 2 x 2 at N = 8: a
gather of the ranks' rectangles, then the front-to-back depth merge on rank 0).  `value` is the better of these two.  A
third sharding, whole frames of the SEQUENCE per rank ("frames": per-GPU work fixed, i.e. weak scaling), is timed and
listed under `partitions` only - it is not the north star's number.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_LANE_OPS = 78.6e12  # plain f32 VALU lane-operations per second: 256 CUs x 4 SIMDs x 32 lanes per clock x 2.4 GHz (= the 157.3 TFLOP/s vector peak / 2 flops per FMA)
ISSUE_PEAK = 0.57              # wave-instructions per cycle per SIMD, measured (see `issue.peak_note`)
TRAFFIC_FILE = os.path.join("profiles", "traffic_r06.json")


MESH_LEAF_BYTES = 528       # sizeof(FhMeshLeaf) = fidget_amd.MESH_LEAF.itemsize (tests/test_bench_protocol.py holds the two together)


def parse_mesh_times(stdout, stderr):
    """What tools/mesh_times.py 10 (MESH_TIMES_REPS >= 2) printed -> the `c5_mesh` object, or None if it did not get through its builds"""
    import re
    builds = [float(m.group(1)) for m in re.finditer(r"^10 build \d+ ([0-9.]+)$", stdout, re.M)]
    inside = [float(m.group(1)) for m in re.finditer(r"^fhip mesh depth 10: .* total ([0-9.]+) s$", stderr, re.M)]
    counts = re.search(r"'triangles': (\d+), 'vertices': (\d+)", stdout)
    if len(builds) < 2:
        return None
    c5 = {"workload": "gyroid-sphere.vm Manifold Dual Contouring, octree depth 10 (1024^3)", "s_per_build": min(builds[1:]),
          "s_per_build_inside_the_library": min(inside[1:]) if len(inside) >= 2 else None, "s_first_build": builds[0],
          "triangles": int(counts.group(1)) if counts else None, "vertices": int(counts.group(2)) if counts else None,
          "note": "wall time of fidget_amd.mesh (fhip_mesh_build + copying triangles and vertices out), best of the builds after the first of this size"}
    # the leaf stage (corners, edge search through the bulk interpreter, gradients, QEF vertices): its leaf records are what the build moves
    ls = [(float(m.group(1)), int(m.group(2))) for m in re.finditer(r"^fhip mesh depth 10: .*leaf kernel ([0-9.]+) s \((\d+) leaves\)", stderr, re.M)]
    if len(ls) >= 2:
        sec, leaves = min(ls[1:])
        c5["roofline"] = {"bound": "hbm", "kernel": "leaf stage (k_mesh_leaf passes + fh_float_eval edge search)", "achieved": leaves * MESH_LEAF_BYTES / sec / 1e9, "peak": 8000.0,
                          "unit": "GB/s", "frac": leaves * MESH_LEAF_BYTES / sec / 1e9 / 8000.0, "leaf_stage_s": sec, "leaves": leaves,
                          "what": f"{MESH_LEAF_BYTES} B of leaf record written per ambiguous leaf cell over the leaf stage's time (the records go to pinned host memory in chunks while it runs)"}
    # MESH_TIMES_PARITY: the CPU oracle's multithreaded build (Octree::build_inner_mt restated) + walk_dual at the same depth, timed, and the
    # device mesh compared with it element for element
    par = re.search(r"^parity depth 10 triangles_equal (True|False) vertices_equal (True|False) cpu_s ([0-9.]+) cpu_build_s ([0-9.]+) threads (\d+) "
                    r"sha_device (\w+) sha_oracle (\w+)$", stdout, re.M)
    if par:
        c5["parity"] = {"triangles_equal": par.group(1) == "True", "vertices_equal": par.group(2) == "True", "sha_device": par.group(6), "sha_oracle": par.group(7),
                        "against": "oracle build_mt + walk_dual at depth 10, element for element"}
        c5["cpu_s_per_build"], c5["cpu_build_s"], c5["cpu_threads"] = float(par.group(3)), float(par.group(4)), int(par.group(5))
    # ... and the distance of the device's vertices from a solve that is not the product's own (LAPACK f64 SVD of the QEFs accumulated from
    # the device's leaf records, qef.rs's rank rule; tests/qef_independent.py): the oracle shares the product's Jacobi solve, this does not
    q = re.search(r"^qef_independent depth (\d+) (\{.*\})$", stdout, re.M)
    if q and c5.get("parity") is not None:
        qq = json.loads(q.group(2))
        c5["parity"]["qef_vs_independent_f64_svd"] = {"depth": int(q.group(1)), "vertices": qq["vertices"], "max_deviation_cell_fraction": qq["max_deviation_cell_fraction"],
                                                      "over_1e-4_of_a_cell": qq["over_1e-4_of_a_cell"], "rank_decisions_within_1e-4_of_the_cutoff": qq["rank_decisions_within_1e-4_of_the_cutoff"]}
    return c5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--model", default="prospero.vm")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline / parity / configuration-3 leg")
    ap.add_argument("--no-general", action="store_true", help="skip the frames with the column-invariance short cuts off (profiling runs: per-kernel statistics of the default path only)")
    ap.add_argument("--only-general", action="store_true", help="profiling runs: time only the frames with the short cuts off")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import fidget_amd as F
    from fidget_amd.dist import combine, gather_blocks, gather_frames, block_split

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run (WORLD_SIZE=1 here)", file=sys.stderr)
            sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    direct_note = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        # The frame's one collective goes through RCCL's C API ON THE RENDER'S STREAM (fidget_amd/dist.py DirectRccl):
        # torch.distributed would run it on a stream of its own behind a cross-stream wait, and such a wait holds the next
        # frame's coarse levels back (one GPU, stand-in tools/fifth_stream.py: the pipelined frame rate halves).  The process
        # group stays for the barrier, the unique id and the timing reduction - and as the fallback if the library cannot be
        # loaded on some rank (agreed on before any rank enters the communicator's collective initialisation).
        from fidget_amd.dist import DirectRccl, use_direct_rccl
        lib_ok = 1
        if os.environ.get("FHIP_NO_DIRECT_RCCL"):
            lib_ok = 0
        else:
            try:
                DirectRccl.probe()
            except Exception as e:      # noqa: BLE001
                lib_ok, direct_note = 0, f"librccl not usable through ctypes ({e!r}): torch.distributed collectives"
        flag = torch.tensor([lib_ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            def bcast(raw, src):
                t = torch.tensor(list(raw), dtype=torch.uint8, device=dev) if rank == src else torch.zeros(len(raw), dtype=torch.uint8, device=dev)
                dist.broadcast(t, src=src)
                return bytes(t.cpu().tolist())
            # one trial of both collectives, checked, before the timed loops rely on them
            try:
                comm = DirectRccl(rank, world, bcast)
                raw = torch.cuda.current_stream(dev).cuda_stream
                t = torch.full((4096,), rank + 1, dtype=torch.int32, device=dev)
                comm.reduce_sum(t, 0, raw)
                send = torch.full((1024, 4), 7 * rank + 3, dtype=torch.int32, device=dev)
                recv = torch.zeros((world, 1024, 4), dtype=torch.int32, device=dev) if rank == 0 else None
                comm.gather(send, recv, 0, raw)
                torch.cuda.synchronize(dev)
                trial_ok = 1
                if rank == 0:
                    want = torch.arange(world, dtype=torch.int32, device=dev) * 7 + 3
                    trial_ok = int(bool((t == world * (world + 1) // 2).all()) and bool((recv == want[:, None, None]).all()))
            except Exception:       # noqa: BLE001
                trial_ok = 0
            flag = torch.tensor([trial_ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                use_direct_rccl(comm)
                direct_note = "RCCL C API on the render's stream (fidget_amd.dist.DirectRccl)"
            else:
                direct_note = "torch.distributed collectives (the trial of the direct RCCL communicator failed)"
        elif direct_note is None:
            direct_note = "torch.distributed collectives (FHIP_NO_DIRECT_RCCL or another rank could not load librccl)"

    n = args.size
    stream = torch.cuda.current_stream(dev)
    mem_info = getattr(torch.cuda, "mem_get_info", None)
    try:        # (device memory the library takes: hipMalloc'd by the context, not by torch)
        free_before = mem_info(dev)[0]
    except Exception:       # noqa: BLE001
        free_before = None
    hip = F.HipContext(local, stream.cuda_stream)
    shape = F.Shape.from_vm(os.path.join(ROOT, "models", args.model), hip=hip)
    out = torch.zeros((n, n, 4), dtype=torch.int32, device=dev)  # GeometryPixel = 4 x 32-bit words

    split = block_split(world)

    def step_columns():
        F.render3d(shape, n, out=out, shard=rank, n_shards=world)
        combine(out, dst=0)  # one RCCL reduce of the partial images (no-op at N = 1)

    def step_blocks():
        F.render3d(shape, n, out=out, block=(rank, split))
        gather_blocks(out, n, split, lambda a, b, d: F.merge_depth(a, b, d, hip=hip), dst=0)

    frames_recv = None

    def step_frames():
        # the frame SEQUENCE sharded by frame: every rank renders a whole frame of its own (frame r of each group of `world`
        # consecutive frames), rank 0 collects the finished frames; no collective inside a frame, no merge rule
        F.render3d(shape, n, out=out)
        gather_frames(out, frames_recv, dst=0)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    frame_ms = []
    tuning = {"frames": 0, "last": None, "per_kind": []}

    def timed(step):
        import gc
        for _ in range(args.warmup):
            step()
        # (the library times the first ~50 queued frames of a kind under its two frame arrangements - stage pipeline, frame lanes - and keeps
        # the faster, capi_render.hpp lane_mode: those frames are warm-up too, counted in config.tuning_frames)
        mine = 0
        while world == 1 and mine < TUNING_CAP and 0 <= hip.lane_tune()["phase"] < 4:       # (per kind of frame: an option change starts the tuner again)
            step()
            mine += 1
        tuning["frames"] += mine
        tuning["per_kind"].append(mine)
        if world > 1:       # (every rank the same number of steps - there are collectives in them -, so a fixed count here: what the tuner takes, and a few)
            for _ in range(60):
                step()
            tuning["frames"] += 60
        tuning["last"] = hip.lane_tune()
        fence()
        # (the host only queues work here: a collection of the interpreter's garbage in the middle of the loop - 20 ms with torch and
        # numpy loaded - drains the three frames the pipeline holds and shows up as one frame of 10 x the median)
        gc.collect()
        gc.disable()
        try:
            return _timed(step)
        finally:
            gc.enable()

    def _timed(step):
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t0 = time.perf_counter()
        marks[0].record(stream)
        for i in range(args.steps):
            step()
            marks[i + 1].record(stream)      # (an event record costs ~1 us on the stream; the frames still queue back to back)
        fence()
        dt = time.perf_counter() - t0
        frame_ms[:] = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def latency(step, reps=7):
        # one frame alone: nothing in flight before it, waited for
        lat = []
        for _ in range(reps):
            fence()
            t0 = time.perf_counter()
            step()
            torch.cuda.synchronize(dev)
            lat.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(lat))

    partitions = {}
    bytes_default = None
    general = None
    host_frame = None
    lat_general = None
    if world > 1:
        # the stream of the context is torch's current stream: collectives and renders are ordered on it
        dt_b = timed(step_blocks)
        img_b, ms_b = out.clone(), list(frame_ms)
        lat_b = latency(step_blocks)
        dt_c = timed(step_columns)
        ms_c = list(frame_ms)
        lat_c = latency(step_columns)
        partitions = {"columns": {"ms_per_step": dt_c / args.steps * 1e3, "frame_latency_ms": lat_c, "scaling": "strong",
                                  "combine": "1 RCCL reduce (SUM) of the full image"},
                      "blocks": {"ms_per_step": dt_b / args.steps * 1e3, "frame_latency_ms": lat_b, "scaling": "strong", "split": list(split),
                                 "combine": "RCCL gather of each rank's rectangle + front-to-back depth merge on rank 0"}}
        if rank == 0:
            partitions["images_equal"] = bool(torch.equal(img_b, out))
            frames_recv = torch.zeros((world, n, n, 4), dtype=torch.int32, device=dev)
        img_c = out.clone()
        dt_f = timed(step_frames)        # one step = `world` frames
        partitions["frames"] = {"ms_per_step": dt_f / args.steps * 1e3, "frames_per_step": world, "scaling": "weak",
                                "combine": "none inside a frame: every rank renders whole frames of the sequence; RCCL gather of the finished frames (16 MiB each) to rank 0",
                                "note": "per-GPU work fixed, total work grows with N: listed for comparison, never `value`"}
        if rank == 0:
            partitions["frames"]["images_equal"] = bool((frames_recv == img_c[None]).all())
        # ... and what DOES shard: the same frame with the column-invariance short cuts off - every tile of every slab has
        # a tape of its own, the leaf stage is 1.0 of its 1.5 ms - and a model whose tapes read z, both by blocks; rank 0's merge of frame k is
        # queued behind its own block of frame k and before frame k + 1, the other ranks are a frame ahead by then
        with hip.options(no_column_inv=1):
            dt_bg = timed(step_blocks)
            lat_bg = latency(step_blocks)
            hip.sync()
        partitions["blocks_general_path"] = {"ms_per_step": dt_bg / args.steps * 1e3, "frame_latency_ms": lat_bg, "scaling": "strong", "split": list(split),
                                             "what": "prospero.vm, column-invariance short cuts off (what a model with z in every tape gets), octant blocks"}
        zm = os.path.join(ROOT, "models", "colonnade.vm")
        if os.path.exists(zm) and args.model == "prospero.vm":
            zshape = F.Shape.from_vm(zm, hip=hip)

            def step_blocks_z():
                F.render3d(zshape, n, out=out, block=(rank, split))
                gather_blocks(out, n, split, lambda a, b, d: F.merge_depth(a, b, d, hip=hip), dst=0)
            dt_z = timed(step_blocks_z)
            partitions["blocks_colonnade"] = {"ms_per_step": dt_z / args.steps * 1e3, "frame_latency_ms": latency(step_blocks_z), "scaling": "strong", "split": list(split),
                                              "what": "colonnade.vm 1024^3 (reads z), octant blocks"}
        step_blocks()
        for k, f in (("columns", 1), ("blocks", 1), ("frames", world), ("blocks_general_path", 1), ("blocks_colonnade", 1)):
            if k in partitions:
                partitions[k]["value"] = (n ** 3) * f / (partitions[k]["ms_per_step"] * 1e-3) / 1e6
        # `value`: ONE frame sharded over the ranks - the north star's number - by the better of the two partitions
        if dt_c <= dt_b:
            dt, step, sharding, lat_default = dt_c, step_columns, "root-tile columns round-robin, 1 RCCL reduce", lat_c
            frame_ms[:] = ms_c
        else:
            dt, step, sharding, lat_default = dt_b, step_blocks, f"blocks {split[0]}x{split[1]}x{split[2]} (octant split), RCCL gather + depth merge", lat_b
            frame_ms[:] = ms_b
    else:
        step = step_columns
        sharding = "single GPU"
        if args.only_general:
            hip.set_option("no_column_inv", 1)
        dt = timed(step)
        lat_default = latency(step)
        img_default = out.clone()
        if free_before is not None:       # what the context holds for the path `value` times (before the general path's frames grow the arena)
            try:
                hip.sync()
                bytes_default = int(free_before - mem_info(dev)[0]) - img_default.numel() * 4
            except Exception:       # noqa: BLE001
                bytes_default = None
        # the same frames with the column-invariance short cuts off: leaves evaluated once per voxel, every tile of a z-stack
        # evaluated - prospero.vm has no z, so every one of its tapes takes the short cuts
        if not args.no_general and not args.only_general:
            saved = list(frame_ms)
            with hip.options(no_column_inv=1):
                dt_g = timed(step)
                lat_general = latency(step)
                hip.sync()
            general = {"ms_per_step": dt_g / args.steps * 1e3, "ms_per_step_median": float(np.median(frame_ms)), "value": (n ** 3) * args.steps / dt_g / 1e6,
                       "frame_latency_ms": lat_general, "image_equals_default_path": bool(torch.equal(out, img_default)),
                       "note": "context option no_column_inv = 1: no tape treated as independent of z"}
            frame_ms[:] = saved
            for _ in range(2):
                step()
        # the blocking call with a host output buffer (pinned): what a caller of the reference's blocking API gets
        pinned = torch.zeros((n, n, 16), dtype=torch.uint8).pin_memory().numpy().view(F.GEOMETRY_PIXEL).reshape(n, n)
        for _ in range(2):
            F.render3d(shape, n, host_out=pinned)
        host_frame = float(np.median([F.render3d(shape, n, host_out=pinned)[2] * 1e3 for _ in range(9)]))
    hip.sync()
    counters = hip.counters()
    device_bytes = None
    if free_before is not None:
        try:
            device_bytes = int(free_before - mem_info(dev)[0])
        except Exception:       # noqa: BLE001
            device_bytes = None

    # ---- per-kernel timing with HIP events on the render stream (separate, profiled frames) ----
    # Every launch of an assembly kernel sits between its own pair of events; the same frames fill the device's op counters
    # (tile levels: tape ops read / written; leaf stage: tape words x passes, lane operations), so numerator and denominator
    # of every roofline figure below come from the same frames.
    PROF_FRAMES = 3

    def profiled(no_inv):
        prof = {"tiles": [0.0, 0], "points": [0.0, 0], "normals": [0.0, 0], "other": [0.0, 0]}
        kern, leaf, tiles = {}, None, None
        with hip.options(no_column_inv=1 if no_inv else hip.option("no_column_inv")):
            hip.profile(True)
            for _ in range(PROF_FRAMES):
                F.render3d(shape, n, out=out, shard=rank, n_shards=world)
                for k, (ms, cnt) in hip.profile_read().items():
                    prof[k][0] += ms
                    prof[k][1] += cnt
                for k, (ms, cnt) in hip.profile_read_kernels().items():
                    kern.setdefault(k, [0.0, 0])
                    kern[k][0] += ms
                    kern[k][1] += cnt
                hip.wave_stats()
                tiles, leaf = hip.tile_phases, hip.leaf_stats()     # (per frame: the counters are reset by every render)
            hip.profile(False)
        return {"prof": prof, "kern": kern, "leaf": leaf, "tiles": tiles}

    prof_default = profiled(False) if not args.only_general else None
    prof_general = profiled(True) if (world == 1 and not args.no_general) else None
    F.render3d(shape, n, out=out, shard=rank, n_shards=world)       # (leaves the timed path's image in `out`)
    if world > 1:
        step()  # leave `out` holding the combined image on rank 0
    fence()

    # N > 1: what each rank's share of a frame costs, stage by stage (sums of kernel times over the rank's profiled frames, columns
    # sharding): the coarse chain - root level forward + prune, level 1 - is as LONG on every rank as on one GPU (each parent / child is
    # one wave whatever their number), the per-slab tile stage, the leaves and the normals shrink with the rank's share; a scaling curve
    # taken with this line can be read against these
    per_rank = None
    if world > 1:
        k, pr = prof_default["kern"], prof_default["prof"]
        ms = lambda d, key: (d.get(key, (0.0, 0))[0]) / PROF_FRAMES
        coarse = sum(ms(k, x) for x in ("fh_tiles", "fh_prune1", "fh_tiles_v64"))
        mine = {"rank": rank, "coarse_chain_ms": coarse, "slab_ms": max(ms(pr, "tiles") - coarse, 0.0) + ms(pr, "points") + ms(pr, "normals"),
                "tile_stage_ms": ms(pr, "tiles"), "leaf_ms": ms(pr, "points"), "normals_ms": ms(pr, "normals"), "other_ms": ms(pr, "other")}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = dt / args.steps * 1e3
    value = (n ** 3) * args.steps / dt / 1e6
    pmain = prof_default or prof_general
    result = {
        "metric": "Mvoxel/s (interval+point eval) on prospero.vm 1024^3",
        "value": value, "unit": "Mvoxel/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "ms_per_step_median": float(np.median(frame_ms)), "ms_per_step_min": float(np.min(frame_ms)),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "tuning_frames": tuning["frames"],
        "frame_latency_ms": lat_default,
        "device_bytes": device_bytes, "device_bytes_timed_path": bytes_default,
        "frame_arrangement": {"untimed_frames_beyond_warmup": tuning["frames"], "per_kind": tuning["per_kind"], "last_measured": tuning["last"],
                              "note": "the library measures a run of queued frames of one kind under its stage pipeline and on its frame lanes and keeps the faster "
                                      "(DESIGN.md section 4); bench.py lets that finish before the timed frames"},
        "general": general,
        "host_output_frame_ms": host_frame,
        "config": {"workload": f"{args.model} 3D heightmap+normals {n}^3, HipShape render hints (the library's tiles: 128/32/8, or 32/8 when the root level has few tiles), world_to_model=I",
                   "sharding": sharding,
                   "general_path": None if not general else {"ms_per_step": general["ms_per_step"], "value": general["value"],
                                                             "frame_latency_ms": general["frame_latency_ms"]},
                   "frames": "queued back to back on one stream, as a caller rendering a sequence would; the library pipelines them (four buffer "
                             "sets per context: the coarse levels of a frame run beside the previous frames' slabs), every frame does all of its "
                             "work; frame_latency_ms is one frame alone, host_output_frame_ms the blocking call with a host buffer",
                   "column_invariance": ("off for every number of this line (--only-general)" if args.only_general else
                                         "prospero.vm reads no z, so `value` evaluates each tape once per pixel column (DESIGN.md section 2); "
                                         "general_path = the same frames with that short cut off, what a model with z gets")},
        "kernel_ms_per_frame": {k: v[0] / PROF_FRAMES for k, v in pmain["prof"].items()},
        "kernel_launches_per_frame": {k: v[1] // PROF_FRAMES for k, v in pmain["prof"].items()},
        "asm_kernel_ms_per_frame": {k: v[0] / PROF_FRAMES for k, v in pmain["kern"].items() if v[1]},
        "arena_ops_last_slab": counters["arena_ops"], "arena_overflow": counters["arena_overflow"],
    }
    if prof_general and prof_default:
        result["general"]["asm_kernel_ms_per_frame"] = {k: v[0] / PROF_FRAMES for k, v in prof_general["kern"].items() if v[1]}
    if partitions:
        result["partitions"] = partitions
        result["collectives"] = direct_note
        result["per_rank"] = per_rank
        result["predicted"] = predict_from_stages(per_rank, n, world, partitions)

    # ---- roofline: numerators from the device counters of the profiled frames themselves ----------------
    # HBM traffic per launch: rocprofv3 PMC passes of tools/profile_round.sh, committed under profiles/ (counters cannot be
    # collected inside the timed run); the file names the hash of the device sources it was measured on and is ignored
    # (traffic = null) when that is not this build.  FETCH_SIZE doubled per MI355X_MICROARCH.md.
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from src_hash import source_hash
    traffic, traffic_note = {}, None
    tpath = os.path.join(ROOT, TRAFFIC_FILE)
    if os.path.exists(tpath):
        traffic = json.load(open(tpath))
        if traffic.get("source_hash") != source_hash():
            traffic_note = f"{TRAFFIC_FILE} was measured on sources {traffic.get('source_hash')}, this build is {source_hash()}: traffic not reported"
            traffic = {}
    else:
        traffic_note = f"{TRAFFIC_FILE} not present: traffic not reported"

    def roof(P, path, kernel, alg_bytes_per_frame, lane_ops_per_frame, note):
        ms, launches = P["kern"].get(kernel, (0.0, 0))
        if not launches:
            return None
        per_frame_ms, launches_pf = ms / PROF_FRAMES, launches // PROF_FRAMES
        achieved = alg_bytes_per_frame / (per_frame_ms * 1e-3) / 1e9
        # (the per-kernel profile times the root level's prune - k_prune2 and the scalar sweep behind it - as one slot, "fh_prune1": its counters
        # are the two kernels' together, per pair of launches)
        tp = traffic.get(path) or {}
        parts = [tp[k] for k in (("k_prune2", "fh_prune1") if kernel == "fh_prune1" else (kernel,)) if k in tp]
        t = None
        if parts:
            t = {"launches_per_frame": max(q["launches_per_frame"] for q in parts)}
            for key in ("fetch_kb_per_frame", "write_kb_per_frame", "valu_per_launch", "salu_per_launch"):
                if all(key in q for q in parts):
                    t[key] = sum(q[key] for q in parts)
        tb = None
        if t and "fetch_kb_per_frame" in t and "write_kb_per_frame" in t:
            tb = (2.0 * t["fetch_kb_per_frame"] + t["write_kb_per_frame"]) * 1024.0 / max(t["launches_per_frame"], 1)
        r = {"bound": "hbm", "kernel": "k_prune2 (+ fh_prune1 behind it)" if kernel == "fh_prune1" else kernel, "path": path, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": achieved / HBM_PEAK_GBS, "traffic": tb,
             "algorithmic_bytes_per_launch": alg_bytes_per_frame / max(launches_pf, 1), "avg_launch_ms": per_frame_ms / max(launches_pf, 1),
             "launches_per_frame": launches_pf, "note": note}
        if lane_ops_per_frame:
            a = lane_ops_per_frame / (per_frame_ms * 1e-3)
            r["alu"] = {"bound": "valu", "achieved": a / 1e12, "peak": VALU_PEAK_LANE_OPS / 1e12, "unit": "T lane-ops/s", "frac": a / VALU_PEAK_LANE_OPS,
                        "lane_ops_per_launch": lane_ops_per_frame / max(launches_pf, 1),
                        "note": "useful f32 lane-operations (tape ops x lanes evaluating them; an interval op counts once per child) against the "
                                "plain-f32 VALU rate: what binds an interpreter - it issues ~20-40 instructions per useful op (DESIGN.md section 6)"}
        if t and "valu_per_launch" in t and per_frame_ms > 0:
            cycles = per_frame_ms / max(launches_pf, 1) * 1e-3 * t.get("clock_hz", 2.4e9)
            ipc = (t["valu_per_launch"] + t["salu_per_launch"]) / (cycles * 1024)
            r["issue"] = {"bound": "instruction issue", "achieved": ipc, "peak": ISSUE_PEAK, "unit": "wave-instructions / cycle / SIMD",
                          "frac": ipc / ISSUE_PEAK, "valu_per_launch": t["valu_per_launch"], "salu_per_launch": t["salu_per_launch"],
                          "peak_note": "measured, not a data-sheet figure: independent v_mov_b32 in 4 waves per SIMD retire one instruction per 7.05 "
                                       "cycles per wave (profiles/r02/ubench.json test 0, 4096 waves) = 0.57 per cycle per SIMD; a lone wave gets "
                                       "one per 5.25 cycles = 0.19"}
        if traffic_note:
            r["traffic_note"] = traffic_note
        return r

    def roofs(P, path):
        leaf, tiles = P["leaf"], P["tiles"]
        kname = "fh_columns"
        r_leaf = roof(P, path, kname, 8.0 * leaf["tape_words_read"] + n * n * 16, leaf["lane_ops"],
                      "leaf interpreter: algorithmic bytes = 8 B x (leaf tape ops x passes of the kernel over the tape) + the 16 B image pixel once, from "
                      "the device's counters of the timed frames; tape words are wave-uniform loads served by L2 / the scalar cache, so the kernel is bound "
                      "by instruction issue, not by HBM (DESIGN.md sections 4 and 6)")
        # tile levels of the frame: l0 the root level, the last one the per-slab level above the leaves (fh_tiles_v32), what lies between -
        # level 1 of the 128 / 32 / 8 hierarchy; none when the library took root tiles of 32^3 (capi_render.hpp root32_max) - fh_tiles_v64
        levels = sorted(int(k[1:]) for k in tiles)
        last = levels[-1] if levels else 0
        lv = [tiles[f"l{l}"] for l in levels if l == last and l > 0]
        mids = [tiles[f"l{l}"] for l in levels if 0 < l < last]
        r_tiles = roof(P, path, "fh_tiles_v32", 8.0 * (sum(v["ops"] for v in lv) + sum(v["ops_written"] for v in lv)),
                       64.0 * sum(v["ops"] for v in lv),
                       "interval interpreter + lockstep prune with the register file in VGPRs (per-slab level): bound by the latency of each parent's "
                       "dependent op chain (one wave per parent); algorithmic bytes = tape ops read + written at that level in the timed frames") if lv else None
        # the two kernels of the coarse chain: level 1 - fh_tiles_v64, one wave per 128^3 parent walking its tape
        # forward and in the lockstep prune - and the root level's prune (k_prune2 + fh_prune1 behind it, timed together as "fh_prune1")
        l0 = tiles.get("l0")
        r_l1 = roof(P, path, "fh_tiles_v64", 8.0 * (sum(v["ops"] for v in mids) + sum(v["ops_written"] for v in mids)), 64.0 * sum(v["ops"] for v in mids),
                    "level 1 (32^3 children of the 128^3 root tiles): interval interpreter + lockstep prune, register file in VGPRs; one wave per parent, "
                    "so the launch lasts as long as its longest parent's dependent chain (~1 000 ops forward, the same backwards); algorithmic bytes = "
                    "tape ops read + written at this level in the timed frames") if mids else None
        r_prune = roof(P, path, "fh_prune1", 8.0 * (3.0 * l0["ops"] + l0["ops_written"]), 0,
                       "root level's prune (prune2.hip k_prune2, one wave per child tape of the root tape, + fh_prune1 for the children it leaves): algorithmic "
                       "bytes = the root tape with its links (8 + 16 B per op) staged once per workgroup + the child tapes written") if l0 else None
        return r_leaf, r_tiles, r_l1, r_prune

    if world > 1:
        prof_general = prof_default = None      # (the roofline objects are rank 0's at N = 1, as the contract says)
    per_frame = lambda r: r["avg_launch_ms"] * r["launches_per_frame"] if r else 0.0
    if prof_general:
        r_leaf, r_tiles, r_l1, r_prune = roofs(prof_general, "general")
        # PRIMARY: the dominant kernel of the general path (what a model with z in its tapes gets) - with that path's own frame time,
        # rate and latency inside the object, so that a record that keeps `roofline` keeps them
        result["roofline"] = dict(max((r for r in (r_leaf, r_tiles, r_l1, r_prune) if r), key=per_frame))
        gp = general or {"ms_per_step": ms_per_step, "value": value, "frame_latency_ms": lat_default}      # (--only-general: the timed frames are the general path's)
        result["roofline"]["path_frame"] = {"path": "general (column-invariance short cuts off)", "ms_per_step": gp["ms_per_step"],
                                            "value": gp["value"], "unit": "Mvoxel/s", "frame_latency_ms": gp["frame_latency_ms"]}
        result["roofline_leaf"], result["roofline_tiles"], result["roofline_tiles_l1"], result["roofline_prune"] = r_leaf, r_tiles, r_l1, r_prune
    if prof_default:
        d_leaf, d_tiles, d_l1, d_prune = roofs(prof_default, "default")
        result["roofline_default_path"] = d_leaf
        result["roofline_tiles_default_path"] = d_tiles
        # the dominant kernel of the path that `value` times (prospero.vm: level 1, fh_tiles_v64)
        dom = max((r for r in (d_leaf, d_tiles, d_l1, d_prune) if r), key=per_frame)
        result["roofline_timed_path"] = dict(dom)
        result["roofline_timed_path"]["path_frame"] = {"path": "default (the frames `value` times)", "ms_per_step": ms_per_step, "value": value,
                                                       "unit": "Mvoxel/s", "frame_latency_ms": lat_default}
        result["roofline_tiles_l1_default_path"], result["roofline_prune_default_path"] = d_l1, d_prune
        if "roofline" not in result:
            result["roofline"] = result["roofline_timed_path"]
    if world == 1:
        result["device_counters"] = {k: {"leaf": v["leaf"], "tile_levels": v["tiles"]} for k, v in (("general", prof_general), ("default", prof_default)) if v}

    # ---- cpu_baseline + parity (oracle; rank 0, N = 1 only) -----------------------------------------------------------------
    if not args.no_cpu and world == 1:
        import oracle as O
        oshape = O.Shape.from_vm(os.path.join(ROOT, "models", args.model))
        ref, st, _ = O.render3d(oshape, n)                     # warm-up frame (also the parity reference)
        got = out.cpu().numpy().view(np.uint32).reshape(n, n, 4)
        want = ref.view(np.uint32).reshape(n, n, 4)
        result["parity"] = {"depth_equal": bool((got[..., 3] == want[..., 3]).all()),
                            "normals_equal": bool((got[..., :3].view(np.float32) == want[..., :3].view(np.float32)).all()),
                            "general_path_image_equals_default": None if not general else general["image_equals_default_path"]}
        cores = O.max_threads()
        CPU_FRAMES = 7
        secs = sorted(O.render3d(oshape, n)[2] for _ in range(CPU_FRAMES))
        med = float(np.median(secs))
        result["cpu_baseline"] = {"value": (n ** 3) / med / 1e6, "unit": "Mvoxel/s", "cores": cores, "kind": "port",
                                  "sample": f"median of {CPU_FRAMES} full {n}^3 frames after one warm-up ({med:.3f} s each, min {secs[0]:.3f}) on {cores} threads; "
                                            "C++ restatement of the reference's VmShape path, OpenMP over root tiles; not the Rust JIT (no toolchain)"}
        result["oracle_counters"] = {k: st[k] for k in ("interval_evals", "interval_ops", "float_evals", "float_points",
                                                         "float_lane_ops", "float_wave_ops", "grad_points")}
        # BASELINE configuration 3 (the gradient path on a tape with transcendental opcodes): frame time and the measured error
        # of the normals, in ulp of the gradient's scale (its largest component), against the oracle (glibc libm)
        bear = os.path.join(ROOT, "models", "bear.vm")
        if os.path.exists(bear) and args.model == "prospero.vm":
            m = 512
            bs, bo = F.Shape.from_vm(bear, hip=hip), O.Shape.from_vm(bear)
            bout = torch.zeros((m, m, 4), dtype=torch.int32, device=dev)
            for _ in range(60):      # (the library's arrangement tuner takes ~50 queued frames of a kind, capi_render.hpp lane_mode)
                F.render3d(bs, m, out=bout)
            fence()
            t0 = time.perf_counter()
            for _ in range(20):
                F.render3d(bs, m, out=bout)
            fence()
            bms = (time.perf_counter() - t0) / 20 * 1e3
            hip.sync()
            hip.profile(True)         # (one frame with the device's op counters: the leaf stage's algorithmic bytes)
            F.render3d(bs, m, out=bout)
            hip.profile_read()
            hip.wave_stats()
            bleaf = hip.leaf_stats()
            hip.profile(False)
            a = bout.cpu().numpy().view(np.uint32).reshape(m, m, 4)
            b = O.render3d(bo, m)[0]
            an, bn = a[..., :3].view(np.float32), b["normal"]
            scale = np.maximum(np.abs(bn).max(axis=2, keepdims=True), 2.0 ** -100)
            with np.errstate(invalid="ignore"):
                ulp = np.abs(an - bn) / (scale * 2.0 ** -23)
            balg = 8.0 * bleaf["tape_words_read"] + 16.0 * m * m
            result["c3_bear"] = {"workload": f"bear.vm 3D heightmap+normals {m}^3", "ms_per_frame": bms, "depth_equal": bool((a[..., 3] == b["depth"]).all()),
                                 "roofline": {"bound": "hbm", "achieved": balg / (bms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": balg / (bms * 1e-3) / 1e9 / 8000.0,
                                              "lane_ops_per_s_T": bleaf["lane_ops"] / (bms * 1e-3) / 1e12,
                                              "what": "leaf tape words x passes x 8 B + 16 B per pixel (device counters of one frame) over the frame time; its tapes' exp / ln / sin / cos "
                                                      "are 45-60 instructions per sample (DESIGN.md section 4), which the lane-op rate counts as one"},
                                 "normal_max_ulp_of_gradient_scale": float(np.nanmax(ulp)), "normals_bit_equal_fraction": float((an.view(np.uint32) == bn.view(np.uint32)).mean()),
                                 "frames": "20 queued back to back on one stream; the library runs whole queued frames on child contexts in turn where that measures faster "
                                           "(option frame_lanes; " + str(hip.lane_frames()) + " frames of this context went that way)",
                                 "note": "transcendental opcodes: the device runs the host libm's f32 routines restated operation by operation "
                                         "(fidget_amd/csrc/trans_libm.hpp; 0 of 2^32 arguments differ per routine, profiles/r04a/math_sweep.json)"}
        # BASELINE configuration 2 (prospero.vm 2D 4096^2) and a model that DOES read z at the headline size (colonnade.vm 1024^3: nothing
        # of the column-invariance short cuts applies to it): ms per queued frame, the image against the oracle's at full size, and the
        # leaf stage's share as a roofline fraction (algorithmic bytes = 8 B x tape ops x passes of the frames' own device counters
        # + the output pixels, over the frame time: an upper bound of any kernel's fraction in that frame)
        if args.model == "prospero.vm" and general:
            try:
                result["multi_gpu_predicted"] = {
                    "what": "8 GPUs, octant split, predicted on ONE GPU: slowest of the 8 blocks rendered alone + xGMI gather + depth merge (bench.py predict_n8)",
                    "headline": predict_n8(F, hip, torch, dev, fence, shape, n, False, lat_default),
                    "general_path": predict_n8(F, hip, torch, dev, fence, shape, n, True, general["frame_latency_ms"])}
            except Exception as e:      # noqa: BLE001
                result["multi_gpu_predicted"] = {"error": repr(e)[:200]}
        if args.model == "prospero.vm":
            try:
                result["c2_2d"] = side_config_2d(F, O, hip, torch, dev, fence, os.path.join(ROOT, "models", "prospero.vm"), 4096)
                col = os.path.join(ROOT, "models", "colonnade.vm")
                if os.path.exists(col):
                    result["c4z_colonnade"] = side_config_3d(F, O, hip, torch, dev, fence, col, 1024)
            except Exception as e:      # noqa: BLE001  (the line's other fields do not depend on these legs)
                result["c2_2d"] = result.get("c2_2d") or {"error": repr(e)[:200]}
        # BASELINE configuration 5 (Manifold Dual Contouring of gyroid-sphere at octree depth 10 = 1024^3: fhip_mesh_build, the octree
        # assembled on the device, the dual walk on the host's threads): seconds per build, in a process of its own (tools/mesh_times.py,
        # the script profiles/r03z/mesh_times.log comes from) so that nothing it does can cost this line
        if args.model == "prospero.vm" and os.path.exists(os.path.join(ROOT, "models", "gyroid-sphere.vm")):
            try:
                import subprocess
                env = dict(os.environ, MESH_TIMES_REPS="5", MESH_TIMES_PARITY="1")     # (the best of four builds after the first: one build in a few is slower by 0.2 s, one in eight by 2 s - DESIGN.md section 9)
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mesh_times.py"), "10"], env=env, capture_output=True, text=True, timeout=420)
                c5 = parse_mesh_times(r.stdout, r.stderr) if r.returncode == 0 else None
                result["c5_mesh"] = c5 if c5 else {"error": f"tools/mesh_times.py 10: rc {r.returncode}", "stderr_tail": r.stderr[-400:]}
            except Exception as e:      # (a time-out included: the line's other fields do not depend on this leg)
                result["c5_mesh"] = {"error": repr(e)}
    line = compact_line(result)
    details_path = write_details(result)
    if details_path:
        line["details"] = details_path
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < LINE_LIMIT, len(text)
    print(text)
    if world > 1:
        dist.destroy_process_group()


TUNING_CAP = 64         # untimed frames beyond --warmup the library's arrangement tuner may take PER KIND of frame (windows of 12, 12 + 4 and 12 queued frames + the verdict; `tuning_frames`
                        # in the line is the sum over the kinds timed - default path, general path - and says how many it took)
XGMI_LINK_GBS = 64.0        # one direction of one xGMI link as a collective sees it (7 links x ~153 GB/s both ways per GPU, MI355X_MICROARCH.md; ~85 % of 76)


def gather_ms(image_bytes, world):
    """rank 0 receives every other rank's rectangle over that rank's own link (point-to-point fabric: the 1 / N shares arrive side by side)"""
    return 0.0 if world < 2 else image_bytes / world / (XGMI_LINK_GBS * 1e9) * 1e3 + 0.02


def predict_from_stages(per_rank, n, world, partitions):
    """N > 1: the frame time the ranks' own stage times predict - the slowest rank's coarse chain (as long on every rank as on one GPU: one wave
    per parent whatever their number) + its share of the slab work, the gather of the 16-byte pixels over xGMI, the merge - next to the
    measured step, so that a scaling curve can be read against the model"""
    crit = max(per_rank, key=lambda q: q["coarse_chain_ms"] + q["slab_ms"])
    g = gather_ms(n * n * 16, world)
    render = crit["coarse_chain_ms"] + crit["slab_ms"]
    return {"model": "slowest rank's (coarse chain + slab stages, kernel times of a frame alone) + gather + merge", "critical_rank": crit["rank"],
            "render_ms": render, "coarse_chain_ms": crit["coarse_chain_ms"], "gather_ms": g, "merge_ms": crit.get("other_ms", 0.0),
            "frame_alone_ms": render + g + crit.get("other_ms", 0.0),
            "measured_frame_alone_ms": min(partitions[k]["frame_latency_ms"] for k in ("columns", "blocks")),
            "measured_ms_per_step": min(partitions[k]["ms_per_step"] for k in ("columns", "blocks")),
            "note": "queued frames overlap (the measured step is shorter than a frame alone); the coarse chain does not shrink with N"}


def predict_n8(F, hip, torch, dev, fence, shape, n, no_inv, frame_alone_ms):
    """N = 1: what eight GPUs would make of this frame under the north star's octant split, from THIS GPU - each of the 2 x 2 x 2 blocks
    rendered alone (waited for), the slowest of them + the gather of the rectangles over xGMI + the depth merge of the two z halves on rank 0
    (measured here on full-size halves).  A prediction, labelled as one: no 8-GPU node was available to the builder."""
    split = (2, 2, 2)
    out = torch.zeros((n, n, 4), dtype=torch.int32, device=dev)
    blocks = []
    ctx = hip.options(no_column_inv=1) if no_inv else hip.options()
    with ctx:
        for k in range(8):
            for _ in range(3):
                F.render3d(shape, n, out=out, block=(k, split))
            ts = []
            for _ in range(5):
                fence()
                t0 = time.perf_counter()
                F.render3d(shape, n, out=out, block=(k, split))
                torch.cuda.synchronize(dev)
                ts.append((time.perf_counter() - t0) * 1e3)
            blocks.append(float(np.median(ts)))
        hip.sync()
    a, b = out.clone(), out.clone()
    ts = []
    for _ in range(7):
        fence()
        t0 = time.perf_counter()
        F.merge_depth(a, b, n, hip=hip)
        torch.cuda.synchronize(dev)
        ts.append((time.perf_counter() - t0) * 1e3)
    merge = float(np.median(ts))
    g = gather_ms(n * n * 16, 8)
    pred = max(blocks) + g + merge
    return {"n_gpus": 8, "split": list(split), "one_gpu_frame_alone_ms": frame_alone_ms, "slowest_block_alone_ms": max(blocks), "fastest_block_alone_ms": min(blocks),
            "gather_ms": g, "merge_ms": merge, "predicted_frame_alone_ms": pred, "predicted_speedup": frame_alone_ms / pred}


def _queued_ms(step, fence, warm=60, frames=20):
    for _ in range(warm):      # (the library's arrangement tuner takes ~50 queued frames of a kind)
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(frames):
        step()
    fence()
    return (time.perf_counter() - t0) / frames * 1e3


def side_config_2d(F, O, hip, torch, dev, fence, model, n):
    import numpy as np
    s, o = F.Shape.from_vm(model, hip=hip), O.Shape.from_vm(model)
    out = torch.zeros((n, n), dtype=torch.float32, device=dev)
    ms = _queued_ms(lambda: F.render2d(s, n, out=out), fence)
    a = out.cpu().numpy().view(np.uint32)
    b = O.render2d(o, n, tile_sizes=F.HIP_TILES_2D)[0].view(np.uint32)
    px = n * n
    return {"workload": f"{os.path.basename(model)} 2D {n}^2", "ms_per_frame": ms, "image_equal": bool((a == b).all()),
            "roofline": {"bound": "hbm", "achieved": px * 4 / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": px * 4 / (ms * 1e-3) / 1e9 / 8000.0,
                         "what": "4 B per output pixel over the frame time (the frame is its tile chain: latency of dependent kernels, DESIGN.md section 6)"}}


def side_config_3d(F, O, hip, torch, dev, fence, model, n):
    import numpy as np
    s, o = F.Shape.from_vm(model, hip=hip), O.Shape.from_vm(model)
    out = torch.zeros((n, n, 4), dtype=torch.int32, device=dev)
    ms = _queued_ms(lambda: F.render3d(s, n, out=out), fence)
    hip.sync()
    hip.profile(True)         # (one frame with the device's op counters, as the main configuration's profiled frames)
    F.render3d(s, n, out=out)
    hip.profile_read()
    hip.wave_stats()
    leaf = hip.leaf_stats()
    hip.profile(False)
    a = out.cpu().numpy().view(np.uint32).reshape(n, n, 4)
    b = O.render3d(o, n)[0]
    nb = (a[..., :3] == b["normal"].view(np.uint32)) | (np.isnan(a[..., :3].view(np.float32)) & np.isnan(b["normal"]))
    alg = 8.0 * leaf["tape_words_read"] + 16.0 * n * n
    return {"workload": f"{os.path.basename(model)} 3D heightmap+normals {n}^3 (z in the tape: no column-invariance short cut applies)", "ms_per_frame": ms,
            "depth_equal": bool((a[..., 3] == b["depth"]).all()), "normals_equal": bool(nb.all()),
            "roofline": {"bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg / (ms * 1e-3) / 1e9 / 8000.0,
                         "what": "leaf tape words x passes x 8 B + 16 B per pixel (device counters of one frame) over the frame time"}}


LINE_LIMIT = 6000       # bytes: the driver's record keeps the line whole only when it is short (round 4's 20.6 KB line came back unparsed)


def _rnd(x, sig=5):
    """floats to `sig` significant digits (the line is a record, not a dump)"""
    if isinstance(x, float):
        return float(f"{x:.{sig}g}")
    if isinstance(x, dict):
        return {k: _rnd(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_rnd(v, sig) for v in x]
    return x


def _roof_short(r):
    """One roofline object of the line: the contract's fields (bound achieved peak unit frac traffic) + what they were computed from,
    the instruction-side figures as fractions; the prose lives in DESIGN.md section 5 and in the details file."""
    if not r:
        return None
    o = {k: r[k] for k in ("bound", "kernel", "path", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                           "avg_launch_ms", "launches_per_frame") if k in r}
    if r.get("alu"):
        o["alu"] = {k: r["alu"][k] for k in ("achieved", "peak", "unit", "frac")}
    if r.get("issue"):
        o["issue"] = {k: r["issue"][k] for k in ("achieved", "peak", "unit", "frac")}
    if r.get("path_frame"):
        o["path_frame"] = {k: r["path_frame"][k] for k in ("ms_per_step", "value", "frame_latency_ms")}
    if r.get("traffic_note"):
        o["traffic_note"] = r["traffic_note"][:120]
    return o


def compact_line(result):
    """The ONE JSON line: one object per fact, no prose beyond a sentence, < LINE_LIMIT bytes (tests/test_bench_protocol.py holds the key
    set and the size).  Everything else bench.py measured - per-kernel times, device counters, the other kernels' rooflines, the notes -
    goes to the details file the line names."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_median", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "tuning_frames", "frame_latency_ms", "host_output_frame_ms", "device_bytes", "device_bytes_timed_path")
    line = {k: result[k] for k in keep if k in result}
    cfg = result["config"]
    line["config"] = {"workload": cfg["workload"], "sharding": cfg["sharding"], "column_invariance": cfg["column_invariance"][:300],
                      "general_path": {k: cfg["general_path"][k] for k in ("ms_per_step", "value", "frame_latency_ms")} if cfg.get("general_path") else None}
    for k in ("roofline", "roofline_timed_path"):
        if result.get(k):
            line[k] = _roof_short(result[k])
    if result.get("cpu_baseline"):
        line["cpu_baseline"] = {k: result["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind", "sample")}
    if result.get("parity"):
        line["parity"] = result["parity"]
    if result.get("c3_bear"):
        line["c3_bear"] = {k: result["c3_bear"].get(k) for k in ("workload", "ms_per_frame", "depth_equal", "normals_bit_equal_fraction")}
        if result["c3_bear"].get("roofline"):
            line["c3_bear"]["roofline"] = {q: result["c3_bear"]["roofline"][q] for q in ("bound", "achieved", "peak", "unit", "frac")}
    if result.get("multi_gpu_predicted"):
        mp = result["multi_gpu_predicted"]
        line["multi_gpu_predicted"] = mp if "error" in mp else {k: ({kk: v[kk] for kk in ("one_gpu_frame_alone_ms", "slowest_block_alone_ms", "gather_ms", "merge_ms", "predicted_frame_alone_ms", "predicted_speedup")}
                                                                    if isinstance(v, dict) else v[:90]) for k, v in mp.items()}
    if result.get("predicted"):
        line["predicted"] = {k: v for k, v in result["predicted"].items() if k not in ("model", "note")}
    for k in ("c2_2d", "c4z_colonnade"):
        if result.get(k):
            line[k] = {kk: (vv if kk != "roofline" else {q: vv[q] for q in ("bound", "achieved", "peak", "unit", "frac")}) for kk, vv in result[k].items()}
    if result.get("c5_mesh"):
        c5 = result["c5_mesh"]
        line["c5_mesh"] = ({k: (c5.get(k) if k != "roofline" else {q: c5[k][q] for q in ("bound", "kernel", "achieved", "peak", "unit", "frac")})
                            for k in ("workload", "s_per_build", "s_per_build_inside_the_library", "triangles", "vertices", "parity", "cpu_s_per_build", "cpu_threads", "roofline") if k in c5}
                           if "error" not in c5 else {"error": str(c5["error"])[:200]})
    if result.get("partitions"):
        short = {}
        for k, v in result["partitions"].items():
            short[k] = {kk: vv for kk, vv in v.items() if kk in ("ms_per_step", "value", "frame_latency_ms", "scaling", "split", "frames_per_step",
                                                                  "images_equal")} if isinstance(v, dict) else v
        line["partitions"] = short
        line["collectives"] = (result.get("collectives") or "")[:100]
        line["per_rank"] = result.get("per_rank")
    line = _rnd(line)
    # should the line still outgrow the limit (a long error text, 8 ranks of stage times): shed the optional objects, never the contract's
    for k in ("per_rank", "multi_gpu_predicted", "c2_2d", "c4z_colonnade", "c3_bear", "c5_mesh", "parity", "roofline_timed_path"):
        if len(json.dumps(line, separators=(",", ":"))) < LINE_LIMIT - 200:
            break
        line.pop(k, None)
    return line


def write_details(result):
    """Everything bench.py measured, as one JSON file next to the run (gpurun_out/ travels back from the GPU box); the path goes into the line"""
    for d in ("gpurun_out", "."):
        try:
            os.makedirs(os.path.join(ROOT, d), exist_ok=True)
            rel = os.path.join(d, f"bench_details_n{result['n_gpus']}.json")
            with open(os.path.join(ROOT, rel), "w") as f:
                json.dump(result, f, indent=1)
            return rel
        except OSError:
            continue
    return None


if __name__ == "__main__":
    main()
