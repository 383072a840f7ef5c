#!/usr/bin/env python3
"""bench.py — headline benchmark of fidget-hip on MI355X.

Metric (BASELINE.json): Mvoxel/s of the heightmap+normals render (interval + point
evaluation) of prospero.vm at 1024^3, nominal volume / wall time
(W*H*D / t / 1e6, the convention of the reference's README.md:152-156).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (the driver launches N>1 through torch.distributed.run).  A step is
one full frame; frames are queued back to back and the library pipelines them (the coarse
levels of frame n + 1 beside the slabs of frame n; `frame_latency_ms` is one frame alone).  With N > 1 three
shardings of fidget_amd/dist.py are timed, K steps each.  One frame sharded over the ranks, no
collective inside the render, total work fixed ("strong"): "columns" (root-tile column index
% N == rank at full depth; ONE RCCL SUM reduce of the partial images) and "blocks" (the north
star's octants, 2 x 2 x 2 at N = 8: a gather of the ranks' own rectangles, then the
front-to-back depth merge on rank 0).  The frame SEQUENCE sharded by frame ("frames": every rank
renders whole frames, rank 0 gathers the finished frames; a step is then N frames and per-GPU
work is fixed: "weak").  A 1024^3 frame of this model is bound by the latency of its coarse tile
levels, which does not shrink with N, so sharding one frame gains little at this size and
sharding the sequence is what scales.  `value` is the fastest of the three
(named in config.sharding, `scaling` says which kind it is); all three are listed under
"partitions" with their own `value`.

Prints ONE JSON line on rank 0, including
  roofline     — for the dominant kernel (fh_tiles, the assembly tile-stage interpreter; a second
                 object covers fh_columns, the leaf interpreter): algorithmic
                 bytes (8 B per tape word per wavefront pass, exact because pruning
                 is deterministic; taken from the oracle's / the device's counters) / the kernel's
                 launches, each between its own pair of HIP events on the stream it is launched
                 on (three extra, un-pipelined frames after the timed loop), vs the 8 TB/s HBM peak;
  cpu_baseline — the C++ oracle (restatement of the reference's VmShape path, OpenMP over
                 root tiles like render_tiles' rayon pool) on this box's host cores, same frame.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--model", default="prospero.vm")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline / parity leg")
    ap.add_argument("--no-general", action="store_true", help="skip the frames with the column-invariance short cuts off (profiling runs: per-kernel statistics of the default path only)")
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
        # frame-level sharding of the frame SEQUENCE: every rank renders a whole frame of its own (frame r of each group of
        # `world` consecutive frames), rank 0 collects the finished frames; no collective inside a frame, no merge rule
        F.render3d(shape, n, out=out)
        gather_frames(out, frames_recv, dst=0)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    frame_ms = []

    def timed(step):
        for _ in range(args.warmup):
            step()
        fence()
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

    partitions = {}
    if world > 1:
        # the stream of the context is torch's current stream: collectives and renders are ordered on it
        dt_b = timed(step_blocks)
        img_b, ms_b = out.clone(), list(frame_ms)
        dt_c = timed(step_columns)
        ms_c = list(frame_ms)
        partitions = {"columns": {"ms_per_step": dt_c / args.steps * 1e3, "combine": "1 RCCL reduce (SUM) of the full image"},
                      "blocks": {"ms_per_step": dt_b / args.steps * 1e3, "split": list(split),
                                 "combine": "RCCL gather of each rank's rectangle + front-to-back depth merge on rank 0"}}
        if rank == 0:
            partitions["images_equal"] = bool(torch.equal(img_b, out))
            frames_recv = torch.zeros((world, n, n, 4), dtype=torch.int32, device=dev)
        img_c = out.clone()
        dt_f = timed(step_frames)        # one step = `world` frames
        partitions["frames"] = {"ms_per_step": dt_f / args.steps * 1e3, "frames_per_step": world,
                                "combine": "none inside a frame: every rank renders whole frames of the sequence; RCCL gather of the finished frames (16 MiB each) to rank 0"}
        if rank == 0:
            partitions["frames"]["images_equal"] = bool((frames_recv == img_c[None]).all())
        for k, f in (("columns", 1), ("blocks", 1), ("frames", world)):
            partitions[k]["value"] = (n ** 3) * f / (partitions[k]["ms_per_step"] * 1e-3) / 1e6
        # `value` is the fastest of the three.  One frame sharded over the ranks (A, B: the north star's split) is bound by
        # the latency of its coarse levels, which does not shrink with the rank count; a sequence of
        # frames sharded by frame (C) is what scales at this size, and then per-GPU work is fixed: "weak".
        dt_one, step_one, sharding_one = ((dt_c, step_columns, "root-tile columns round-robin, 1 RCCL reduce") if dt_c <= dt_b else
                                          (dt_b, step_blocks, f"blocks {split[0]}x{split[1]}x{split[2]} (octant split), RCCL gather + depth merge"))
        if dt_f / world < dt_one:
            dt, step, frames_per_step = dt_f, step_frames, world
            sharding = f"whole frames of the sequence round-robin over the ranks, RCCL gather of the finished frames to rank 0 (one frame sharded: {sharding_one}, see partitions)"
        else:
            dt, step, frames_per_step, sharding = dt_one, step_one, 1, sharding_one
            frame_ms[:] = ms_c if dt_c <= dt_b else ms_b
    else:
        step, frames_per_step = step_columns, 1
        dt = timed(step)
        sharding = "single GPU"
    # the same frames with the column-invariance short cuts off (FHIP_NO_COLUMN_INV: leaves evaluated once per voxel, every tile
    # of a z-stack evaluated) - prospero.vm has no z, so every one of its tapes takes them; this is what a model with z in
    # every tape gets from the same kernels
    general = None
    if world == 1 and not args.no_general:
        os.environ["FHIP_NO_COLUMN_INV"] = "1"
        saved = list(frame_ms)
        dt_g = timed(step)
        general = {"ms_per_step": dt_g / args.steps * 1e3, "ms_per_step_median": float(np.median(frame_ms)), "value": (n ** 3) * args.steps / dt_g / 1e6,
                   "note": "FHIP_NO_COLUMN_INV=1: no tape treated as independent of z (same image)"}
        frame_ms[:] = saved
        del os.environ["FHIP_NO_COLUMN_INV"]
        for _ in range(2):
            step()
    # one frame alone (nothing in flight before it, waited for): asynchronous frames are pipelined - the coarse levels of
    # frame n + 1 run beside the slabs of frame n - so the throughput above is not 1 / latency
    lat = []
    for _ in range(5):
        fence()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize(dev)
        lat.append((time.perf_counter() - t0) * 1e3)
    hip.sync()
    counters = hip.counters()

    # ---- per-kernel timing with HIP events on the render stream (separate, profiled frames) ----
    hip.profile(True)
    prof = {"tiles": [0.0, 0], "points": [0.0, 0], "normals": [0.0, 0], "other": [0.0, 0]}
    PROF_FRAMES = 3
    kern = {}   # per assembly kernel: every launch between its own pair of HIP events
    for _ in range(PROF_FRAMES):
        F.render3d(shape, n, out=out, shard=rank, n_shards=world)
        for k, (ms, cnt) in hip.profile_read().items():
            prof[k][0] += ms
            prof[k][1] += cnt
        for k, (ms, cnt) in hip.profile_read_kernels().items():
            kern.setdefault(k, [0.0, 0])
            kern[k][0] += ms
            kern[k][1] += cnt
    # ... and one profiled frame with the column-invariance short cuts off: the leaf kernel then does all the work the
    # algorithmic byte count stands for (with them on it skips most of it, and its roofline fraction flatters it)
    kern_general, tile_phases_general = {}, None
    if world == 1 and not args.no_general:
        os.environ["FHIP_NO_COLUMN_INV"] = "1"
        F.render3d(shape, n, out=out)
        hip.profile_read()
        for k, (ms, cnt) in hip.profile_read_kernels().items():
            kern_general[k] = (ms, cnt)
        hip.wave_stats()
        tile_phases_general = hip.tile_phases     # tape ops read / written per tile level when no tile is skipped as a copy along z
        del os.environ["FHIP_NO_COLUMN_INV"]
        hip.profile(False)
        F.render3d(shape, n, out=out)       # (leaves the default path's image in `out` and its counters in the context)
        hip.profile(True)
        F.render3d(shape, n, out=out)
        hip.profile_read(); hip.profile_read_kernels()
    hip.profile(False)
    hip.wave_stats()
    tile_phases = hip.tile_phases  # device counters of the last frame: tape ops read / written per tile level
    if world > 1:
        step()  # leave `out` holding the combined image on rank 0
        fence()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = dt / args.steps * 1e3
    value = (n ** 3) * frames_per_step * args.steps / dt / 1e6
    result = {
        "metric": "Mvoxel/s (interval+point eval) on prospero.vm 1024^3",
        "value": value, "unit": "Mvoxel/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "ms_per_step_median": float(np.median(frame_ms)), "ms_per_step_min": float(np.min(frame_ms)),
        "higher_is_better": True, "scaling": "weak" if frames_per_step > 1 else "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "frame_latency_ms": float(np.median(lat)),
        "without_column_invariance": general,
        "config": {"workload": f"{args.model} 3D heightmap+normals {n}^3, HipShape render hints (tiles 128/32/8), world_to_model=I",
                   "sharding": sharding,
                   "frames": "queued back to back on one stream, as a caller rendering a sequence would; the library pipelines them (two buffer "
                             "sets per context: the coarse levels of a frame run beside the previous frame's slabs), every frame does all of its "
                             "work; frame_latency_ms is one frame alone",
                   "column_invariance": "tapes that read no input varying along z (under this camera: no z) are evaluated once per pixel "
                                        "column / once per z-stack of tiles (DESIGN.md section 2); prospero.vm is an extrusion, so all of its "
                                        "tapes qualify - `without_column_invariance` times the same frames with the short cuts off"},
        "kernel_ms_per_frame": {k: v[0] / PROF_FRAMES for k, v in prof.items()},
        "kernel_launches_per_frame": {k: v[1] // PROF_FRAMES for k, v in prof.items()},
        "asm_kernel_ms_per_frame": {k: v[0] / PROF_FRAMES for k, v in kern.items() if v[1]},
        "arena_ops_last_slab": counters["arena_ops"], "arena_overflow": counters["arena_overflow"],
    }
    if partitions:
        result["partitions"] = partitions
        result["collectives"] = direct_note

    # ---- cpu_baseline + parity + algorithmic bytes (oracle; rank 0, N = 1 only) -----------------
    if not args.no_cpu and world == 1:
        import oracle as O
        oshape = O.Shape.from_vm(os.path.join(ROOT, "models", args.model))
        ref, st, _ = O.render3d(oshape, n)                     # warm-up frame (also the parity reference)
        got = out.cpu().numpy().view(np.uint32).reshape(n, n, 4)
        want = ref.view(np.uint32).reshape(n, n, 4)
        result["parity"] = {"depth_equal": bool((got[..., 3] == want[..., 3]).all()),
                            "normals_equal": bool((got[..., :3].view(np.float32) == want[..., :3].view(np.float32)).all())}
        cores = O.max_threads()
        CPU_FRAMES = 10
        secs = sorted(O.render3d(oshape, n)[2] for _ in range(CPU_FRAMES))
        med = float(np.median(secs))
        small = max(n // 4, 64)
        O.render3d(oshape, small, threads=1)
        one = float(np.median([O.render3d(oshape, small, threads=1)[2] for _ in range(3)]))
        result["cpu_baseline"] = {"value": (n ** 3) / med / 1e6, "unit": "Mvoxel/s", "cores": cores, "kind": "port",
                                  "sample": f"median of {CPU_FRAMES} full {n}^3 frames after one warm-up frame ({med:.3f} s each, min {secs[0]:.3f}, "
                                            f"on {cores} threads); C++ restatement of the reference VmShape interpreter path (OpenMP over root "
                                            "tiles like render_tiles' rayon pool), not the Rust JIT (published JIT/VM ratio on M1 Max: "
                                            "61.7/23.6 = 2.6x, README.md:154)",
                                  "one_thread": {"value": (small ** 3) / one / 1e6, "unit": "Mvoxel/s", "cores": 1,
                                                 "sample": f"median of 3 frames at {small}^3 (1/{(n // small) ** 3} of the volume), one thread"}}
        # ---- roofline ---------------------------------------------------------------
        # Algorithmic bytes: 8 B per tape word read per wavefront pass + 8 B per tape word written
        # + 2 bit per recorded choice + the W*H*16 B image once.  The tile stage's op counts come
        # from the device's own counters (its subdivision 128/32/8 differs from the oracle's
        # 128/64/32/16/8 schedule; pruning is deterministic, so they are exact for this frame);
        # the leaf stage's from the oracle (same leaves, same pruned tapes).
        # HBM traffic per launch: rocprofv3 PMC passes of tools/profile_round.sh, committed under profiles/ (counters cannot
        # be collected inside the timed run); the file names the hash of the device sources it was measured on and is
        # ignored (traffic = null) when that is not this build.  FETCH_SIZE doubled per MI355X_MICROARCH.md.
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from src_hash import source_hash
        traffic, traffic_note = {}, None
        tpath = os.path.join(ROOT, "profiles", "traffic_r02.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath))
            if traffic.get("source_hash") != source_hash():
                traffic_note = f"profiles/traffic_r02.json was measured on sources {traffic.get('source_hash')}, this build is {source_hash()}: traffic not reported"
                traffic = {}

        def roof(kernel, alg_bytes, k_ms, launches, note):
            # the kernel's own launches, each timed with HIP events on its stream (profiled frames); these
            # averages are the ones profiles/*/kernel_stats.csv (rocprofv3 --stats) has to agree with
            if kernel in kern and kern[kernel][1]:
                k_ms, launches = kern[kernel][0] / PROF_FRAMES, kern[kernel][1] // PROF_FRAMES
            achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
            t = traffic.get(kernel)
            tb = None
            if t:
                tb = (2.0 * t["fetch_kb_per_frame"] + t["write_kb_per_frame"]) * 1024.0 / max(t["launches_per_frame"], 1)
            r = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": achieved / HBM_PEAK_GBS, "traffic": tb,
                 "algorithmic_bytes_per_launch": alg_bytes / max(launches, 1), "avg_launch_ms": k_ms / max(launches, 1),
                 "launches_per_frame": launches, "note": note}
            if t and "valu_per_launch" in t and k_ms > 0:
                # the bound that actually holds: wave-instructions issued per SIMD per cycle against the ceiling the
                # micro-benchmarks give for independent VALU work at 4 waves per SIMD (profiles/r02/ubench.json)
                cycles = k_ms / max(launches, 1) * 1e-3 * t.get("clock_hz", 2.4e9)
                ipc = (t["valu_per_launch"] + t["salu_per_launch"]) / (cycles * 1024)
                r["issue"] = {"bound": "instruction issue", "achieved": ipc, "peak": 0.57, "unit": "wave-instructions / cycle / SIMD",
                              "frac": ipc / 0.57, "valu_per_launch": t["valu_per_launch"], "salu_per_launch": t["salu_per_launch"]}
            if traffic_note:
                r["traffic_note"] = traffic_note
            g = kern_general.get(kernel)
            if g and g[1] and g[0] > 0:
                # the same algorithmic bytes over the kernel's time when nothing is skipped as column-invariant
                r["without_column_invariance"] = {"avg_launch_ms": g[0] / g[1], "achieved": alg_bytes / (g[0] * 1e-3) / 1e9,
                                                  "frac": alg_bytes / (g[0] * 1e-3) / 1e9 / HBM_PEAK_GBS}
            return r

        kms, kl = result["kernel_ms_per_frame"], result["kernel_launches_per_frame"]
        # Two kernels carry the frame: fh_columns (leaf interpreter) and fh_tiles_v32 (tile stage of the per-slab level and part
        # of level 1).  `roofline` is the one with more time per frame in this run (profiles/: rocprofv3 --stats agrees), the other
        # follows as `roofline_leaf` / `roofline_tiles`.  ALGORITHMIC bytes: the leaf stage's from the oracle (every voxel of every
        # leaf), the tile stage's from the device's op counters of a frame with the column-invariance short cuts off (every tile of
        # the 128 / 32 / 8 subdivision) - prospero's tapes are column-invariant, so the default path evaluates one leaf per stack,
        # once per pixel, and one z-layer of tiles: it touches far fewer bytes (`traffic`), and `without_column_invariance` gives
        # the same figure with the short cuts off.
        leaf_bytes = 8.0 * st["float_wave_ops"] + n * n * 16
        r_leaf = roof("fh_columns", leaf_bytes, kms["points"], 8,
                      "tape words are wave-uniform loads served by L2: the leaf interpreter is bound by instruction issue (see `issue`), "
                      "not by HBM: DESIGN.md sections 4 and 6.  Algorithmic bytes are the oracle's (every voxel of every leaf); the "
                      "column-invariance short cuts skip most of that work for prospero.vm - see `without_column_invariance`")
        tp = tile_phases_general or tile_phases
        lv = [v for k, v in tp.items() if int(k[1:]) >= 2]
        tile_bytes = 8.0 * (sum(v["ops"] for v in lv) + sum(v["ops_written"] for v in lv))
        r_tiles = roof("fh_tiles_v32", tile_bytes, kms["tiles"], 8,
                       "interval interpreter + lockstep prune with the register file in VGPRs: bound by the latency of each parent's "
                       "dependent op chain (one wave per parent) and by instruction issue; algorithmic bytes = tape ops read + written at "
                       "the per-slab level with no tile skipped as a copy along z")
        def per_frame(r):
            return r["avg_launch_ms"] * r["launches_per_frame"]
        result["roofline"] = r_tiles if per_frame(r_tiles) > per_frame(r_leaf) else r_leaf
        result["roofline_leaf"], result["roofline_tiles"] = r_leaf, r_tiles
        result["oracle_counters"] = {k: st[k] for k in ("interval_evals", "interval_ops", "float_evals", "float_points",
                                                         "float_lane_ops", "float_wave_ops", "grad_points")}
    print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
