"""Multi-GPU decomposition of the 3D render: one process per GPU, no collective inside the render.

Two partitions of the volume, both exact (the 3D result is the per-pixel maximum depth with the normal of the winning
voxel, DESIGN.md §2, so it does not depend on how the volume is cut):

A. columns - root tile `ri` (x-major index, fidget-raster/src/lib.rs:116-123) belongs to rank `ri % world`, at full
   depth: front-to-back culling works per rank, every pixel is produced by exactly one rank (zero on the others), and
   the partial images combine with ONE integer SUM reduce of the raw 16-byte GeometryPixel words.

B. blocks (the north star's octants: 2 x 2 x 2 on 8 GPUs) - rank r renders block r of an nx x ny x nz split: a rectangle
   of root-tile columns and a range of z-slabs.  Rank 0 GATHERS the ranks' own rectangles (image / (nx * ny) pixels
   each, not the whole image) and merges the nz ranges of each rectangle front to back with the stitch rule of
   fidget-raster/src/voxel.rs:527-550 (larger depth wins, ties to the range nearer the camera, then the depth >= D-1
   clamp): fhip_merge_depth.  A z split forfeits occlusion culling between the ranges (the back ranks render what the
   front would have hidden), which is why A exists; bench.py measures both.

C. frames - a SEQUENCE of frames sharded by frame: every rank renders whole frames, rank 0 gathers the finished images
   (gather_frames).  A 1024^3 frame is bound by the latency of its coarse tile levels, which sharding the frame does not
   shorten; sharding the sequence is what scales at that size (bench.py measures it next to A and B).

The collectives can be queued through RCCL's C API on the render's own stream (DirectRccl): a collective library's stream
waiting for the frame's end costs the frame pipeline half its rate.  The mesh build shards by the root's octants
(mesh_sharded: fhip_mesh_sample_part per rank, fhip_mesh_merge on one).
"""
import numpy as np

VM_TILES_3D = (128, 64, 32, 16, 8)  # fidget-core/src/vm/mod.rs:251-253


def root_tile(max_dim, tiles=VM_TILES_3D):
    """Root tile size for an image (TileSizesRef::new, fidget-raster/src/lib.rs:59-66): the
    smallest listed size that still covers the image, or the largest one."""
    i = len(tiles)
    for k, t in enumerate(tiles):
        if t < max_dim:
            i = k
            break
    return tiles[max(i - 1, 0)]


def owner_map(width, height, root, world):
    """[height, width] array: rank that renders each pixel (partition A)."""
    roots_y = (height + root - 1) // root
    x = np.arange(width)[None, :] // root
    y = np.arange(height)[:, None] // root
    return ((x * roots_y + y) % world).astype(np.int32)


# ---- RCCL on the caller's stream ---------------------------------------------------------------------------------------
class DirectRccl:
    """An RCCL communicator driven through its C API, so that a collective is queued ON THE STREAM THE RENDER IS ON.

    torch.distributed runs its collectives on a stream of its own that waits for an event on the caller's stream.  On
    MI355X that wait - a barrier packet in whichever hardware queue the foreign stream shares with one of the library's
    streams - holds the next frame's coarse levels back until this frame has finished: measured on one GPU with a stand-in
    (tools/fifth_stream.py: a stream that only WAITS for the frame's end), the pipelined frame rate halves (1.02 -> 2.05 ms per
    frame).  Queued on the render's own stream the collective orders itself after the frame without any cross-stream wait.

    rank 0 creates the unique id; `bcast_bytes(b, src)` hands it to the other ranks (torch.distributed's broadcast of a byte
    tensor in bench.py).  One communicator per process / device, like the torch process group beside it."""
    INT32, SUM = 2, 0            # ncclInt32, ncclSum (rccl.h)

    @staticmethod
    def probe(lib_path=None):
        """load the library and look the entry points up, nothing else (raises if that fails): what every rank checks before
        any of them enters ncclCommInitRank, which is a collective"""
        DirectRccl(0, 0, lib_path=lib_path)

    def __init__(self, rank, world, bcast_bytes=None, lib_path=None):
        import ctypes as C
        import os
        self.C = C
        self.comm = None
        cands = [lib_path] if lib_path else []
        try:
            import torch
            cands.append(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))     # the one torch itself loaded
        except Exception:
            pass
        cands += ["librccl.so", "/opt/rocm/lib/librccl.so"]
        self.lib = None
        for c in cands:
            try:
                self.lib = C.CDLL(c)
                break
            except OSError:
                continue
        if self.lib is None:
            raise OSError("librccl.so not found")

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]

        L = self.lib
        L.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        L.ncclReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclCommDestroy.argtypes = [C.c_void_p]
        L.ncclGetErrorString.restype = C.c_char_p
        for f in (L.ncclGetUniqueId, L.ncclCommInitRank, L.ncclReduce, L.ncclGather, L.ncclCommDestroy):
            f.restype = C.c_int
        if world == 0:          # probe()
            return
        uid = UniqueId()
        if rank == 0:
            self._check(L.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        raw = C.string_at(C.addressof(uid), 128)
        if world > 1:
            if bcast_bytes is None:
                raise ValueError("a broadcast for the unique id is needed with more than one rank")
            raw = bcast_bytes(raw, 0)
            C.memmove(C.addressof(uid), raw, 128)
        self.rank, self.world = rank, world
        self.comm = C.c_void_p()
        self._check(L.ncclCommInitRank(C.byref(self.comm), world, uid, rank), "ncclCommInitRank")

    def _check(self, r, what):
        if r != 0:
            raise RuntimeError(f"{what}: {self.lib.ncclGetErrorString(r).decode()}")

    def reduce_sum(self, t, dst, stream):
        """in-place integer SUM of the int32 tensor `t` onto rank dst, queued on `stream` (a raw hipStream_t)"""
        self._check(self.lib.ncclReduce(t.data_ptr(), t.data_ptr(), t.numel(), self.INT32, self.SUM, dst, self.comm, stream), "ncclReduce")

    def gather(self, send, recv, dst, stream):
        """every rank's `send` (int32, same size) into `recv` on rank dst ([world * send.numel()], None elsewhere), on `stream`"""
        self._check(self.lib.ncclGather(send.data_ptr(), recv.data_ptr() if recv is not None else None, send.numel(), self.INT32, dst,
                                        self.comm, stream), "ncclGather")

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None


_direct = None


def use_direct_rccl(comm):
    """From here on combine / gather_blocks queue their collective through `comm` (a DirectRccl) on torch's current stream."""
    global _direct
    _direct = comm


def combine(out, dst=0):
    """Partition A: sum the ranks' partial images (int32 view of GeometryPixel) onto rank `dst`.
    `out` is a torch tensor; a no-op for a single process."""
    import torch.distributed as dist
    if _direct is not None and _direct.world > 1:
        import torch
        _direct.reduce_sum(out, dst, torch.cuda.current_stream(out.device).cuda_stream)
        return out
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(out, dst=dst, op=dist.ReduceOp.SUM)
    return out


# ---- partition B ---------------------------------------------------------------------------------------------------
def block_split(world):
    """(nx, ny, nz) with nx * ny * nz == world: powers of two go to x, y, z in turn (8 -> 2 x 2 x 2: octants), any odd
    factor to x."""
    s = [1, 1, 1]
    w, axis = world, 0
    while w % 2 == 0 and w > 1:
        s[axis % 3] *= 2
        axis += 1
        w //= 2
    s[0] *= w
    return tuple(s)


def block_coords(index, split):
    nx, ny, nz = split
    return index % nx, (index // nx) % ny, index // (nx * ny)


def _axis_range(n_roots, parts, i, root, limit):
    """pixel range of the root tiles t with t * parts // n_roots == i (the rule of fhip_render3d_block)"""
    ts = [t for t in range(n_roots) if t * parts // n_roots == i]
    if not ts:
        return 0, 0
    return min(ts[0] * root, limit), min((ts[-1] + 1) * root, limit)


def block_rect(width, height, root, split, index):
    """(y0, y1, x0, x1): the pixels block `index` renders (its root-tile columns, clipped to the image)"""
    ix, iy, _ = block_coords(index, split)
    x0, x1 = _axis_range((width + root - 1) // root, split[0], ix, root, width)
    y0, y1 = _axis_range((height + root - 1) // root, split[1], iy, root, height)
    return y0, y1, x0, x1


def gather_blocks(out, image_depth, split, merge, dst=0, root=None):
    """Partition B: every rank passes its partial image `out` ([H, W, 4] int32 torch tensor, zero outside its block);
    rank `dst` gathers the ranks' rectangles and merges the z ranges of each with `merge(front, back, image_depth)`
    (in place on `front`; fidget_amd.merge_depth on the GPU).  Returns the full image on `dst` (in `out`), `out` elsewhere.
    `root`: the root tile size of the RENDER (its tile_sizes after trimming, lib.rs:59-66) when it was given explicit
    tile sizes - the blocks are runs of root-tile columns, so the rectangles gathered here must be cut with the same size;
    default: the root tile of the default tile list, which is what render3d(block=...) without tile_sizes uses."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    H, W = out.shape[0], out.shape[1]
    root = root_tile(max(W, H)) if root is None else int(root)
    assert root > 0 and root % 8 == 0, "root tile size of the render (a multiple of the 8-voxel leaf)"
    rects = [block_rect(W, H, root, split, r) for r in range(world)]
    area = max(max((y1 - y0) * (x1 - x0) for y0, y1, x0, x1 in rects), 1)
    y0, y1, x0, x1 = rects[rank]
    send = torch.zeros((area, 4), dtype=out.dtype, device=out.device)
    n_mine = (y1 - y0) * (x1 - x0)
    if n_mine:
        send[:n_mine] = out[y0:y1, x0:x1].reshape(n_mine, 4)
    if world > 1 and _direct is not None and _direct.world == world:
        recv = torch.empty((world,) + tuple(send.shape), dtype=send.dtype, device=send.device) if rank == dst else None
        _direct.gather(send, recv, dst, torch.cuda.current_stream(send.device).cuda_stream)
        parts = [recv[r] for r in range(world)] if rank == dst else None
    elif world > 1:
        parts = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
        dist.gather(send, parts, dst=dst)
    else:
        parts = [send]
    if rank != dst:
        return out
    return assemble_blocks(parts, out, rects, split, image_depth, merge)


def gather_frames(out, recv, dst=0):
    """Partition C (frame-level sharding of a SEQUENCE of frames): every rank has rendered a whole frame of its own into `out`
    ([H, W, 4] int32); rank `dst` collects them into `recv` ([world, H, W, 4], None elsewhere) - frame r of the group of
    `world` consecutive frames comes from rank r.  No merge rule: the frames are independent."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    if world > 1 and _direct is not None and _direct.world == world:
        import torch
        _direct.gather(out, recv if rank == dst else None, dst, torch.cuda.current_stream(out.device).cuda_stream)
    elif world > 1:
        dist.gather(out, [recv[r] for r in range(world)] if rank == dst else None, dst=dst)
    elif recv is not None:
        recv[0].copy_(out)
    return recv


def assemble_blocks(parts, out, rects, split, image_depth, merge):
    """rank-0 half of gather_blocks: parts[r] = block r's rectangle as [area, 4] words (padded), rects[r] its pixel range"""
    world = len(parts)
    nx, ny, nz = split
    for bxy in range(nx * ny):
        ranks = [bxy + nx * ny * iz for iz in range(nz - 1, -1, -1)]     # nearest the camera first
        ranks = [r for r in ranks if r < world]
        y0, y1, x0, x1 = rects[ranks[0]]
        n = (y1 - y0) * (x1 - x0)
        if n == 0:
            continue
        acc = parts[ranks[0]][:n].contiguous()
        for r in ranks[1:]:
            merge(acc, parts[r][:n].contiguous(), image_depth)
        out[y0:y1, x0:x1] = acc.reshape(y1 - y0, x1 - x0, 4)
    return out


# ---- the mesh build sharded by the root's octants (Mesh) --------------------------------------------------
def mesh_part_octants(part, n_parts):
    """the root's octants part `part` of `n_parts` evaluates (the rule of fhip_mesh_sample_part): o * n_parts // 8 == part"""
    return [o for o in range(8) if o * n_parts // 8 == part]


def gather_bytes(buf, dst=0, device=None):
    """Variable-length uint8 buffers of all ranks -> list in rank order on rank `dst` (None elsewhere): lengths by one
    all_gather, then one send / recv per rank (the gather of variable-length cells and vertices of SURVEY 8e).  `device`:
    where the tensors of the exchange live (a CUDA device for the nccl backend, None = host for gloo)."""
    import torch
    import torch.distributed as dist
    buf = np.ascontiguousarray(buf, np.uint8)
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if world == 1:
        return [buf]
    rank = dist.get_rank()
    size = torch.tensor([buf.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size)
    if rank == dst:
        out = []
        for r in range(world):
            if r == dst:
                out.append(buf)
                continue
            t = torch.empty(int(sizes[r].item()), dtype=torch.uint8, device=device)
            if t.numel():
                dist.recv(t, src=r)
            out.append(t.cpu().numpy())
        return out
    if buf.size:
        dist.send(torch.from_numpy(buf).to(device) if device is not None else torch.from_numpy(buf), dst=dst)
    return None


class _SharedParts:
    """Transport of the parts between the processes of ONE node without a copy through the devices: every rank writes its
    part into a POSIX shared-memory segment of its own, rank `dst` maps the others' segments and merges out of them."""

    def __init__(self, tag):
        self.tag, self.own, self.maps = tag, None, []

    def name(self, rank):
        return f"fhip_mesh_{self.tag}_{rank}"

    def alloc(self, rank, nbytes):
        from multiprocessing import shared_memory
        self.own = shared_memory.SharedMemory(name=self.name(rank), create=True, size=max(int(nbytes), 1))
        return np.frombuffer(self.own.buf, np.uint8, int(nbytes))

    def attach(self, rank, nbytes):
        from multiprocessing import shared_memory
        m = shared_memory.SharedMemory(name=self.name(rank))
        try:        # (Python < 3.13 registers attached segments with this process' resource tracker too, which would unlink the
            #          owner's segment a second time at exit and say so on stderr: the owner unlinks it, nobody else)
            from multiprocessing import resource_tracker
            resource_tracker.unregister(m._name, "shared_memory")
        except Exception:
            pass
        self.maps.append(m)
        return np.frombuffer(m.buf, np.uint8, int(nbytes))

    def close(self):
        for m in self.maps:
            m.close()
        self.maps = []
        if self.own is not None:
            self.own.close()
            self.own.unlink()
            self.own = None


def mesh_sharded(make_part, merge, dst=0, transport=None, device=None):
    """Octree::build_inner_mt across the ranks (fidget-mesh/src/octree.rs:94-210): rank r runs the device side of the build
    for its octants - `make_part(r, world, alloc)` returns the flat part buffer (fidget_amd.mesh_part; `alloc(nbytes)` hands
    out the memory to write it into) - the buffers travel to rank `dst`, which calls `merge(parts)` (fidget_amd.mesh_merge)
    and returns its result; the other ranks return None.  Transport "shm": shared memory (all ranks on one node - the
    default when they are); "dist": gather_bytes over the process group.  More than 8 ranks: the ranks above 7 idle."""
    import os
    import socket
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    n_parts = min(world, 8)
    if world == 1:
        return merge([make_part(0, 1, lambda n: np.zeros(int(n), np.uint8))])
    if transport is None:
        hosts = [None] * world
        dist.all_gather_object(hosts, socket.gethostname())
        transport = "shm" if len(set(hosts)) == 1 else "dist"
    if transport == "dist":
        mine = make_part(rank, n_parts, lambda n: np.zeros(int(n), np.uint8)) if rank < n_parts else np.zeros(0, np.uint8)
        parts = gather_bytes(mine, dst=dst, device=device)
        return merge(parts[:n_parts]) if rank == dst else None
    tag = [f"{os.getpid()}_{int.from_bytes(os.urandom(4), 'little')}"] if rank == dst else [None]
    dist.broadcast_object_list(tag, src=dst)
    shm = _SharedParts(tag[0])
    mine = None
    try:
        mine = make_part(rank, n_parts, lambda n: shm.alloc(rank, n)) if rank < n_parts else np.zeros(0, np.uint8)
        sizes = [None] * world
        dist.all_gather_object(sizes, int(mine.size))        # (also the barrier: every segment is written)
        result = None
        err = None
        if rank == dst:
            try:
                # (a rank whose make_part never asked `alloc` for memory has no segment: its part is empty, nothing to attach)
                parts = [mine if r == rank else (shm.attach(r, sizes[r]) if sizes[r] else np.zeros(0, np.uint8)) for r in range(n_parts)]
                result = merge(parts)
                del parts
            except Exception as e:      # noqa: BLE001 - the other ranks wait at the barrier below: reach it, then raise
                err = e
        dist.barrier()                                         # the segments stay until the merge has read them
        if err is not None:
            raise err
        return result
    finally:
        mine = None          # (the arrays over a segment must be gone before it can be closed)
        shm.close()
