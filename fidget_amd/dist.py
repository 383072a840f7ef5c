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


def combine(out, dst=0):
    """Partition A: sum the ranks' partial images (int32 view of GeometryPixel) onto rank `dst`.
    `out` is a torch tensor; a no-op for a single process."""
    import torch.distributed as dist
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


def gather_blocks(out, image_depth, split, merge, dst=0):
    """Partition B: every rank passes its partial image `out` ([H, W, 4] int32 torch tensor, zero outside its block);
    rank `dst` gathers the ranks' rectangles and merges the z ranges of each with `merge(front, back, image_depth)`
    (in place on `front`; fidget_amd.merge_depth on the GPU).  Returns the full image on `dst` (in `out`), `out` elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    H, W = out.shape[0], out.shape[1]
    root = root_tile(max(W, H))
    rects = [block_rect(W, H, root, split, r) for r in range(world)]
    area = max(max((y1 - y0) * (x1 - x0) for y0, y1, x0, x1 in rects), 1)
    y0, y1, x0, x1 = rects[rank]
    send = torch.zeros((area, 4), dtype=out.dtype, device=out.device)
    n_mine = (y1 - y0) * (x1 - x0)
    if n_mine:
        send[:n_mine] = out[y0:y1, x0:x1].reshape(n_mine, 4)
    if world > 1:
        parts = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
        dist.gather(send, parts, dst=dst)
    else:
        parts = [send]
    if rank != dst:
        return out
    return assemble_blocks(parts, out, rects, split, image_depth, merge)


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
