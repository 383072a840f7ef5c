"""Multi-GPU decomposition of the 3D render.

The frame shards by root-tile column: root tile `ri` (x-major index, fidget-raster/src/lib.rs:116-123)
belongs to rank `ri % world`.  A column keeps its full depth, so front-to-back occlusion culling
works per rank and no merge rule is needed: every pixel is produced by exactly one rank and is
zero on the others.  Partial images therefore combine with ONE integer SUM reduce of the raw
16-byte GeometryPixel words (bit exact, no float arithmetic involved).
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
    """[height, width] array: rank that renders each pixel."""
    roots_y = (height + root - 1) // root
    x = np.arange(width)[None, :] // root
    y = np.arange(height)[:, None] // root
    return ((x * roots_y + y) % world).astype(np.int32)


def combine(out, dst=0):
    """Sum the ranks' partial images (int32 view of GeometryPixel) onto rank `dst`.
    `out` is a torch tensor; a no-op for a single process."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(out, dst=dst, op=dist.ReduceOp.SUM)
    return out
