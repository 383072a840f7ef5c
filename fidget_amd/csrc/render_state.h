// Device-resident state of one render, shared by the host driver (capi.hip) and the
// kernels.  Plain C structs; all device pointers.
#pragma once
#include <stdint.h>

#include "tape_format.h"

#define FH_MAX_LEVELS 8
#define FH_MAX_INPUTS 16
#define FH_MAX_SLABS 64
#define FH_MAX_GROUPS 32  // independent sub-tapes of a root min / max (tape parallelism at level 0): at most; FHIP_GROUPS when a tape is built

// One wave's worth of interval work: a parent tile (or, at level 0, a run of root tiles)
// together with the tape that evaluates its children.
struct FhGroup {
    FhTapeRef tape;
    uint32_t x, y, z;     // parent corner in pixels / voxels (levels >= 1); z also used at level 0
    uint32_t first, n;    // level 0 only: first root-tile index (x-major) and how many (<= 16)
    uint32_t stride;      // level 0 only: index step between the group's root tiles (multi-GPU shards)
};

// A parent tile in flight between the kernels of the split 3D tile stage (setup -> evaluate +
// prune -> push); one entry per lane = per child tile.  SoA so that every access is coalesced.
#define FH_SLOT_LANES 64
// One op of the root tree over the terms (host_graph.hpp TopOp)
struct FhTopOp {
    uint8_t op, out, a_kind, b_kind;   // kinds: 0 top register, 1 term, 2 immediate
    uint32_t a, b;
};

struct FhSlot {
    FhTapeRef tape;                 // parent tape
    uint32_t level;
    uint64_t act;                   // in : children to evaluate (inside the image, not occluded)
    uint32_t base, overflow;        // out: arena reservation of the pruned tapes / arena full
    float* tvals;                   // in : tape groups: where this block's term intervals go ([term][lane])
    float xyz[6][FH_SLOT_LANES];    // in : x.lo x.hi y.lo y.hi z.lo z.hi (model space)
    uint32_t corner[3][FH_SLOT_LANES];  // in : child corner (voxels)
    float res[2][FH_SLOT_LANES];    // out: interval result
    uint32_t c_off[FH_SLOT_LANES];  // out: child tape (the parent's when nothing was pruned)
    uint32_t c_len[FH_SLOT_LANES];
    uint32_t c_rc[FH_SLOT_LANES];   //      n_regs | n_choices << 16
};

// An ambiguous smallest tile: evaluated point by point
#define FH_HIT_BUCKETS 64u
#define FH_HIT_STRIDE 64u

struct FhLeaf {
    FhTapeRef tape;
    uint32_t x, y, z;
};

// Entry of the 3D leaf table: the leaf kernel finds everything it needs to start on a leaf here, one load for the up to four
// leaves of a block of footprints, instead of one dependent load per leaf (the table entry, then the FhLeaf record)
struct FhLeafRef {
    uint32_t id;         // leaf index + 1 in `leaves` (what goes into the z-buffer word); 0 = no leaf
    uint32_t off;        // tape offset in the arena
    uint32_t len_regs;   // tape length | n_regs << 24
    uint32_t xy;         // corner x | y << 16
};

// fidget_raster::voxel::GeometryPixel (fidget-raster/src/voxel.rs:122-134)
struct FhGeometryPixel {
    float normal[3];
    uint32_t depth;
};

// Per-render constants
struct FhRender {
    float mat[16];                   // screen -> model, row major (voxel.rs:107-109 / pixel.rs:281-285)
    uint32_t width, height, depth;   // depth = 0 for 2D
    float z;                         // 2D slice height
    uint32_t pixel_perfect;
    uint32_t n_levels;
    uint32_t tiles[FH_MAX_LEVELS];   // tile sizes, largest first (after TileSizesRef trimming)
    uint32_t slab;                   // 3D: voxels per z-slab = a multiple of tiles[0] (one root-tile layer, or several taken in ONE step of the
                                     // per-slab chains: the tile chain's length is its number of steps, not the work per step)
    uint32_t roots_x, roots_y;       // root tile grid
    uint32_t max_regs, max_choices;  // of the root tape: bounds for every tape of the frame
    uint32_t in_kind[FH_MAX_INPUTS]; // per input slot: 0 x, 1 y, 2 z, 3 bound constant
    float in_value[FH_MAX_INPUTS];
    uint32_t tag[FH_MAX_LEVELS];     // 2D: the level a fill of this level says it was decided at (pixel.rs:225-229) - its own index, or, for a level the
                                     // library put between two of the caller's (fan-out above 64), the index of the caller's level below it
};

struct FhRenderState {
    FhRender P;
    // tape arena (8-byte ops)
    uint64_t* arena;
    uint32_t arena_cap, arena_head, arena_root_end, arena_overflow;
    // level queues: groups whose tape fits the small LDS layout grow from the front
    // (count / cursor), the others from the back (count_big / cursor_big)
    FhGroup* queue[FH_MAX_LEVELS];
    uint32_t count[FH_MAX_LEVELS], cursor[FH_MAX_LEVELS];
    uint32_t count_big[FH_MAX_LEVELS], cursor_big[FH_MAX_LEVELS];
    uint32_t qcap[FH_MAX_LEVELS];   // capacity of queue[l]
    uint32_t queue_overflow;
    // 3D: the first `pre_levels` tile levels are evaluated for ALL z-slabs in one go at the start
    // of the frame (their cost is latency, not throughput); their output, the level-`pre_levels`
    // work, is parked per slab in squeue[slab * squeue_cap ..] and their tapes stay in the arena
    // below arena_frame_end for the whole frame.
    uint32_t pre_levels, n_slabs, squeue_cap, arena_frame_end;
    FhGroup* squeue;
    uint32_t scount[FH_MAX_SLABS], scount_big[FH_MAX_SLABS];
    // split 3D tile stage: slots[0] = tapes that fit the small LDS layout, slots[1] = the others
    FhSlot* slots[2];
    uint32_t slot_cap[2];
    // tape parallelism at level 0 (host_graph.hpp plan_terms): the root min / max tree's terms are
    // evaluated by n_tgroups independent tapes (in the arena right after the root tape) into
    // tvals[block][term][lane]; the tree itself (ttop, n_top ops) runs over those values; chsrc tells
    // for every choice of the root tape where it was recorded; chwr[slot][word][lane] then holds the
    // choice words of the root tape as its own forward pass would have written them.
    FhTapeRef tgroup[FH_MAX_GROUPS];
    uint32_t n_tgroups, n_terms, n_top, top_chain, troot_len, troot_choices, troot_regs;
    const FhTopOp* ttop;
    const uint32_t* chsrc;
    float* tvals;        // [block][term][lane] x {lo, hi}
    uint8_t* topch;      // [block][child][top op]
    uint32_t* chwr;
    // pre-pass levels: choice words [slot][word][lane] of the forward pass, read by the
    // one-wave-per-child prune (fh_prune1); [0] small-LDS list (16 words per slot), [1] the other
    uint32_t* chw[2];
    uint32_t setup_cur[FH_MAX_LEVELS];
    uint32_t n_slots[2][FH_MAX_LEVELS], eval_cur[2][FH_MAX_LEVELS], push_cur[FH_MAX_LEVELS];
    // leaves
    FhLeaf* leaves;
    uint32_t leaf_cap, n_leaves, leaf_cursor, leaf_cursor_big, normal_cursor, normal_cursor_big;
    uint32_t n_leaves_lds;  // 3D: leaves of this slab that need the LDS register file (> 32 registers)
    // 3D: this frame met one of the rare large tapes - a leaf of more than 32 registers, or a parent of a per-slab tile level outside the
    // small slot list - in this slab context (never reset by a slab; k_finish3d tells the host: capi_render.hpp rare mode)
    uint32_t rare_seen;
    uint32_t norm_asm_regs; // ... and to be the normals kernel's (32: the HIP kernel with the small LDS file; 40: fh_normals) - a footprint with a leaf beyond: list 2, k_normals3d<big>
    uint32_t leaf_asm_regs; // 3D: the most registers a leaf may need to be the leaf kernel's (32: the HIP classes 0 and 1; the assembly kernels' largest shape) - beyond: k_leaves3d<2>
    FhLeafRef* leaf_table;  // 3D: [layer][footprint] -> leaf id + 1 and what the leaf kernel needs of the leaf (layer = 8-voxel layer of the slab)
    uint32_t slab_z;        // 3D: z of the current slab's first voxel (a leaf's z = slab_z + 8 * layer)
    uint32_t frame_stamp;   // a number no other frame of this context has: what the linked prune signs the links it leaves in the arena with (prune2.hip)
    // 3D: footprints that own leaves this slab, by register-file class (<=16, <=32, LDS)
    uint32_t* fp_list[3];
    uint32_t fp_count[3], fp_cursor[3];
    // 3D: the slab's leaves that own a hit of the finished z-buffer (leaf index + 1; k_hits3d), one entry = one wave pass of the normals
    // kernel: FH_HIT_BUCKETS counters FH_HIT_STRIDE words apart, then as many lists of the launch's `bucket_cap` entries each
    uint32_t* hit_list;
    // 3D: min-depth pyramid, one array per tile level
    uint32_t* mind[FH_MAX_LEVELS];
    // images
    uint64_t* zbuf;         // 3D: depth << 32 | leaf id (+1) of a hit whose normal is pending
    float* normals;         // 3D: 3 floats per pixel
    float* image2d;         // 2D: RawDistancePixel bits
    // Register files (and choice words, prune maps) of tapes too large for the 160 KB of LDS - the role of the reference's spill
    // slots (RegOp::Load / Store, compiler/alloc.rs:116-125: registers beyond the file live in memory): the root-sized kernel
    // variants then keep theirs in HBM, one region of gscratch_stride bytes per workgroup; 0: LDS as usual
    char* gscratch;
    uint32_t gscratch_stride, pad_gscratch;
    // statistics (optional, for bench / roofline accounting); want_stats: count tape ops per level (two
    // same-address atomics per parent tile: only in profiled frames)
    uint32_t want_stats, pad_stats;
    unsigned long long stat[64];
    // ... and of the leaf stage, counted where the leaves are queued (tpush_body, profiled frames only): [0] leaves, [1] their tape
    // ops, [2] tape ops x passes of the leaf kernel over the tape (<= 8 registers: one pass covers the leaf's 8 voxels per
    // pixel column, <= 16: two, <= 32: four, LDS class: eight; a leaf of a column-invariant parent: one) = 8-byte tape
    // words the leaf kernel reads, [3] tape ops x voxels evaluated (64 pixel columns x 8 voxels, or x 1 when column-invariant)
    unsigned long long leaf_stat[8];
};
