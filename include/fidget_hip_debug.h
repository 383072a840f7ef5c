/* fidget_hip_debug.h - diagnostics of libfidget_hip.so (NOT part of the drop-in surface of fidget_hip.h).
 * Used by tests/ and tools/ only: wave statistics, dumps of the device arena and queues, micro-benchmarks. */
#ifndef FIDGET_HIP_DEBUG_H
#define FIDGET_HIP_DEBUG_H
#include "fidget_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Wave busy-time statistics of the last render, 4 words per kernel kind (3D tile levels 0-4,
 * columns class 0, columns classes 1-2, 2D tiles): sum and max of per-wave busy ticks
 * (100 MHz), waves that found work, work units; then per 3D tile level: [32+l] ticks in the
 * forward interval pass, [40+l] ticks in classify + prune, [48+l] tape ops evaluated, [56+l] ops of pruned tapes written. */
fhip_status fhip_debug_stats(fhip_ctx* ctx, uint64_t out[64]);
/* Leaf-stage counters of the last profiled 3D frame, counted where the leaves are queued: [0] leaves, [1] their tape ops,
 * [2] tape ops x passes of the leaf kernel over the tape (8-byte words it reads), [3] tape ops x voxels evaluated. */
fhip_status fhip_debug_leaf_stats(fhip_ctx* ctx, uint64_t out[8]);
/* The per-op links of a tape for the linked prune (prune2.hip): word 0 = producer op of operand a | of operand b << 16 (0xFFFF none),
 * word 1 = choice index | op class << 16; returns the number of ops, 0 when the tape does not qualify. */
uint32_t fhip_debug_tape_links(const fhip_tape* tape, uint64_t* out, uint32_t cap);
/* ... and its root chain (acc = min / max(acc, term) all the way to the OUTPUT op) as the liveness pass of the linked prune gets it:
   choice ordinal | op index << 16 per chain op, evaluation order; returns the chain's length, 0 when the root is no chain */
uint32_t fhip_debug_tape_chain(const fhip_tape* tape, uint32_t* out, uint32_t cap);
/* Copies the FhLeaf records (24 bytes: tape offset, length, registers | choices << 16, x, y, z) of the
 * last slab of the last 3D frame; returns their number. */
uint32_t fhip_debug_leaves(fhip_ctx* ctx, void* out, uint32_t cap);
fhip_status fhip_debug_probe(fhip_ctx* ctx, float* out);  /* ISA probe (gen_interp.py gen_probe), 16 x 64 floats */
/* 3D frames of this context that went to a frame lane so far (option frame_lanes: whole frames of a queued sequence on child contexts;
 * the statistics the other debug calls return are then those of the last frame that did NOT) */
uint64_t fhip_debug_lane_frames(const fhip_ctx* ctx);
/* 3D frames of this context (its lanes included) rendered in rare mode so far: the launches that exist for tapes beyond the assembly kernels'
 * register files folded into launches the slab makes anyway - taken while the last finished frame met no such tape */
uint64_t fhip_debug_rare_frames(const fhip_ctx* ctx);
/* the arrangement tuner's state for the kind of 3D frame (tape, image size) queued last: returns its phase (0 / 1 / 2 measuring the stage pipeline,
 * the lanes, the stage pipeline again; 3 waiting for the last window's end; 4 decided; -1 none); ms[0 .. 2] = ms per frame of the three windows,
 * *lanes = the decision */
int fhip_debug_lane_tune(const fhip_ctx* ctx, float ms[3], int* lanes);
/* embedded copy `copy` of compiled routine `fn` (gen_trans.py COPIES / FUNCS + FUNCS4) over the floats with bits first .. first + n - 1
 * against the routine as the HIP kernels inline it: out = {results whose bits differ, an input where they do} */
fhip_status fhip_debug_trans_probe(fhip_ctx* ctx, uint32_t copy, uint32_t fn, uint32_t first, uint64_t n, uint64_t out[2]);
uint32_t fhip_debug_arena(fhip_ctx* ctx, uint32_t off, uint32_t n, uint64_t* out);  /* ops of the tape arena after a frame */
/* Times `reps` passes of the point interpreter over `tape` in `n_waves` waves
 * (variant 0: 16 registers x 4 voxels, 1: 32 x 2, 2: LDS register file, 3: 32 x 1). */
fhip_status fhip_debug_bench(fhip_ctx* ctx, const fhip_tape* tape, uint32_t n_waves, uint32_t reps, int variant,
                             double* ms);

/* Instruction-cost micro-benchmark `test` of fh_ubench (fidget_amd/csrc/gen_ubench.py): shader clocks per pattern, per wave */
fhip_status fhip_debug_ubench(fhip_ctx* ctx, uint32_t test, uint32_t iters, uint32_t n_waves, float* out);
/* Accuracy of transcendental opcode `op` (0 sin 1 cos 2 tan 3 asin 4 acos 5 atan 6 exp 7 ln) on the device against `ref` (the host
 * libm's f32 results for the floats with bit patterns first + i * stride, i < n): out = {max ulp distance, results that differ,
 * results more than 1 ulp apart, input bits of the worst case} */
fhip_status fhip_debug_math_sweep(fhip_ctx* ctx, int op, uint32_t first, uint32_t stride, uint64_t n, const float* ref, uint64_t out[4]);
/* Work-queue entries (36-byte FhGroup records: tape offset, length, registers | choices << 16, x, y, z, ...) the
 * last 3D frame left behind.  kind 0: queue of tile level `index`; kind 1: parked queue of z-slab `index`.
 * counts[0] = entries whose tape fits the small register-file layout (written first), counts[1] = the others. */
uint32_t fhip_debug_groups(fhip_ctx* ctx, int kind, uint32_t index, void* out, uint32_t cap, uint32_t counts[2]);

/* The host side of fhip_mesh_build on an octree given by the caller (no device involved): Octree::walk_dual over `cells`
 * ([n_cells][8] x {kind (0 invalid 1 empty 2 full 3 branch 4 leaf), mask, index}, u32 each), `root` (same three words) and `verts`
 * ([n_verts][3] floats), sequentially (parallel = 0: the recursion of dc.rs) or by independent sub-walks on the host's threads
 * (parallel = 1: what fhip_mesh_build runs).  Call with tris = verts_out = NULL for the sizes (counts = {triangles, vertices}),
 * then again with buffers of 3 u64 per triangle and 3 floats per vertex. */
void fhip_debug_walk_dual(const uint32_t* cells, uint64_t n_cells, const uint32_t* root, const float* verts, uint64_t n_verts, int parallel,
                          uint64_t counts[2], uint64_t* tris, float* verts_out);

#ifdef __cplusplus
}
#endif
#endif
