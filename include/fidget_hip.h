/* fidget_hip.h — C ABI of libfidget_hip.so: the MI355X (gfx950) backend for Fidget's
 * evaluation hot path.
 *
 * This is the drop-in boundary.  A Rust crate `fidget-hip` binds these symbols
 *  and implements fidget_core::eval::{Function, MathFunction,
 * Tape, TracingEvaluator, BulkEvaluator} + render::RenderHints on top of them, the
 * same way `fidget-jit` wraps `VmData` and replaces only tapes and evaluators
 * (/root/reference/fidget-jit/src/lib.rs:872-998).  Every entry point names the
 * reference interface it replaces.
 *
 * Conventions: plain C, no unwinding; every call returns an fhip_status; all
 * `const T*` / `T*` arguments are CALLER-OWNED HOST buffers unless a parameter is
 * documented as a device pointer; one fhip_ctx per host thread / HIP stream
 * (evaluators in the reference are per-thread too, fidget-raster/src/lib.rs:129-133).
 */
#ifndef FIDGET_HIP_H
#define FIDGET_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum fhip_status {
    FHIP_OK = 0,
    FHIP_ERR_BAD_VAR_SLICE = 1,     /* var/mod.rs:151-165  TracingArgError / BulkArgError::BadVarSlice */
    FHIP_ERR_MISMATCHED_SLICES = 2, /* var/mod.rs:167-197  BulkArgError::MismatchedSlices */
    FHIP_ERR_BAD_CHOICE_SLICE = 3,  /* vm/data.rs:129-134  BadTrace(BadChoiceSlice) */
    FHIP_ERR_MISSING_VAR = 4,       /* shape/mod.rs:391-396 MissingVar */
    FHIP_ERR_BAD_TAPE = 5,          /* malformed bytecode / bad node (context BadNode) */
    FHIP_ERR_UNSUPPORTED = 6,       /* e.g. > 256 live registers, tile fan-out > 64 */
    FHIP_ERR_HIP = 7,               /* a HIP runtime call failed; see fhip_last_error */
    FHIP_ERR_CANCELLED = 8,         /* render/config.rs:38-80 CancelToken */
    FHIP_ERR_PARSE = 9,             /* context/mod.rs ParseError */
    FHIP_ERR_OVERFLOW = 10          /* a device work queue overflowed (sizes are bounds; should not happen) */
} fhip_status;

typedef struct fhip_ctx fhip_ctx;     /* device, stream, scratch pools        */
typedef struct fhip_tape fhip_tape;   /* one compiled function (host + device) */
typedef struct fhip_graph fhip_graph; /* host mirror of fidget_core::Context   */

/* ---- context ------------------------------------------------------------------------ */
/* `stream` is a hipStream_t (NULL = the default stream), e.g. torch's current stream. */
fhip_status fhip_ctx_create(int device, void* stream, fhip_ctx** out);
void fhip_ctx_destroy(fhip_ctx* ctx);
const char* fhip_last_error(const fhip_ctx* ctx);
fhip_status fhip_ctx_sync(fhip_ctx* ctx);
/* Gives back the device and pinned memory a context keeps between calls for speed alone: the mesher's leaf records (17 GB after a
 * depth-10 build), its landing area, the frame lanes (child contexts with buffers of their own - up to four, each a context's worth of
 * buffers: option frame_lanes).  Waits for the context's work; whatever is needed again is made again by the call that needs it. */
fhip_status fhip_ctx_trim(fhip_ctx* ctx);
/* The tape arena of a buffer set is sized by need: 128 MB, or about a byte per voxel for a 3D frame whose root tape reads an input that
 * changes along a pixel column, grown when a frame ran out (that frame is still right: its tiles kept their parents' tapes - and many
 * times slower), never beyond option arena_mb.  A caller that knows better - a sequence of frames of a heavier model to come - says
 * so here: every set's arena is at least `megabytes` from the next frame on (capped by arena_mb).  Waits for the context's own work. */
fhip_status fhip_ctx_reserve_arena(fhip_ctx* ctx, size_t megabytes);
/* The device evaluates sin cos tan asin acos atan atan2 exp ln with the routines of glibc 2.35's x86-64 libm (its FMA variants)
 * restated operation by operation (fidget_amd/csrc/trans_libm.hpp) - the libm the reference's f32 methods call on the deployment
 * image, whose values its own bulk test demands bit for bit (fidget-core/src/eval/test/float_slice.rs:404-412).  On a host with
 * ANOTHER libm (glibc >= 2.41's CORE-MATH routines, a CPU without FMA, musl) the reference's CPU evaluators - and this repository's
 * oracle - return other bits for some arguments, and a comparison of device against host values fails for that reason alone.
 * This call says so: 32 arguments per routine (ordinary values, the reduction's range boundaries, tiny and huge ones) through the
 * restated routines compiled for the host against the RUNNING libm; returns the number that differ (0: this host's libm is the one
 * the device restates) and, when `msg` is given, the first difference by name - "sinf(0x1.8p+1): device family 0x..., host libm
 * 0x...".  No device is touched.  fhip_ctx_create runs it once per process and prints the message to stderr if there is one
 * (FHIP_QUIET=1: does not). */
int fhip_libm_probe(char* msg, size_t cap);
/* render/config.rs:38-80: cooperative cancellation, honoured between kernel waves */
void fhip_cancel(fhip_ctx* ctx);
void fhip_cancel_reset(fhip_ctx* ctx);
/* A cancel flag of the caller's - one byte, non-zero = cancel - that the context's checks read beside its own until fhip_cancel_watch(ctx,
 * NULL): fidget_core::render::CancelToken (render/config.rs:38-78) is an Arc<AtomicBool> whose address into_raw() hands out, so the
 * Rust crate passes the EvalConfig's token for the duration of a render (voxel.rs:52-61; its cancel_render test, voxel.rs:573-590) and
 * a cancel from any thread ends the render with FHIP_ERR_CANCELLED.  The byte must stay valid while it is watched. */
void fhip_cancel_watch(fhip_ctx* ctx, const void* flag);
/* Behaviour switches of a context (the role of the reference's config structs - ThreadPool / TileSizes / RenderConfig carry
 * the reference's knobs; these carry the back end's own: which kernels, how many slab contexts, the column-invariance short
 * cuts ...).  A context reads FHIP_<NAME> from the environment ONCE, when it is created; afterwards only this call changes
 * a switch - a render never consults the environment.  It waits for the context's frames in flight first.  Names and
 * defaults: FH_OPTION_LIST in fidget_amd/csrc/capi_core.hpp; e.g. "no_column_inv", "frame_lanes",
 * "arena_mb".  Unknown names (and the two that are fixed at creation) return FHIP_ERR_UNSUPPORTED. */
fhip_status fhip_ctx_set_option(fhip_ctx* ctx, const char* name, int value);
fhip_status fhip_ctx_get_option(const fhip_ctx* ctx, const char* name, int* value);

/* ---- tapes -------------------------------------------------------------------------- */
/* The wire format produced by fidget_bytecode::Bytecode::new(&VmData)
 * (fidget-bytecode/src/lib.rs:11-42, 203-332).  Replaces JitFunction's
 * `point_tape/interval_tape/float_slice_tape/grad_slice_tape` compilation
 * (fidget-jit/src/lib.rs:875-908): one device tape serves all four evaluators.
 * Variable slots are the reference's VarMap indices. */
fhip_status fhip_tape_from_bytecode(fhip_ctx* ctx, const uint32_t* words, size_t n_words, fhip_tape** out);
void fhip_tape_free(fhip_tape* tape);
uint32_t fhip_tape_len(const fhip_tape* tape);          /* Function::size      eval/mod.rs:171 */
uint32_t fhip_tape_choice_count(const fhip_tape* tape); /* Function::can_simplify != 0 */
uint32_t fhip_tape_reg_count(const fhip_tape* tape);
uint32_t fhip_tape_var_count(const fhip_tape* tape);    /* Tape::vars().len()  eval/mod.rs:41 */
uint32_t fhip_tape_output_count(const fhip_tape* tape); /* Tape::output_count  eval/mod.rs:47 */
/* Copy the device-format ops (8 bytes each, evaluation order) to `ops`; returns the length */
uint32_t fhip_tape_ops(const fhip_tape* tape, uint64_t* ops, uint32_t cap);
/* The tape as the reference's VmData<N> holds it: RegTape::new::<N> (fidget-core/src/compiler/reg_tape.rs:26-32) - the
 * single-pass RegisterAllocator<N> with its LRU eviction and Load / Store spills to memory slots >= N
 * (compiler/alloc.rs:13-708, compiler/lru.rs:19-76) - over this tape's ops, and Bytecode::new of that
 * (fidget-bytecode/src/lib.rs:203-332).  Host side only; what the device runs is the library's own dense allocation.
 * reg_ops (may be NULL): 4 words per RegOp in evaluation order (VmData::iter_asm, vm/data.rs:320-323): opcode (tape_format.h FhOp; 52
 * Load, 53 Store), out, a (lhs / the stored register), then b, or the immediate bits, or the input / output / memory slot.
 * words (may be NULL): the bytecode, start and end markers included.  info = { RegTape::len(), RegTape::slot_count(),
 * Bytecode::reg_count, Bytecode::mem_count }.  1 <= n_regs <= 255.  FHIP_ERR_UNSUPPORTED: the reserved register 255 would be in
 * use (lib.rs ReservedRegister); info[0..1] are valid then. */
fhip_status fhip_tape_reg_tape(const fhip_tape* tape, uint32_t n_regs, uint32_t* reg_ops, uint32_t cap_ops, uint32_t* words,
                               uint32_t cap_words, uint32_t info[4]);

/* Tape parallelism (no counterpart in the reference): when the root of the function is a min / max
 * of many parts, the same function as `count` independent tapes whose outputs combine, in order,
 * with FH_MIN_RR (30) / FH_MAX_RR (31).  The renderers evaluate the root level that way; exposed
 * for tests.  0 groups = the tape does not split. */
uint32_t fhip_tape_group_count(const fhip_tape* tape);
int fhip_tape_group_op(const fhip_tape* tape);
fhip_status fhip_tape_group(fhip_ctx* ctx, const fhip_tape* tape, uint32_t g, fhip_tape** out);
/* The form of that split the 3D renderer uses at its root level: groups that output the terms of the
 * root min / max tree, the tree as a small program over them, and a table saying where each choice of
 * the full tape is recorded.  Returns the number of groups (0: not split);
 * info = { terms, tree ops, tree registers, choices covered (= fhip_tape_choice_count) }. */
uint32_t fhip_tape_term_plan(const fhip_tape* tape, uint32_t info[4]);
/* ... its parts, for tests: group g as a tape of its own (its OUTPUT ops carry the term index);
 * the tree, 3 words per op: op | out << 8 | a_kind << 16 | b_kind << 24 (kinds: 0 tree register, 1 term,
 * 2 immediate bits), a, b (returns the number of ops); and per choice of the full tape, in tape order,
 * where it is recorded: group << 24 | choice index there, group 255 = op index of the tree. */
fhip_status fhip_tape_term_group(fhip_ctx* ctx, const fhip_tape* tape, uint32_t g, fhip_tape** out);
uint32_t fhip_tape_term_tree(const fhip_tape* tape, uint32_t* words, uint32_t cap_ops);
uint32_t fhip_tape_term_choice_src(const fhip_tape* tape, uint32_t* src, uint32_t cap);

/* Function::simplify (eval/mod.rs:147-160; VmData::simplify vm/data.rs:123-318).
 * `choices` is one byte per choice op in evaluation order, values 1/2/3 = Left/Right/Both. */
fhip_status fhip_simplify(fhip_ctx* ctx, const fhip_tape* tape, const uint8_t* choices, uint32_t n_choices,
                          fhip_tape** child);

/* ---- evaluators (trait surface; batched over n samples) ----------------------------- */
/* TracingEvaluator<Data = Interval>::eval (eval/tracing.rs:26-64, vm/mod.rs:325-538).
 * vars: [n][n_vars][2] (lo,hi); out: [n][n_outputs][2]; choices: [n][choice_count] or NULL;
 * simplify: [n] (1 = a trace would be returned) or NULL.  n_vars may exceed the tape's. */
fhip_status fhip_interval_eval(fhip_ctx* ctx, const fhip_tape* tape, const float* vars, uint32_t n_vars, uint32_t n,
                               float* out, uint8_t* choices, uint8_t* simplify);
/* TracingEvaluator<Data = f32>::eval (vm/mod.rs:543-760).  vars: [n][n_vars]; out: [n][n_outputs] */
fhip_status fhip_point_eval(fhip_ctx* ctx, const fhip_tape* tape, const float* vars, uint32_t n_vars, uint32_t n,
                            float* out, uint8_t* choices, uint8_t* simplify);
/* BulkEvaluator<Data = f32>::eval (eval/bulk.rs:23-58, vm/mod.rs:794-1086).
 * vars[i] -> lens[i] floats (all lens must be equal, else MISMATCHED_SLICES);
 * out[o] -> n floats for each of the tape's outputs. */
fhip_status fhip_float_eval(fhip_ctx* ctx, const fhip_tape* tape, const float* const* vars, const uint32_t* lens,
                            uint32_t n_vars, float* const* out);
/* BulkEvaluator<Data = Grad>::eval (vm/mod.rs:1091-1397).  Grad = {v,dx,dy,dz} (types/grad.rs:4-13) */
fhip_status fhip_grad_eval(fhip_ctx* ctx, const fhip_tape* tape, const float* const* vars, const uint32_t* lens,
                           uint32_t n_vars, float* const* out);

/* ---- batched renders (the throughput path) ------------------------------------------ */
/* fidget_raster::pixel::{RenderConfig, EvalConfig} (fidget-raster/src/pixel.rs:27-57) */
typedef struct fhip_render2d_config {
    uint32_t width, height;
    const float* world_to_model;   /* row-major 3x3, NULL = identity */
    float z;
    int pixel_perfect;
    const uint32_t* tile_sizes;    /* NULL = RenderHints::tile_sizes_2d() of the HIP shape: {128,16}, as fidget-jit (fidget-jit/src/lib.rs:984-986):
                                    * 64 children per parent.  Fills carry the level they were decided at (pixel.rs:225-229), so images
                                    * compare bit for bit with the reference rendered with the same tile sizes */
    uint32_t n_tile_sizes;
    const uint64_t* var_keys;      /* ShapeVars<f32>: Var::V index -> value */
    const float* var_values;
    uint32_t n_vars;
    const int32_t* axis_slots;     /* VarMap slot of X, Y, Z (-1 = absent); NULL = use the tape's own map */
} fhip_render2d_config;
/* fidget_raster::voxel::{RenderConfig, EvalConfig} (fidget-raster/src/voxel.rs:25-54) */
typedef struct fhip_render3d_config {
    uint32_t width, height, depth;
    const float* world_to_model;   /* row-major 4x4, NULL = identity */
    const uint32_t* tile_sizes;    /* NULL = RenderHints::tile_sizes_3d() of the HIP shape: the root tile the VmShape hints give for this
                                    * image size, then fan-out 4^3: {128,32,8} - or {32,8} when the root level has few tiles (a small
                                    * image, a part of a frame, a model without z).  Any list TileSizes::new accepts
                                    * (render/mod.rs:181-251) is accepted; one the kernels cannot take as given (leaves other than 8,
                                    * fan-out above 64) is replaced by the library's: a 3D image does not depend on the tile sizes */
    uint32_t n_tile_sizes;
    const uint64_t* var_keys;
    const float* var_values;
    uint32_t n_vars;
    const int32_t* axis_slots;
} fhip_render3d_config;

/* fidget_raster::pixel::render (pixel.rs:452-492).  out: width*height RawDistancePixel (f32
 * bit patterns, pixel.rs:159-241); a device pointer when out_is_device != 0, in which case
 * the call is asynchronous on the context's stream. */
fhip_status fhip_render2d(fhip_ctx* ctx, const fhip_tape* tape, const fhip_render2d_config* cfg, float* out,
                          int out_is_device);
/* fidget_raster::voxel::render (voxel.rs:500-553).  out: width*height GeometryPixel
 * {f32 normal[3]; u32 depth} (voxel.rs:122-134).
 * Asynchronous renders (out_is_device) of one context are pipelined across frames: the context keeps four sets of device
 * buffers, and the coarse tile levels of a frame run on an internal stream beside the slabs of the frame before it.  For the
 * caller nothing changes: `out` is written on the context's stream, in call order; fhip_ctx_sync waits for every frame. */
fhip_status fhip_render3d(fhip_ctx* ctx, const fhip_tape* tape, const fhip_render3d_config* cfg, void* out,
                          int out_is_device);
/* Multi-GPU 3D, partition A: render only the root-tile columns whose index (x-major, lib.rs:116-123) satisfies
 * index % n_shards == shard, at full depth (front-to-back culling intact).  Other pixels are left {0,0,0,0}: every pixel is
 * produced by exactly one shard, so the partial images combine with an integer SUM of the raw words (or a gather). */
fhip_status fhip_render3d_shard(fhip_ctx* ctx, const fhip_tape* tape, const fhip_render3d_config* cfg, void* out,
                                int out_is_device, uint32_t shard, uint32_t n_shards);
/* Multi-GPU 3D, partition B (the octant split): block `index` = ix + split[0] * (iy + split[1] * iz) of a
 * split[0] x split[1] x split[2] division of the volume: blocks of root-tile columns in x and y, of z-slabs (root-tile
 * layers) in z; iz = split[2] - 1 is nearest the camera.  Pixels outside the block's columns stay {0,0,0,0}.  Blocks
 * that differ only in iz cover the same pixels and combine with fhip_merge_depth, front range first. */
fhip_status fhip_render3d_block(fhip_ctx* ctx, const fhip_tape* tape, const fhip_render3d_config* cfg, void* out,
                                int out_is_device, uint32_t index, const uint32_t split[3]);
/* The stitch rule of voxel.rs:527-550 across z ranges, in place on `front` (device pointers, n_pixels GeometryPixel each):
 * the larger depth wins, a tie keeps `front` (a hit there carries the normal; the equal depth on the other side is a
 * filled tile's z + T + 1, which has none), then depth >= image_depth - 1 -> (image_depth, [0, 0, 1]). */
fhip_status fhip_merge_depth(fhip_ctx* ctx, void* front, const void* back, uint64_t n_pixels, uint32_t image_depth);

/* ---- post-processing: fidget_raster::effects (fidget-raster/src/effects.rs) --------------------
 * The step right after a render; images are width*height arrays as the renders produce them.  With on_device != 0
 * every pointer is a device pointer and the call is asynchronous on the context's stream (the image a render just
 * left in HBM is consumed in place); otherwise host buffers. */
/* denoise_normals (effects.rs:17-36, denoise_pixel 252-326): GeometryPixel image -> GeometryPixel image */
fhip_status fhip_denoise_normals(fhip_ctx* ctx, const void* image, uint32_t width, uint32_t height, void* out, int on_device);
/* compute_ssao (effects.rs:73-95, compute_pixel_ssao 156-250): out = width*height f32, NaN where depth == 0.  The
 * reference draws the sampling kernel (3 x n_kernel, ssao_kernel 395-424) and the rotation noise (2 x n_noise,
 * ssao_noise 430-448) from rand::rng(); here the caller passes them: kernel[i*3 + {0,1,2}], noise[i*2 + {0,1}]. */
fhip_status fhip_compute_ssao(fhip_ctx* ctx, const void* image, uint32_t width, uint32_t height, uint32_t depth, const float* kernel,
                              uint32_t n_kernel, const float* noise, uint32_t n_noise, float* out, int on_device);
/* blur_ssao (effects.rs:98-115, compute_pixel_blur 329-392) */
fhip_status fhip_blur_ssao(fhip_ctx* ctx, const float* ssao, uint32_t width, uint32_t height, float* out, int on_device);
/* apply_shading (effects.rs:42-67, shade_pixel 118-153): ssao = blurred occlusion map or NULL; out = width*height*3 bytes */
fhip_status fhip_apply_shading(fhip_ctx* ctx, const void* image, uint32_t width, uint32_t height, uint32_t depth, const float* ssao,
                               uint8_t* out_rgb, int on_device);
/* RawDistancePixel image -> RGBA8: mode 0 to_rgba_bitmap (effects.rs:443-464), 1 the same with transparent = true,
 * 2 to_debug_bitmap (467-496), 3 to_rgba_distance (506-547) */
fhip_status fhip_to_rgba(fhip_ctx* ctx, const float* image, uint32_t width, uint32_t height, int mode, uint8_t* out_rgba, int on_device);

/* ---- meshing: the evaluation side of fidget_mesh::Octree::build (fidget-mesh/src/octree.rs) -----------------
 * Octree cells of [-1, 1]^3 to `depth` (Settings::depth), classified by interval evaluation (recurse, octree.rs:521-583), and
 * every ambiguous cell of the last level sampled as leaf() does (octree.rs:590-862): corner mask, Manifold Dual Contouring
 * edges, 4 x 16-point edge search, intersections, gradients, one QEF vertex per cell vertex.  What has no evaluation in it
 * (cell collapse, the dual walk) stays with the caller.  Leaf record (fhip_mesh_counts out[4] bytes, 528): f32 bounds[6]
 * (x.lo x.hi y.lo y.hi z.lo z.hi); u64 path (3 bits per level, leading 1); u32 mask, n_edges, n_verts, pad; u16 inter[12][3]
 * (+ 4 u16 pad); f32 pos[12][3]; f32 grad[12][4] (dx dy dz v); f32 vert[4][3]; f32 qef_err[4].  mask 0 / 255: the cell turned
 * out Empty / Full at its corners (n_edges = 0). */
typedef struct fhip_mesh fhip_mesh;
fhip_status fhip_mesh_sample(fhip_ctx* ctx, const fhip_tape* tape, uint32_t depth, const float* world_to_model, const int32_t* axis_slots,
                             const uint64_t* var_keys, const float* var_values, uint32_t n_vars, fhip_mesh** out);
/* fidget_mesh::Octree::build + Octree::walk_dual (octree.rs:48-68, 219-225; Settings: depth, world_to_model): fhip_mesh_sample, then
 * the octree assembled from the device's results - cell collapse (check_done / try_collapse, octree.rs:256-385) with the merged
 * Hermite data included - ON THE DEVICE, level by level, and the dual walk (dc.rs, builder.rs) on the device too: the recursion's calls
 * as level arrays in call order, MeshBuilder's numbering by first use through atomic minima and prefix sums (neither the leaf records
 * nor the octree leave HBM: the host receives the mesh) -> Mesh { vertices, triangles } (lib.rs:64-69): the cells, vertices and
 * triangles of the single-threaded recursion, in its order.  Tapes of 256 ops and more are simplified once on the way down
 * (octree.rs:546-553; option "mesh_simplify_min_ops", 0 = never): the mesh does not depend on it.  Context option "mesh_device_walk" 0:
 * the walk on the host's threads (independent sub-walks; FHIP_MESH_THREADS, default: all cores up to 32); "mesh_device_assembly" 0: the assembly on the
 * host's threads too (independent subtrees, as build_inner_mt octree.rs:94-210), from copies of the levels and the leaf records -
 * the path fhip_mesh_merge takes; same per-cell functions, same octree.  The leaf records are not kept with the mesh
 * (fhip_mesh_leaves after a build copies nothing). */
fhip_status fhip_mesh_build(fhip_ctx* ctx, const fhip_tape* tape, uint32_t depth, const float* world_to_model, const int32_t* axis_slots,
                            const uint64_t* var_keys, const float* var_values, uint32_t n_vars, fhip_mesh** out);
void fhip_mesh_vertices(const fhip_mesh* mesh, float* out);        /* counts[6] x 3 floats */
void fhip_mesh_triangles(const fhip_mesh* mesh, uint64_t* out);    /* counts[7] x 3 vertex indices */
/* The same arrays where the mesh holds them, without a copy (15 M triangles are 360 MB: a tenth of a second of a 0.35 s build) - what
 * `Mesh { vertices, triangles }` (fidget-mesh/src/mesh.rs) can borrow or take; valid until fhip_mesh_free. */
const float* fhip_mesh_vertices_ptr(const fhip_mesh* mesh);
const uint64_t* fhip_mesh_triangles_ptr(const fhip_mesh* mesh);
void fhip_mesh_free(fhip_mesh* mesh);
/* out = {cells interval-evaluated, Full, Empty, ambiguous cells at the leaf depth, bytes per leaf record, levels visited,
 *        mesh vertices, mesh triangles} */
void fhip_mesh_counts(const fhip_mesh* mesh, uint64_t out[8]);
void fhip_mesh_leaves(const fhip_mesh* mesh, void* out);
/* The build sharded by the root's octants, as Octree::build_inner_mt hands the root's eight children to its workers
 * (octree.rs:94-123) - here to up to eight GPUs.  fhip_mesh_sample_part runs the device side (cell classification, leaf
 * sampling) for the octants o with o * n_parts / 8 == part (8 parts: one octant each; 2 parts: the z halves); every part
 * evaluates the root cell itself.  Its results - per level the cells' classes and slots, and the leaf records - are written
 * as one flat buffer by fhip_mesh_part_export (fhip_mesh_part_bytes long), which is what travels between processes.
 * fhip_mesh_merge takes the buffers of ALL parts (parts[k] = part k), puts the level arrays together (build_inner_mt's index
 * remapping, octree.rs:176-195: slots of later parts shifted by the ambiguous cells before them), and runs octree assembly
 * (check_done on the merged tree, octree.rs:197-208) and the dual walk: the mesh is the one fhip_mesh_build gives on one GPU,
 * vertex for vertex and triangle for triangle.  The buffers (8-byte aligned) are only read during the call.  A merge needs no
 * device; with a context it reuses the context's host-side caches. */
fhip_status fhip_mesh_sample_part(fhip_ctx* ctx, const fhip_tape* tape, uint32_t depth, const float* world_to_model, const int32_t* axis_slots,
                                  const uint64_t* var_keys, const float* var_values, uint32_t n_vars, uint32_t part, uint32_t n_parts, fhip_mesh** out);
uint64_t fhip_mesh_part_bytes(const fhip_mesh* mesh);
void fhip_mesh_part_export(const fhip_mesh* mesh, void* out);
fhip_status fhip_mesh_merge(fhip_ctx* ctx, const void* const* parts, const uint64_t* part_bytes, uint32_t n_parts, const float* world_to_model,
                            fhip_mesh** out);

/* ---- profiling ----------------------------------------------------------------------- */
/* When enabled, every kernel launch of a render is bracketed by HIP events on the context's
 * stream; fhip_profile_read returns per-kernel-class totals of the last render. */
enum { FHIP_K_TILES = 0, FHIP_K_POINTS = 1, FHIP_K_NORMALS = 2, FHIP_K_OTHER = 3, FHIP_K_COUNT = 4 };
void fhip_profile_enable(fhip_ctx* ctx, int on);
fhip_status fhip_profile_read(fhip_ctx* ctx, double ms[4], uint32_t launches[4]);
/* ... and per assembly kernel, each launch bracketed by its own pair of events:
 * index 0 fh_columns, 1 / 2 fh_float_eval_{16x4, 32x2}, 3 fh_tiles, 4 fh_prune1 */
fhip_status fhip_profile_read_kernels(fhip_ctx* ctx, double ms[8], uint32_t launches[8]);
/* Counters of the last render: [0] arena ops used (peak), [1] arena overflows, [2] leaves of the last slab, [3] queue overflows,
 * [4..5] queue entries per tile level below the root; [6] (context total) frames whose tile stage ran on the HIP C++ kernels without a
 * switch asking for it (tapes beyond the assembly kernels' register files); [7] (context total) 3D frames whose tile_sizes were valid but not a list the
 * kernels take - leaves other than 8^3, a fan-out above 64 - and were rendered with the library's own list (same image) */
fhip_status fhip_render_counters(fhip_ctx* ctx, uint64_t out[8]);

/* Diagnostics (wave statistics, arena / queue dumps, micro-benchmarks, the tape-group plans) are declared in
 * fidget_hip_debug.h: they are not part of the surface a binding needs. */

/* ---- host mirror of fidget_core::Context (no Rust toolchain here; tests + demos) ------ */
/* Opcode numbers = declaration order of UnaryOpcode / BinaryOpcode (context/op.rs:11-48). */
fhip_graph* fhip_graph_new(void);
void fhip_graph_free(fhip_graph* g);
uint32_t fhip_graph_len(const fhip_graph* g);
uint32_t fhip_graph_var(fhip_graph* g, int axis_or_3, uint64_t index); /* 0 X, 1 Y, 2 Z, 3 Var::V(index) */
uint32_t fhip_graph_constant(fhip_graph* g, float v);
uint32_t fhip_graph_unary(fhip_graph* g, int opcode, uint32_t a);              /* 0xFFFFFFFF = BadNode */
uint32_t fhip_graph_binary(fhip_graph* g, int opcode, uint32_t a, uint32_t b);
uint32_t fhip_graph_from_text(fhip_graph* g, const char* text);               /* Context::from_text */
/* MathFunction::new (eval/mod.rs:203-208) */
fhip_status fhip_tape_from_graph(fhip_ctx* ctx, const fhip_graph* g, const uint32_t* roots, uint32_t n_roots,
                                 fhip_tape** out);
/* VarMap of a graph-built tape: slot of X/Y/Z (axis 0..2) or of Var::V(index); -1 if absent */
int fhip_tape_axis_slot(const fhip_tape* tape, int axis);
int fhip_tape_var_slot(const fhip_tape* tape, uint64_t index);

/* RegionSize::screen_to_world (render/region.rs:87-108); n = 2 -> 3x3, n = 3 -> 4x4, row major */
void fhip_screen_to_world(const uint32_t* size, int n, float* out);

#ifdef __cplusplus
}
#endif
#endif
