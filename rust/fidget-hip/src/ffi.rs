//! The raw binding of `include/fidget_hip.h` (the C ABI of libfidget_hip.so): every function the crate calls, with the header's
//! argument order and types.  `tests/test_rust_binding.py` in the fidget-hip repository parses this file and the header and fails
//! on any drift (a function missing on either side of the list below, another arity, another pointer / integer type, another
//! struct field order).
#![allow(non_camel_case_types)]
#![allow(missing_docs)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct fhip_ctx   { _p: [u8; 0] }
#[repr(C)] pub struct fhip_tape  { _p: [u8; 0] }
#[repr(C)] pub struct fhip_mesh  { _p: [u8; 0] }

pub type fhip_status = c_int;           // 0 = OK, see fidget_hip.h
pub const FHIP_ERR_BAD_VAR_SLICE: c_int = 1;     // -> TracingEvalError / BulkEvalError::BadVarSlice
pub const FHIP_ERR_MISMATCHED_SLICES: c_int = 2; // -> BulkEvalError::MismatchedSlices
pub const FHIP_ERR_BAD_CHOICE_SLICE: c_int = 3;  // -> BadTrace::BadChoiceSlice

#[repr(C)]
pub struct fhip_render3d_config {
    pub width: u32, pub height: u32, pub depth: u32,
    pub world_to_model: *const f32,          // row-major 4x4 or null
    pub tile_sizes: *const u32, pub n_tile_sizes: u32,
    pub var_keys: *const u64, pub var_values: *const f32, pub n_vars: u32,
    pub axis_slots: *const i32,              // VarMap slots of X, Y, Z (-1 = absent)
}
#[repr(C)]
pub struct fhip_render2d_config {
    pub width: u32, pub height: u32,
    pub world_to_model: *const f32,          // row-major 3x3 or null
    pub z: f32, pub pixel_perfect: c_int,
    pub tile_sizes: *const u32, pub n_tile_sizes: u32,
    pub var_keys: *const u64, pub var_values: *const f32, pub n_vars: u32,
    pub axis_slots: *const i32,
}

#[link(name = "fidget_hip")]
extern "C" {
    pub fn fhip_ctx_create(device: c_int, stream: *mut c_void, out: *mut *mut fhip_ctx) -> fhip_status;
    pub fn fhip_ctx_destroy(ctx: *mut fhip_ctx);
    pub fn fhip_libm_probe(msg: *mut c_char, cap: usize) -> c_int;            // this host's libm against the routines the device restates
    pub fn fhip_last_error(ctx: *const fhip_ctx) -> *const c_char;
    pub fn fhip_cancel(ctx: *mut fhip_ctx);
    pub fn fhip_cancel_reset(ctx: *mut fhip_ctx);
    pub fn fhip_cancel_watch(ctx: *mut fhip_ctx, flag: *const c_void);    // the caller's one-byte cancel flag (CancelToken::into_raw), null: none
    pub fn fhip_ctx_sync(ctx: *mut fhip_ctx) -> fhip_status;     // waits for the asynchronous renders, reports their overflow flags
    pub fn fhip_ctx_trim(ctx: *mut fhip_ctx) -> fhip_status;     // caches kept for speed alone (mesh leaf records, frame lanes) go back
    pub fn fhip_ctx_reserve_arena(ctx: *mut fhip_ctx, megabytes: usize) -> fhip_status;     // a sequence of heavier frames to come: arenas sized ahead
    pub fn fhip_ctx_set_option(ctx: *mut fhip_ctx, name: *const c_char, value: c_int) -> fhip_status;   // behaviour switches: see below
    pub fn fhip_ctx_get_option(ctx: *const fhip_ctx, name: *const c_char, value: *mut c_int) -> fhip_status;

    pub fn fhip_tape_from_bytecode(ctx: *mut fhip_ctx, words: *const u32, n_words: usize,
                                   out: *mut *mut fhip_tape) -> fhip_status;
    pub fn fhip_tape_free(tape: *mut fhip_tape);
    pub fn fhip_tape_len(tape: *const fhip_tape) -> u32;
    pub fn fhip_tape_choice_count(tape: *const fhip_tape) -> u32;
    // RegTape::new::<N> + Bytecode::new of a tape, by the reference's own allocator (LRU eviction, Load / Store spills) on the host:
    // for callers that want VmData<N>'s view of a simplified tape back (len with spills, iter_asm, wire format); info = [len,
    // slot_count, reg_count, mem_count]
    pub fn fhip_tape_reg_tape(tape: *const fhip_tape, n_regs: u32, reg_ops: *mut u32, cap_ops: u32, words: *mut u32, cap_words: u32,
                              info: *mut u32) -> fhip_status;
    pub fn fhip_simplify(ctx: *mut fhip_ctx, tape: *const fhip_tape, choices: *const u8, n: u32,
                         child: *mut *mut fhip_tape) -> fhip_status;

    pub fn fhip_interval_eval(ctx: *mut fhip_ctx, tape: *const fhip_tape, vars: *const f32, n_vars: u32, n: u32,
                              out: *mut f32, choices: *mut u8, simplify: *mut u8) -> fhip_status;
    pub fn fhip_point_eval(ctx: *mut fhip_ctx, tape: *const fhip_tape, vars: *const f32, n_vars: u32, n: u32,
                           out: *mut f32, choices: *mut u8, simplify: *mut u8) -> fhip_status;
    pub fn fhip_float_eval(ctx: *mut fhip_ctx, tape: *const fhip_tape, vars: *const *const f32, lens: *const u32,
                           n_vars: u32, out: *const *mut f32) -> fhip_status;
    pub fn fhip_grad_eval(ctx: *mut fhip_ctx, tape: *const fhip_tape, vars: *const *const f32, lens: *const u32,
                          n_vars: u32, out: *const *mut f32) -> fhip_status;

    pub fn fhip_render2d(ctx: *mut fhip_ctx, tape: *const fhip_tape, cfg: *const fhip_render2d_config,
                         out: *mut f32, out_is_device: c_int) -> fhip_status;
    pub fn fhip_render3d(ctx: *mut fhip_ctx, tape: *const fhip_tape, cfg: *const fhip_render3d_config,
                         out: *mut c_void, out_is_device: c_int) -> fhip_status;
    pub fn fhip_render3d_shard(ctx: *mut fhip_ctx, tape: *const fhip_tape, cfg: *const fhip_render3d_config,
                               out: *mut c_void, out_is_device: c_int, shard: u32, n_shards: u32) -> fhip_status;
    pub fn fhip_render3d_block(ctx: *mut fhip_ctx, tape: *const fhip_tape, cfg: *const fhip_render3d_config,
                               out: *mut c_void, out_is_device: c_int, index: u32, split: *const u32) -> fhip_status;
    pub fn fhip_merge_depth(ctx: *mut fhip_ctx, front: *mut c_void, back: *const c_void, n_pixels: u64, image_depth: u32) -> fhip_status;

    // fidget_raster::effects on the image a render left in HBM (on_device = 1) or on host buffers
    pub fn fhip_denoise_normals(ctx: *mut fhip_ctx, image: *const c_void, w: u32, h: u32, out: *mut c_void, on_device: c_int) -> fhip_status;
    pub fn fhip_compute_ssao(ctx: *mut fhip_ctx, image: *const c_void, w: u32, h: u32, depth: u32, kernel: *const f32, n_kernel: u32,
                             noise: *const f32, n_noise: u32, out: *mut f32, on_device: c_int) -> fhip_status;
    pub fn fhip_blur_ssao(ctx: *mut fhip_ctx, ssao: *const f32, w: u32, h: u32, out: *mut f32, on_device: c_int) -> fhip_status;
    pub fn fhip_apply_shading(ctx: *mut fhip_ctx, image: *const c_void, w: u32, h: u32, depth: u32, ssao: *const f32,
                              out_rgb: *mut u8, on_device: c_int) -> fhip_status;
    pub fn fhip_to_rgba(ctx: *mut fhip_ctx, image: *const f32, w: u32, h: u32, mode: c_int, out_rgba: *mut u8, on_device: c_int) -> fhip_status;

    // fidget_mesh::Octree::build + walk_dual
    pub fn fhip_mesh_build(ctx: *mut fhip_ctx, tape: *const fhip_tape, depth: u32, world_to_model: *const f32, axis_slots: *const i32,
                           var_keys: *const u64, var_values: *const f32, n_vars: u32, out: *mut *mut fhip_mesh) -> fhip_status;
    pub fn fhip_mesh_counts(mesh: *const fhip_mesh, out: *mut u64);          // [6] vertices, [7] triangles
    pub fn fhip_mesh_vertices(mesh: *const fhip_mesh, out: *mut f32);
    pub fn fhip_mesh_triangles(mesh: *const fhip_mesh, out: *mut u64);
    pub fn fhip_mesh_vertices_ptr(mesh: *const fhip_mesh) -> *const f32;       // the arrays where the mesh holds them (valid until fhip_mesh_free)
    pub fn fhip_mesh_triangles_ptr(mesh: *const fhip_mesh) -> *const u64;
    pub fn fhip_mesh_free(mesh: *mut fhip_mesh);
    // the build sharded by the root's octants (Octree::build_inner_mt across GPUs): a part per process, merged in one
    pub fn fhip_mesh_sample_part(ctx: *mut fhip_ctx, tape: *const fhip_tape, depth: u32, world_to_model: *const f32, axis_slots: *const i32,
                                 var_keys: *const u64, var_values: *const f32, n_vars: u32, part: u32, n_parts: u32,
                                 out: *mut *mut fhip_mesh) -> fhip_status;
    pub fn fhip_mesh_part_bytes(mesh: *const fhip_mesh) -> u64;
    pub fn fhip_mesh_part_export(mesh: *const fhip_mesh, out: *mut c_void);
    pub fn fhip_mesh_merge(ctx: *mut fhip_ctx, parts: *const *const c_void, part_bytes: *const u64, n_parts: u32,
                           world_to_model: *const f32, out: *mut *mut fhip_mesh) -> fhip_status;
}
