// libfidget_hip.so: C ABI (include/fidget_hip.h) + host driver of the render pipeline.
// Single translation unit: the kernels are included so that one `hipcc -shared` builds
// everything for gfx950.
#include <hip/hip_runtime.h>
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fidget_hip.h"
#include "../../include/fidget_hip_debug.h"
#include "host_graph.hpp"
#include "host_regtape.hpp"
#include "kernels.hip"
#include "effects.hip"
#include "mesh.hip"
#include "prune2.hip"
#include "host_mesh.hpp"

#define FH_LDS_MAX 163840  // 160 KiB per workgroup on gfx950

// ----------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

static std::atomic<uint64_t> g_tape_serial{1};
struct fhip_tape {
    const uint64_t serial = g_tape_serial.fetch_add(1);   // identity for "these tapes are already in the arena"
    fh::HostTape t;
    mutable uint64_t* d_ops = nullptr;  // uploaded on first device use (tape construction is host-only)
    // tape parallelism (host_graph.hpp split_root): when the root is a min / max of many parts, the
    // same function as `groups.size()` independent tapes whose outputs combine with `group_op`
    std::vector<fh::HostTape> groups;
    int group_op = -1;
    // ... and the renderer's form of it (plan_terms): groups that output the root tree's terms, the
    // tree as a small program over them, and where every choice of the full tape is recorded
    fh::TermPlan plan;
    std::vector<fh::HostTape> tgroups;
    mutable FhTopOp* d_top = nullptr;
    mutable uint32_t* d_chsrc = nullptr;
    mutable uint64_t* d_links = nullptr;   // links of the full tape (host_graph.hpp compute_links) for the linked prune, when it qualifies
    mutable uint64_t* d_ctab = nullptr;    // ... and per choice its op's operands and index
    mutable bool links_tried = false;
    // A tape is immutable and may be shared by contexts on different threads (one context per thread, as the
    // reference's workers): its lazily created device copies are made under this lock, on the device of the first
    // context that needs them (HIP allocations are visible to every device of the process with peer access; a tape
    // used from several devices should be built per device)
    mutable std::mutex upload_lock;
    mutable int device = -1;
};
struct fhip_graph {
    fh::Graph g;
};

// The assembly interpreters (gen_interp.py -> interp_gfx950.co), embedded at build time
#ifndef __HIP_DEVICE_COMPILE__
__asm__(".section .rodata\n.global fh_interp_co\n.p2align 6\nfh_interp_co:\n.incbin \"" FH_INTERP_CO "\"\n.previous\n");
#endif
extern "C" const char fh_interp_co[];
enum { FH_ASM_COLUMNS = 0, FH_ASM_FLOAT_16x4, FH_ASM_FLOAT_32x2, FH_ASM_TILES, FH_ASM_PRUNE1, FH_ASM_TILES_V32, FH_ASM_TILES_V64, FH_ASM_PROBE, FH_ASM_UBENCH, FH_ASM_COLUMNS_T, FH_ASM_NORMALS, FH_ASM_NORMALS_T, FH_ASM_TILES_T, FH_ASM_TILES_V32_T, FH_ASM_TILES_V64_T, FH_ASM_FLOAT_16x4_T, FH_ASM_FLOAT_32x2_T, FH_ASM_COUNT };
static const char* const FH_ASM_NAMES[FH_ASM_COUNT] = {"fh_columns", "fh_float_eval_16x4", "fh_float_eval_32x2", "fh_tiles", "fh_prune1",
                                                       "fh_tiles_v32", "fh_tiles_v64", "fh_probe", "fh_ubench", "fh_columns_t", "fh_normals", "fh_normals_t", "fh_tiles_t", "fh_tiles_v32_t", "fh_tiles_v64_t", "fh_float_eval_16x4_t", "fh_float_eval_32x2_t"};
// register-file shapes of the VGPR tile kernels (gen_tilesv.py): registers, choices
static const uint32_t V32_REGS = 32, V32_CHOICES = 256, V64_REGS = 64, V64_CHOICES = 512;

// Behaviour switches of a context - diagnostics and tuning, none is needed in normal use.  They are part of the context, not of
// the process: read ONCE from the environment when the context is created (FHIP_<NAME IN CAPITALS>, for runs of unmodified
// programs under a switch) and changed afterwards only through fhip_ctx_set_option(ctx, "<name>", value) - nothing in a
// render's launch path looks at the environment.  name, default; DESIGN.md section 5 says what each one selects.
#define FH_OPTION_LIST(X)                                                                                                       \
    X(no_asm, 0) X(no_split, 0) X(probe, 0) X(no_pipeline, 0) X(slab_contexts, 4) X(no_frame_pipeline, 0) X(arena_mb, 4096)     \
    X(vm_tiles, 0) X(no_columns_t, 0) X(no_split_2d, 0) X(no_asm_tiles, 0) X(prune1_levels, 1) X(no_prune1, 0)                  \
    X(no_tape_groups, 0) X(stats, 0) X(one_each_tiles, 0) X(pipe_serial, 0) X(no_tiles_v, 0) X(no_both_lists, 0)               \
    X(v32_waves, 16) X(v64_waves, 8) X(v64_slab_waves, 128) X(no_mid, 0) X(push_waves, 2) X(no_column_inv, 0) X(no_zrep, 0)     \
    X(debug_zfill, 0) X(old_pyr, 0) X(no_slab_begin, 0) X(tail_stream, 1) X(col_waves, 0) X(col_blkl, 2) X(l1_split, 1) X(prune2, 1) X(prune2_l1, 0) X(no_asm_normals, 0) X(no_asm_tiles_t, 0) X(normals_waves, 8) X(prune2_probe_level, 0) X(mesh_device_assembly, 1) X(slab_layers, 4) X(l1_on_side, 1) X(tiles_stream, 2) X(frame_sets, 3)                        \
    /* fixed when the context is created (they decide which streams exist): environment only */                                 \
    X(leaf_streams, 1) X(pre_priority, 0)
struct FhOptions {
#define X(name, dflt) int name = dflt;
    FH_OPTION_LIST(X)
#undef X
};
struct FhOptionEntry { const char* name; int FhOptions::*field; };
static const FhOptionEntry FH_OPTION_TABLE[] = {
#define X(name, dflt) {#name, &FhOptions::name},
    FH_OPTION_LIST(X)
#undef X
};
static void options_from_env(FhOptions& o) {
    for (const FhOptionEntry& e : FH_OPTION_TABLE) {
        std::string var = "FHIP_";
        for (const char* c = e.name; *c; c++) var += (char)toupper((unsigned char)*c);
        if (const char* v = getenv(var.c_str())) o.*(e.field) = *v ? atoi(v) : 1;    // (set but empty counts as 1)
    }
}

// Everything one frame of a render owns on the device.  A context holds two sets: an asynchronous 3D render takes the set
// the previous frame did not use, so that its coarse levels (which keep a few hundred waves busy for most of a millisecond)
// run on a stream of their own beside the previous frame's slabs (frame pipelining, FHIP_NO_FRAME_PIPELINE=1 turns it off).
struct FrameBufs {
    DevBuf state, arena, leaves, leaf_table, zbuf, normals, fp_lists, mind, squeue, slots[2], leaves_b, leaf_table_b, fp_lists_b, chw[2], tvals, topch, chwr, gscratch;
    DevBuf queue[FH_MAX_LEVELS];
    uint32_t frame_stamp = 0;       // FhRenderState::frame_stamp of the last frame prepared
    uint64_t resident_serial = 0;   // the root (and group) tapes at the bottom of the arena belong to this tape
    uint32_t resident_groups = 0;
    uint32_t forked = 0;            // slab contexts of the last 3D frame of this set (0: not pipelined)
    bool async_pending = false;     // the last render of this set left its result on the device: its overflow flags have not been read yet
    hipEvent_t ev_done = nullptr;   // recorded when the last frame of this set has been queued completely
    bool ev_done_valid = false;
    void release_all() {
        DevBuf* bufs[] = {&state, &arena, &leaves, &leaf_table, &zbuf, &normals, &fp_lists, &mind, &squeue, &slots[0], &slots[1],
                          &leaves_b, &leaf_table_b, &fp_lists_b, &chw[0], &chw[1], &tvals, &topch, &chwr, &gscratch};
        for (DevBuf* b : bufs) b->release();
        for (auto& q : queue) q.release();
        if (ev_done) (void)hipEventDestroy(ev_done);
        ev_done = nullptr;
    }
};
struct fhip_ctx : FrameBufs {
    FhOptions opt;                  // behaviour switches (FH_OPTION_LIST): environment at creation, fhip_ctx_set_option later
    // the sets of the frames before the current one: a frame takes the set used longest ago (ring of 1 + FH_EXTRA_SETS).  Three
    // sets: one frame alone takes ~1.5 ms from its first coarse-level kernel to its image, so with two sets - a set is free
    // again when its frame is complete - no more than two frames per 1.5 ms could ever be under way (a fourth set: measured, no gain)
#define FH_EXTRA_SETS 2
    FrameBufs others[FH_EXTRA_SETS];
    uint32_t extra_sets = FH_EXTRA_SETS;     // option frame_sets - 1
    bool frame_pipeline = true;
    hipStream_t stream_pre = nullptr;   // coarse levels of a pipelined frame
    hipStream_t stream_leaf2 = nullptr; // FHIP_LEAF_STREAMS=2 (diagnostics): the leaf kernels of odd slabs
    hipEvent_t ev_rest_fork = nullptr, ev_rest_join = nullptr;
    hipEvent_t ev_pre = nullptr, ev_l0 = nullptr;
    hipModule_t asm_mod = nullptr;
    hipFunction_t asm_fn[FH_ASM_COUNT] = {};
    bool use_asm = true;  // FHIP_NO_ASM=1 keeps everything on the C++ kernels (diagnostics)
    bool probe = false;     // FHIP_PROBE=1: per-phase clocks in fh_tiles (slows it down; tools/wave_stats.py)
    bool use_split = true;  // FHIP_NO_SPLIT=1: monolithic k_tiles for the 3D tile stage (diagnostics)
    // 3D: the tile stage of slab k+1 runs on a second stream while slab k's leaves are evaluated
    bool use_pipeline = true;  // FHIP_NO_PIPELINE=1 serialises the slabs on one stream (diagnostics)
    hipStream_t stream2 = nullptr, stream3 = nullptr;
    std::vector<hipEvent_t> ev_tiles, ev_leaves, ev_aux;
    hipEvent_t ev_fork = nullptr;
    FhRenderState last_state_b;
    uint32_t slab_contexts = 4;   // FHIP_SLAB_CONTEXTS (2 .. 4): how far the tile chain may run ahead of the leaf chain (measured: 2.03 / 1.60 / 1.55 ms per frame with 2 / 3 / 4)
    int device = 0;
    hipStream_t stream = nullptr;
    int n_cu = 256;
    std::string err;
    bool launch_failed = false;     // an assembly kernel launch of the current frame failed (reported when the frame has been queued)
    std::atomic<int> cancelled{0};
    DevBuf tmp_out, io_a, io_b, io_c, io_d, io_e;
    DevBuf sticky;      // one word: a queue overflow of ANY asynchronous frame since the last fhip_ctx_sync (k_finish3d latches it)
    struct Staging { void* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool used = false; } staging[8];   // pinned (upload_frame)
    void* mesh_pinned = nullptr;      // fhip_mesh_build: the leaf records' landing area on the host, kept between calls (pinning 17 GB takes over a second)
    size_t mesh_pinned_cap = 0;
    // ... and the two largest host-side temporaries of the assembly, kept for the same reason (fresh memory of that size is
    // faulted in page by page and handed back page by page): the octree's cell / vertex arrays and the dual walk's first-use table
    void* mesh_octree_cache = nullptr;       // fhmesh::Octree*
    uint32_t* mesh_first = nullptr;
    size_t mesh_first_cap = 0;
    uint32_t staging_next = 0;
    size_t arena_bytes = (size_t)4 << 30;  // tape arena (FHIP_ARENA_MB overrides)
    bool profiling = false;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> prof_events;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> asm_events;   // ... and per assembly kernel launch
    FhRenderState last_state;
    bool have_last_state = false;
};

// The cached forms of some options (what the rest of the driver reads)
static void apply_options(fhip_ctx* c) {
    c->use_asm = c->opt.no_asm == 0;
    c->use_split = c->opt.no_split == 0;
    c->probe = c->opt.probe != 0;
    c->use_pipeline = c->opt.no_pipeline == 0;
    c->frame_pipeline = c->opt.no_frame_pipeline == 0;
    c->slab_contexts = (uint32_t)std::min(4, std::max(2, c->opt.slab_contexts));
    c->arena_bytes = (size_t)std::max(1, c->opt.arena_mb) << 20;
    c->extra_sets = (uint32_t)std::min(FH_EXTRA_SETS, std::max(1, c->opt.frame_sets - 1));
}

static fhip_status finish_render(fhip_ctx* ctx);
static void mesh_cache_release(void* octree);      // (defined with the mesh code)
static fhip_status fail(fhip_ctx* ctx, fhip_status s, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return s;
}
#define HIP_TRY(ctx, call)                                                                                       \
    do {                                                                                                         \
        hipError_t e_ = (call);                                                                                  \
        if (e_ != hipSuccess)                                                                                    \
            return fail(ctx, FHIP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));                   \
    } while (0)

template <class F>
static void launch(fhip_ctx* ctx, int klass, F&& f) {
    if (ctx->profiling) {
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        (void)hipEventRecord(a, ctx->stream);
        f();
        (void)hipEventRecord(b, ctx->stream);
        ctx->prof_events.push_back({klass, {a, b}});
    } else {
        f();
    }
}

extern "C" {

// ---- context ---------------------------------------------------------------------------
fhip_status fhip_ctx_create(int device, void* stream, fhip_ctx** out) {
    if (!out) return FHIP_ERR_BAD_TAPE;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device >= count) return FHIP_ERR_HIP;
    fhip_ctx* c = new fhip_ctx();
    c->device = device;
    c->stream = (hipStream_t)stream;
    if (hipSetDevice(device) != hipSuccess) { delete c; return FHIP_ERR_HIP; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
    options_from_env(c->opt);
    apply_options(c);
    // allow the full 160 KiB of LDS for the interpreters' register files
    const void* fns[] = {(const void*)k_eval_f32<false>, (const void*)k_eval_interval<false>, (const void*)k_eval_grad<false>,
                         (const void*)k_tiles<false, false, true, 16>, (const void*)k_tiles<false, true, true, 16>,
                         (const void*)k_tiles<true, false, true, 16>, (const void*)k_tiles<true, true, true, 16>,
                         (const void*)k_tiles<false, false, true, 64>, (const void*)k_tiles<false, true, true, 64>,
                         (const void*)k_tiles<true, false, true, 64>, (const void*)k_tiles<true, true, true, 64>,
                         (const void*)k_pixels2d<0, false>, (const void*)k_pixels2d<0, true>,
                         (const void*)k_leaves3d<2, 0, 1, false>, (const void*)k_leaves3d<2, 0, 1, true>,
                         (const void*)k_normals3d<false, true>, (const void*)k_normals3d<true, true>};
    for (const void* f : fns) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS_MAX);
    if (hipModuleLoadData(&c->asm_mod, fh_interp_co) != hipSuccess) { delete c; return FHIP_ERR_HIP; }
    for (int i = 0; i < FH_ASM_COUNT; i++)
        if (hipModuleGetFunction(&c->asm_fn[i], c->asm_mod, FH_ASM_NAMES[i]) != hipSuccess) { delete c; return FHIP_ERR_HIP; }
    (void)hipFuncSetAttribute((const void*)c->asm_fn[FH_ASM_TILES], hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS_MAX);
    (void)hipFuncSetAttribute((const void*)c->asm_fn[FH_ASM_TILES_T], hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS_MAX);
    (void)hipGetLastError();
    {   // the side stream carries the (latency-bound) tile stage of the next slab: highest priority
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, hi) != hipSuccess) { delete c; return FHIP_ERR_HIP; }
    }
    (void)hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&c->ev_pre, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&c->ev_l0, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&c->ev_rest_fork, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&c->ev_rest_join, hipEventDisableTiming);
    if (c->opt.leaf_streams == 2) (void)hipStreamCreateWithFlags(&c->stream_leaf2, hipStreamNonBlocking);
    (void)hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming);
    for (auto& o : c->others) (void)hipEventCreateWithFlags(&o.ev_done, hipEventDisableTiming);
    if (hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking) != hipSuccess) { delete c; return FHIP_ERR_HIP; }
    {   // FHIP_PRE_PRIORITY: 0 default, 1 lowest, 2 highest (diagnostics)
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        const int pp = c->opt.pre_priority;
        const hipError_t e = pp == 0 ? hipStreamCreateWithFlags(&c->stream_pre, hipStreamNonBlocking)
                                     : hipStreamCreateWithPriority(&c->stream_pre, hipStreamNonBlocking, pp == 1 ? lo : hi);
        if (e != hipSuccess) { delete c; return FHIP_ERR_HIP; }
    }
    if (c->sticky.ensure(256) != hipSuccess || hipMemset(c->sticky.p, 0, 256) != hipSuccess) { delete c; return FHIP_ERR_HIP; }
    c->ev_tiles.resize(FH_MAX_SLABS); c->ev_leaves.resize(FH_MAX_SLABS); c->ev_aux.resize(FH_MAX_SLABS);
    for (int i = 0; i < FH_MAX_SLABS; i++) {
        (void)hipEventCreateWithFlags(&c->ev_tiles[i], hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&c->ev_leaves[i], hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&c->ev_aux[i], hipEventDisableTiming);
    }
    (void)hipFuncSetAttribute((const void*)k_prune2, hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS_MAX);
    {
        const void* fb[] = {(const void*)k_teval3d<false, true>, (const void*)k_teval3d<true, true>};
        for (const void* f : fb) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS_MAX);
    }
    *out = c;
    return FHIP_OK;
}
void fhip_ctx_destroy(fhip_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->stream_pre) (void)hipStreamSynchronize(c->stream_pre);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    (void)hipStreamSynchronize(c->stream);
    DevBuf* bufs[] = {&c->tmp_out, &c->io_a, &c->io_b, &c->io_c, &c->io_d, &c->io_e, &c->sticky};
    for (DevBuf* b : bufs) b->release();
    c->release_all();
    for (auto& o : c->others) o.release_all();
    if (c->stream_pre) (void)hipStreamDestroy(c->stream_pre);
    if (c->stream_leaf2) { (void)hipStreamSynchronize(c->stream_leaf2); (void)hipStreamDestroy(c->stream_leaf2); }
    if (c->ev_rest_fork) (void)hipEventDestroy(c->ev_rest_fork);
    if (c->ev_rest_join) (void)hipEventDestroy(c->ev_rest_join);
    if (c->ev_pre) (void)hipEventDestroy(c->ev_pre);
    if (c->ev_l0) (void)hipEventDestroy(c->ev_l0);
    for (auto& sg : c->staging) { if (sg.p) (void)hipHostFree(sg.p); if (sg.ev) (void)hipEventDestroy(sg.ev); }
    if (c->mesh_pinned) (void)hipHostFree(c->mesh_pinned);
    mesh_cache_release(c->mesh_octree_cache);
    free(c->mesh_first);
    if (c->asm_mod) (void)hipModuleUnload(c->asm_mod);
    if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
    if (c->stream3) { (void)hipStreamSynchronize(c->stream3); (void)hipStreamDestroy(c->stream3); }
    for (hipEvent_t e : c->ev_aux) (void)hipEventDestroy(e);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (hipEvent_t e : c->ev_tiles) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_leaves) (void)hipEventDestroy(e);
    for (auto& e : c->prof_events) { (void)hipEventDestroy(e.second.first); (void)hipEventDestroy(e.second.second); }
    for (auto& e : c->asm_events) { (void)hipEventDestroy(e.second.first); (void)hipEventDestroy(e.second.second); }
    delete c;
}
const char* fhip_last_error(const fhip_ctx* c) { return c ? c->err.c_str() : "no context"; }
// Waits for everything queued on the context.  An asynchronous render (out_is_device) cannot report what only the
// device knows when it returns: its queue-overflow flag (the queues are sized to exact upper bounds, so this would be
// a bug, not a condition) is read here.
fhip_status fhip_ctx_sync(fhip_ctx* c) {
    (void)hipSetDevice(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    fhip_status st = FHIP_OK;
    for (auto& o : c->others)
        if (o.async_pending) {      // the frames before the last one (frame pipelining): same check, then back to the last frame's set
            std::swap(static_cast<FrameBufs&>(*c), o);
            c->async_pending = false;
            const fhip_status s1 = finish_render(c);
            if (st == FHIP_OK) st = s1;
            std::swap(static_cast<FrameBufs&>(*c), o);
        }
    if (c->async_pending) {
        c->async_pending = false;
        const fhip_status s2 = finish_render(c);
        if (st == FHIP_OK) st = s2;
    }
    // frames older than the last two (their buffer sets have been re-used since): the flag every frame's last kernel latches
    uint32_t sticky = 0;
    HIP_TRY(c, hipMemcpy(&sticky, c->sticky.p, 4, hipMemcpyDeviceToHost));
    if (sticky) {
        HIP_TRY(c, hipMemset(c->sticky.p, 0, 4));
        if (st == FHIP_OK) st = fail(c, FHIP_ERR_OVERFLOW, "device work queue overflow in an earlier asynchronous frame");
    }
    return st;
}
void fhip_cancel(fhip_ctx* c) { c->cancelled.store(1); }
void fhip_cancel_reset(fhip_ctx* c) { c->cancelled.store(0); }
// Behaviour switches (FH_OPTION_LIST above).  Waits for the frames in flight first: a switch never changes under a frame.
fhip_status fhip_ctx_set_option(fhip_ctx* c, const char* name, int value) {
    if (!c || !name) return FHIP_ERR_UNSUPPORTED;
    if (!strcmp(name, "leaf_streams") || !strcmp(name, "pre_priority")) return fail(c, FHIP_ERR_UNSUPPORTED, std::string(name) + " is fixed when the context is created");
    for (const FhOptionEntry& e : FH_OPTION_TABLE)
        if (!strcmp(e.name, name)) {
            if (c->opt.*(e.field) == value) return FHIP_OK;
            const fhip_status st = fhip_ctx_sync(c);
            c->opt.*(e.field) = value;
            apply_options(c);
            return st;
        }
    return fail(c, FHIP_ERR_UNSUPPORTED, std::string("unknown option ") + name);
}
fhip_status fhip_ctx_get_option(const fhip_ctx* c, const char* name, int* value) {
    if (!c || !name || !value) return FHIP_ERR_UNSUPPORTED;
    for (const FhOptionEntry& e : FH_OPTION_TABLE)
        if (!strcmp(e.name, name)) { *value = c->opt.*(e.field); return FHIP_OK; }
    return FHIP_ERR_UNSUPPORTED;
}

// ---- tapes -----------------------------------------------------------------------------
static fhip_status finish_tape(fhip_ctx* ctx, fh::SsaProgram& prog, fhip_tape** out) {
    std::string err;
    fhip_tape* t = new fhip_tape();
    if (!fh::allocate(prog, t->t, err)) { delete t; return fail(ctx, FHIP_ERR_UNSUPPORTED, err); }
    if (t->t.n_vars > FH_MAX_INPUTS) { delete t; return fail(ctx, FHIP_ERR_UNSUPPORTED, "more than 16 input variables"); }
    // (FHIP_GROUPS_MIN_OPS / FHIP_GROUPS_MIN_TERMS: tests lower the thresholds to send small shapes down this path)
    const size_t min_ops = getenv("FHIP_GROUPS_MIN_OPS") ? (size_t)atol(getenv("FHIP_GROUPS_MIN_OPS")) : 1024;
    const uint32_t min_terms = getenv("FHIP_GROUPS_MIN_TERMS") ? (uint32_t)atol(getenv("FHIP_GROUPS_MIN_TERMS")) : 32;
    const uint32_t want_groups = (uint32_t)std::min<long>(FH_MAX_GROUPS, std::max<long>(2, getenv("FHIP_GROUPS") ? atol(getenv("FHIP_GROUPS")) : 32));
    if (t->t.ops.size() >= min_ops && !getenv("FHIP_NO_GROUPS")) {
        std::vector<fh::SsaProgram> gp;
        const int op = fh::split_root(prog, want_groups, min_terms, gp);
        if (op >= 0) {
            t->groups.resize(gp.size());
            bool ok = true;
            for (size_t g = 0; g < gp.size() && ok; g++) ok = fh::allocate(gp[g], t->groups[g], err);
            if (ok) t->group_op = op; else t->groups.clear();
        }
        if (fh::plan_terms(prog, want_groups, min_terms, 16, t->plan)) {
            t->tgroups.resize(t->plan.groups.size());
            bool ok = true;
            for (size_t g = 0; g < t->tgroups.size() && ok; g++) ok = fh::allocate(t->plan.groups[g], t->tgroups[g], err);
            if (!ok) t->tgroups.clear();
            t->plan.groups.clear();
        }
    }
    *out = t;
    return FHIP_OK;
}
uint32_t fhip_tape_group_count(const fhip_tape* tape) { return (uint32_t)tape->groups.size(); }
int fhip_tape_group_op(const fhip_tape* tape) { return tape->group_op; }
fhip_status fhip_tape_group(fhip_ctx* ctx, const fhip_tape* tape, uint32_t g, fhip_tape** out) {
    if (g >= tape->groups.size()) return fail(ctx, FHIP_ERR_BAD_TAPE, "no such tape group");
    fhip_tape* t = new fhip_tape();
    t->t = tape->groups[g];
    *out = t;
    return FHIP_OK;
}
fhip_status fhip_tape_term_group(fhip_ctx* ctx, const fhip_tape* tape, uint32_t g, fhip_tape** out) {
    if (g >= tape->tgroups.size()) return fail(ctx, FHIP_ERR_BAD_TAPE, "no such term group");
    fhip_tape* t = new fhip_tape();
    t->t = tape->tgroups[g];
    *out = t;
    return FHIP_OK;
}
uint32_t fhip_tape_term_tree(const fhip_tape* tape, uint32_t* words, uint32_t cap_ops) {
    const uint32_t n = (uint32_t)std::min<size_t>(tape->plan.top.size(), cap_ops);
    for (uint32_t i = 0; i < n; i++) {
        const fh::TopOp& o = tape->plan.top[i];
        words[3 * i] = (uint32_t)o.op | ((uint32_t)o.out << 8) | ((uint32_t)o.a_kind << 16) | ((uint32_t)o.b_kind << 24);
        words[3 * i + 1] = o.a; words[3 * i + 2] = o.b;
    }
    return (uint32_t)tape->plan.top.size();
}
uint32_t fhip_tape_term_choice_src(const fhip_tape* tape, uint32_t* src, uint32_t cap) {
    const uint32_t n = (uint32_t)std::min<size_t>(tape->plan.choice_src.size(), cap);
    for (uint32_t i = 0; i < n; i++) src[i] = tape->plan.choice_src[i];
    return (uint32_t)tape->plan.choice_src.size();
}
uint32_t fhip_tape_term_plan(const fhip_tape* tape, uint32_t info[4]) {
    info[0] = tape->plan.n_terms; info[1] = (uint32_t)tape->plan.top.size(); info[2] = tape->plan.top_regs;
    info[3] = (uint32_t)tape->plan.choice_src.size();
    return (uint32_t)tape->tgroups.size();
}
// Launch one of the assembly kernels: `waves` single-wave workgroups, raw kernarg block
static hipError_t launch_asm(fhip_ctx* ctx, int which, uint32_t waves, void* args, size_t bytes, size_t lds = 0, uint32_t grid_y = 1,
                             hipStream_t stream = nullptr) {
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &bytes, HIP_LAUNCH_PARAM_END};
    hipStream_t const st = stream ? stream : ctx->stream;
    hipEvent_t ea = nullptr, eb = nullptr;
    if (ctx->profiling) { (void)hipEventCreate(&ea); (void)hipEventCreate(&eb); (void)hipEventRecord(ea, st); }
    const hipError_t e = hipModuleLaunchKernel(ctx->asm_fn[which], waves, grid_y, 1, WAVE, 1, 1, (unsigned)lds, st, nullptr, extra);
    if (ctx->profiling) { (void)hipEventRecord(eb, st); ctx->asm_events.push_back({which, {ea, eb}}); }
    if (e != hipSuccess) {
        ctx->launch_failed = true;
        if (ctx->err.empty()) ctx->err = std::string("launch of ") + FH_ASM_NAMES[which] + ": " + hipGetErrorString(e);
    }
    return e;
}
// ... the *_t tile kernels add the unary transcendental ones
static bool tape_tiles_t_ok(const fh::HostTape& t) {
    for (uint64_t w : t.ops) {
        const uint32_t op = FH_W_OP((uint32_t)w);
        if (op == FH_RAND) return false;
        if (op >= FH_ADD_RR) {
            const int base = op >= FH_SUB_IR ? (int[]){1, 3, 4, 5, 6, 7}[op - FH_SUB_IR] : (int)((op - FH_ADD_RR) % 12);
            if (base == 4 || base == 6 || base == 7) return false;  // atan2, mix, mod
        }
    }
    return true;
}
static bool tape_has_mod(const fh::HostTape& t) {
    for (uint64_t w : t.ops) {
        const uint32_t op = FH_W_OP((uint32_t)w);
        if (op == FH_MOD_RR || op == FH_MOD_RI || op == FH_MOD_IR) return true;
    }
    return false;
}
// The assembly interpreters implement every opcode except the transcendental, modulo and rng ones
static bool tape_asm_ok(const fh::HostTape& t) {
    for (uint64_t w : t.ops) {
        const uint32_t op = FH_W_OP((uint32_t)w);
        if ((op >= FH_SIN && op <= FH_LN) || op == FH_RAND) return false;
        if (op >= FH_ADD_RR) {
            const int base = op >= FH_SUB_IR ? (int[]){1, 3, 4, 5, 6, 7}[op - FH_SUB_IR] : (int)((op - FH_ADD_RR) % 12);
            if (base == 4 || base == 6 || base == 7) return false;  // atan2, mix, mod
        }
    }
    return true;
}

static fhip_status tape_to_device(fhip_ctx* ctx, const fhip_tape* t) {
    std::lock_guard<std::mutex> guard(t->upload_lock);
    (void)hipSetDevice(ctx->device);
    // (a tape's lazily made device copies live on the device of the first context that needed them: a tape used from several
    // devices has to be built per device - refused rather than dereferenced from the wrong one)
    if (t->device >= 0 && t->device != ctx->device) return fail(ctx, FHIP_ERR_UNSUPPORTED, "this tape's device copies belong to another device: build the tape per device");
    if (t->d_ops) return FHIP_OK;
    t->device = ctx->device;
    size_t bytes = (t->t.ops.size() + 16) * 8;  // slack: the interpreters prefetch up to 12 ops past the end
    uint64_t* d = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d, bytes));
    HIP_TRY(ctx, hipMemset(d, 0, bytes));
    HIP_TRY(ctx, hipMemcpy(d, t->t.ops.data(), t->t.ops.size() * 8, hipMemcpyHostToDevice));
    t->d_ops = d;   // published only when complete
    return FHIP_OK;
}
fhip_status fhip_tape_from_bytecode(fhip_ctx* ctx, const uint32_t* words, size_t n_words, fhip_tape** out) {
    fh::SsaProgram prog;
    std::string err;
    if (!fh::from_bytecode(words, n_words, prog, err)) return fail(ctx, FHIP_ERR_BAD_TAPE, err);
    return finish_tape(ctx, prog, out);
}
fhip_status fhip_tape_from_graph(fhip_ctx* ctx, const fhip_graph* g, const uint32_t* roots, uint32_t n_roots,
                                 fhip_tape** out) {
    fh::SsaProgram prog;
    std::string err;
    std::vector<fh::NodeId> r(roots, roots + n_roots);
    if (!fh::flatten(g->g, r, prog, err)) return fail(ctx, FHIP_ERR_BAD_TAPE, err);
    return finish_tape(ctx, prog, out);
}
void fhip_tape_free(fhip_tape* t) {
    if (!t) return;
    if (t->d_ops) (void)hipFree(t->d_ops);
    if (t->d_top) (void)hipFree(t->d_top);
    if (t->d_chsrc) (void)hipFree(t->d_chsrc);
    if (t->d_links) (void)hipFree(t->d_links);
    if (t->d_ctab) (void)hipFree(t->d_ctab);
    delete t;
}
uint32_t fhip_tape_len(const fhip_tape* t) { return (uint32_t)t->t.ops.size(); }
uint32_t fhip_tape_choice_count(const fhip_tape* t) { return t->t.n_choices; }
uint32_t fhip_tape_reg_count(const fhip_tape* t) { return t->t.n_regs; }
uint32_t fhip_tape_var_count(const fhip_tape* t) { return t->t.n_vars; }
uint32_t fhip_tape_output_count(const fhip_tape* t) { return t->t.n_outputs; }
fhip_status fhip_tape_reg_tape(const fhip_tape* t, uint32_t n_regs, uint32_t* reg_ops, uint32_t cap_ops, uint32_t* words, uint32_t cap_words,
                               uint32_t info[4]) {
    fh::RegTapeOut rt;
    std::string err;
    for (int i = 0; i < 4; i++) info[i] = 0;
    if (!fh::reg_tape(t->t, n_regs, rt, err)) return FHIP_ERR_BAD_TAPE;
    info[0] = (uint32_t)rt.ops.size(); info[1] = rt.slot_count;
    if (reg_ops)
        for (size_t i = 0; i < rt.ops.size() && i < cap_ops; i++) {
            const fh::RegOp& o = rt.ops[rt.ops.size() - 1 - i];       // evaluation order
            reg_ops[4 * i] = o.op; reg_ops[4 * i + 1] = o.out; reg_ops[4 * i + 2] = o.a;
            reg_ops[4 * i + 3] = fh_is_rr(o.op) ? (uint32_t)o.b : o.w;
        }
    std::vector<uint32_t> w;
    const bool ok = fh::reg_tape_bytecode(rt, n_regs, w, info[2], info[3]);
    if (!ok) return FHIP_ERR_UNSUPPORTED;
    if (words) for (size_t i = 0; i < w.size() && i < cap_words; i++) words[i] = w[i];
    return FHIP_OK;
}
uint32_t fhip_tape_ops(const fhip_tape* t, uint64_t* ops, uint32_t cap) {
    for (uint32_t i = 0; i < t->t.ops.size() && i < cap; i++) ops[i] = t->t.ops[i];
    return (uint32_t)t->t.ops.size();
}
int fhip_tape_axis_slot(const fhip_tape* t, int axis) { return (axis >= 0 && axis < 3) ? t->t.vars.axis[axis] : -1; }
int fhip_tape_var_slot(const fhip_tape* t, uint64_t index) { return t->t.vars.slot_of(3, index); }

// Host form of the device prune sweep (kernels.hip: prune_sweep<true>), same algorithm.
fhip_status fhip_simplify(fhip_ctx* ctx, const fhip_tape* tape, const uint8_t* choices, uint32_t n_choices,
                          fhip_tape** child) {
    const fh::HostTape& p = tape->t;
    if (n_choices != p.n_choices) return fail(ctx, FHIP_ERR_BAD_CHOICE_SLICE, "choice slice length mismatch");
    std::vector<int> map(FH_MAX_REGS, -1);
    fh::RegPool pool;
    std::vector<uint64_t> rev;
    uint32_t ci = n_choices, kept = 0;
    auto use = [&](uint32_t r) { if (map[r] < 0) map[r] = pool.take(); return (uint32_t)map[r]; };
    for (size_t k = p.ops.size(); k-- > 0;) {
        const uint64_t w = p.ops[k];
        const uint32_t w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
        const uint32_t op = FH_W_OP(w0), ro = FH_W_OUT(w0), ra = FH_W_A(w0), rb = w1;
        const bool is_choice = fh_is_choice(op);
        uint32_t c = FH_CHOICE_BOTH;
        if (is_choice) {
            c = choices[--ci];
            if (c == FH_CHOICE_UNKNOWN) return fail(ctx, FHIP_ERR_BAD_CHOICE_SLICE, "Choice::Unknown in trace");
        }
        if (op == FH_OUTPUT) { rev.push_back(fh_pack(op, 0, use(ra), 0, w1)); continue; }
        const int no = map[ro];
        if (no < 0) continue;
        map[ro] = -1;
        int alias = -1;
        bool copy_imm = false;
        if (op == FH_COPY_REG) alias = (int)ra;
        else if (is_choice && c == FH_CHOICE_LEFT) alias = (int)ra;
        else if (is_choice && c == FH_CHOICE_RIGHT) { if (fh_is_rr(op)) alias = (int)rb; else copy_imm = true; }
        if (alias >= 0) {
            if (map[alias] < 0) { map[alias] = no; continue; }
            pool.give(no);
            rev.push_back(fh_pack(FH_COPY_REG, no, map[alias], 0, 0));
            continue;
        }
        pool.give(no);
        if (copy_imm) { rev.push_back(fh_pack(FH_COPY_IMM, no, 0, 0, w1)); continue; }
        uint32_t na = 0, nb = 0;
        if (op != FH_INPUT && op != FH_COPY_IMM) na = use(ra);
        if (fh_is_rr(op)) nb = use(rb);
        if (is_choice) kept++;
        rev.push_back(fh_pack(op, no, na, nb, w1));
    }
    fhip_tape* t = new fhip_tape();
    t->t.ops.assign(rev.rbegin(), rev.rend());
    t->t.n_regs = pool.high;
    t->t.n_choices = kept;
    t->t.n_outputs = p.n_outputs;
    t->t.n_vars = p.n_vars;  // children keep the parent's variable slots (vm/data.rs:316)
    t->t.vars = p.vars;
    *child = t;
    return FHIP_OK;
}

// ---- evaluators ------------------------------------------------------------------------
static fhip_status tracing_eval(fhip_ctx* ctx, const fhip_tape* tape, const float* vars, uint32_t n_vars, uint32_t n,
                                float* out, uint8_t* choices, uint8_t* simplify, bool interval) {
    const fh::HostTape& t = tape->t;
    if (n_vars < t.n_vars) return fail(ctx, FHIP_ERR_BAD_VAR_SLICE, "too few variables");
    if (n == 0) return FHIP_OK;
    { fhip_status ts_ = tape_to_device(ctx, tape); if (ts_) return ts_; }
    const uint32_t comp = interval ? 2 : 1;
    const size_t nv = std::max<uint32_t>(n_vars, 1);
    std::vector<float> hv((size_t)n * nv * comp, 0.0f);
    if (interval) {
        if (n_vars) memcpy(hv.data(), vars, (size_t)n * n_vars * 8);
    } else {
        for (uint32_t i = 0; i < n; i++)
            for (uint32_t v = 0; v < n_vars; v++) hv[(size_t)v * n + i] = vars[(size_t)i * n_vars + v];  // -> [var][n]
    }
    const size_t out_elems = (size_t)n * t.n_outputs * comp;
    const size_t ch_bytes = (size_t)n * std::max<uint32_t>(t.n_choices, 1);
    HIP_TRY(ctx, ctx->io_a.ensure(hv.size() * 4));
    HIP_TRY(ctx, ctx->io_b.ensure(std::max<size_t>(out_elems, 1) * 4));
    HIP_TRY(ctx, ctx->io_c.ensure(ch_bytes));
    HIP_TRY(ctx, ctx->io_d.ensure(n));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->io_a.p, hv.data(), hv.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->io_b.p, 0xFF, std::max<size_t>(out_elems, 1) * 4, ctx->stream));  // NaN prefill (vm/mod.rs:314-319)
    HIP_TRY(ctx, hipMemsetAsync(ctx->io_c.p, 0, ch_bytes, ctx->stream));
    const uint32_t grid = (n + WAVE - 1) / WAVE;
    const uint32_t nr = std::max<uint32_t>(t.n_regs, 1);
    const size_t lds = (size_t)nr * WAVE * 4 * comp;
    const bool g = lds > FH_LDS_MAX;  // register file too large for LDS: global scratch slab
    if (g) HIP_TRY(ctx, ctx->io_e.ensure(lds * grid));
    if (interval) {
        if (g) hipLaunchKernelGGL(k_eval_interval<true>, dim3(grid), dim3(WAVE), 0, ctx->stream, tape->d_ops, (uint32_t)t.ops.size(),
                           (const float2*)ctx->io_a.p, (uint32_t)nv, n, (float2*)ctx->io_b.p, t.n_outputs,
                           (uint8_t*)ctx->io_c.p, (uint8_t*)ctx->io_d.p, t.n_choices, (IV*)ctx->io_e.p, nr);
        else hipLaunchKernelGGL(k_eval_interval<false>, dim3(grid), dim3(WAVE), lds, ctx->stream, tape->d_ops, (uint32_t)t.ops.size(),
                           (const float2*)ctx->io_a.p, (uint32_t)nv, n, (float2*)ctx->io_b.p, t.n_outputs,
                           (uint8_t*)ctx->io_c.p, (uint8_t*)ctx->io_d.p, t.n_choices, (IV*)nullptr, nr);
    } else {
        if (g) hipLaunchKernelGGL(k_eval_f32<true>, dim3(grid), dim3(WAVE), 0, ctx->stream, tape->d_ops, (uint32_t)t.ops.size(),
                           (const float*)ctx->io_a.p, n, (float*)ctx->io_b.p, (uint8_t*)ctx->io_c.p,
                           (uint8_t*)ctx->io_d.p, t.n_choices, (float*)ctx->io_e.p, nr);
        else hipLaunchKernelGGL(k_eval_f32<false>, dim3(grid), dim3(WAVE), lds, ctx->stream, tape->d_ops, (uint32_t)t.ops.size(),
                           (const float*)ctx->io_a.p, n, (float*)ctx->io_b.p, (uint8_t*)ctx->io_c.p,
                           (uint8_t*)ctx->io_d.p, t.n_choices, (float*)nullptr, nr);
    }
    HIP_TRY(ctx, hipGetLastError());
    std::vector<float> ho(std::max<size_t>(out_elems, 1));
    HIP_TRY(ctx, hipMemcpyAsync(ho.data(), ctx->io_b.p, out_elems * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (choices && t.n_choices)
        HIP_TRY(ctx, hipMemcpyAsync(choices, ctx->io_c.p, (size_t)n * t.n_choices, hipMemcpyDeviceToHost, ctx->stream));
    if (simplify) HIP_TRY(ctx, hipMemcpyAsync(simplify, ctx->io_d.p, n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (interval) memcpy(out, ho.data(), out_elems * 4);
    else
        for (uint32_t i = 0; i < n; i++)
            for (uint32_t o = 0; o < t.n_outputs; o++) out[(size_t)i * t.n_outputs + o] = ho[(size_t)o * n + i];
    return FHIP_OK;
}
fhip_status fhip_interval_eval(fhip_ctx* ctx, const fhip_tape* tape, const float* vars, uint32_t n_vars, uint32_t n,
                               float* out, uint8_t* choices, uint8_t* simplify) {
    return tracing_eval(ctx, tape, vars, n_vars, n, out, choices, simplify, true);
}
fhip_status fhip_point_eval(fhip_ctx* ctx, const fhip_tape* tape, const float* vars, uint32_t n_vars, uint32_t n,
                            float* out, uint8_t* choices, uint8_t* simplify) {
    return tracing_eval(ctx, tape, vars, n_vars, n, out, choices, simplify, false);
}

static fhip_status bulk_eval(fhip_ctx* ctx, const fhip_tape* tape, const float* const* vars, const uint32_t* lens,
                             uint32_t n_vars, float* const* out, uint32_t comp) {
    const fh::HostTape& t = tape->t;
    if (n_vars < t.n_vars) return fail(ctx, FHIP_ERR_BAD_VAR_SLICE, "too few variable slices");
    const uint32_t n = n_vars ? lens[0] : 0;  // vm/mod.rs:808
    for (uint32_t v = 1; v < n_vars; v++)
        if (lens[v] != n) return fail(ctx, FHIP_ERR_MISMATCHED_SLICES, "variable slices differ in length");
    if (n == 0) return FHIP_OK;
    { fhip_status ts_ = tape_to_device(ctx, tape); if (ts_) return ts_; }
    const size_t row = (size_t)n * comp;
    HIP_TRY(ctx, ctx->io_a.ensure(row * 4 * n_vars));
    HIP_TRY(ctx, ctx->io_b.ensure(row * 4 * std::max<uint32_t>(t.n_outputs, 1)));
    for (uint32_t v = 0; v < n_vars; v++)
        HIP_TRY(ctx, hipMemcpyAsync((float*)ctx->io_a.p + v * row, vars[v], row * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->io_b.p, 0xFF, row * 4 * std::max<uint32_t>(t.n_outputs, 1), ctx->stream));
    const uint32_t grid = (n + WAVE - 1) / WAVE;
    const uint32_t nr = std::max<uint32_t>(t.n_regs, 1);
    const size_t lds = (size_t)nr * WAVE * 4 * comp;
    const bool g = lds > FH_LDS_MAX;
    if (g) HIP_TRY(ctx, ctx->io_e.ensure(lds * grid));
    if (comp == 1 && ctx->use_asm && nr <= 32) {
        // 64 * ZB samples per wave, register file in VGPRs (gen_interp.py); tapes with transcendental / modulo / rng opcodes: the kernels
        // whose handlers call the compiled routines
        const bool plain = tape_asm_ok(t);
        struct { const uint64_t* tape; const float* vars; float* out; uint32_t len, n; } ka = {
            tape->d_ops, (const float*)ctx->io_a.p, (float*)ctx->io_b.p, (uint32_t)t.ops.size(), n};
        const uint32_t per = nr <= 16 ? 256 : 128;
        HIP_TRY(ctx, launch_asm(ctx, nr <= 16 ? (plain ? FH_ASM_FLOAT_16x4 : FH_ASM_FLOAT_16x4_T) : (plain ? FH_ASM_FLOAT_32x2 : FH_ASM_FLOAT_32x2_T), (n + per - 1) / per, &ka,
                                sizeof(ka)));
    } else if (comp == 1) {
        if (g) hipLaunchKernelGGL(k_eval_f32<true>, dim3(grid), dim3(WAVE), 0, ctx->stream, tape->d_ops, (uint32_t)t.ops.size(),
                           (const float*)ctx->io_a.p, n, (float*)ctx->io_b.p, (uint8_t*)nullptr, (uint8_t*)nullptr, 0u, (float*)ctx->io_e.p, nr);
        else hipLaunchKernelGGL(k_eval_f32<false>, dim3(grid), dim3(WAVE), lds, ctx->stream, tape->d_ops, (uint32_t)t.ops.size(),
                           (const float*)ctx->io_a.p, n, (float*)ctx->io_b.p, (uint8_t*)nullptr, (uint8_t*)nullptr, 0u, (float*)nullptr, nr);
    } else {
        if (g) hipLaunchKernelGGL(k_eval_grad<true>, dim3(grid), dim3(WAVE), 0, ctx->stream, tape->d_ops, (uint32_t)t.ops.size(),
                           (const float4*)ctx->io_a.p, n, (float4*)ctx->io_b.p, (GR*)ctx->io_e.p, nr);
        else hipLaunchKernelGGL(k_eval_grad<false>, dim3(grid), dim3(WAVE), lds, ctx->stream, tape->d_ops, (uint32_t)t.ops.size(),
                           (const float4*)ctx->io_a.p, n, (float4*)ctx->io_b.p, (GR*)nullptr, nr);
    }
    HIP_TRY(ctx, hipGetLastError());
    for (uint32_t o = 0; o < t.n_outputs; o++)
        HIP_TRY(ctx, hipMemcpyAsync(out[o], (float*)ctx->io_b.p + o * row, row * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return FHIP_OK;
}
fhip_status fhip_float_eval(fhip_ctx* ctx, const fhip_tape* tape, const float* const* vars, const uint32_t* lens,
                            uint32_t n_vars, float* const* out) {
    return bulk_eval(ctx, tape, vars, lens, n_vars, out, 1);
}
fhip_status fhip_grad_eval(fhip_ctx* ctx, const fhip_tape* tape, const float* const* vars, const uint32_t* lens,
                           uint32_t n_vars, float* const* out) {
    return bulk_eval(ctx, tape, vars, lens, n_vars, out, 4);
}

// ---- geometry --------------------------------------------------------------------------
// RegionSize::screen_to_world (render/region.rs:87-108): identity, then nalgebra's
// append_translation_mut(-center) and append_nonuniform_scaling_mut(scale, -scale, ..)
void fhip_screen_to_world(const uint32_t* size, int n, float* out) {
    const int d = n + 1;
    float center[3] = {0, 0, 0};
    uint32_t smallest = size[0];
    for (int i = 0; i < n; i++) { center[i] = (float)size[i] / 2.0f; smallest = std::min(smallest, size[i]); }
    center[1] -= 1.0f;
    const float scale = 2.0f / (float)smallest;
    for (int i = 0; i < d * d; i++) out[i] = (i / d == i % d) ? 1.0f : 0.0f;
    for (int col = 0; col < d; col++)
        for (int row = 0; row < n; row++) out[row * d + col] += (-center[row]) * out[n * d + col];
    for (int row = 0; row < n; row++) {
        float s = scale;
        if (row == 1) s *= -1.0f;
        for (int col = 0; col < d; col++) out[row * d + col] *= s;
    }
}
// nalgebra's small-matrix product: per output column, accumulate a[:,k] * b[k][col] for k = 0..d-1
static void mat_product(const float* a, const float* b, int d, float* out) {
    for (int col = 0; col < d; col++)
        for (int row = 0; row < d; row++) {
            float acc = a[row * d] * b[col];
            for (int k = 1; k < d; k++) acc = a[row * d + k] * b[k * d + col] + acc;
            out[row * d + col] = acc;
        }
}

// ---- renders ---------------------------------------------------------------------------
static const uint32_t VM_TILES_2D[] = {128, 32, 8};        // fidget-core/src/vm/mod.rs:255-257
static const uint32_t VM_TILES_3D[] = {128, 64, 32, 16, 8};  // fidget-core/src/vm/mod.rs:251-253
// RenderHints of the HIP shape (the reference lets every shape type pick its own, shape.rs RenderHints):
// a fan-out of 4^3 = 64 children fills a wavefront
// (128 -> 32 -> 8).  The root tile stays the one the reference's VmShape hints give for the image
// size, so that exactly the same voxels are covered (a root tile overhanging the image in z is
// evaluated there by the reference too).

struct RenderSetup {
    FhRenderState S;
    std::vector<FhGroup> roots;
    uint32_t n_slabs = 1, n_layers = 1;      // z-slabs (steps of the per-slab chains), root-tile layers
    uint32_t slab_lo = 0, slab_hi = 1;   // z-slabs this render covers (all of them unless the volume is split in z: octant shards)
    size_t lds_tiles_mid = 0, lds_tiles_big = 0, lds_tiles_small = 0, lds_points_big = 0, lds_normals_big = 0, lds_normals_small = 0;
    uint32_t table_words = 0, n_footprints = 0, groups_per_slab = 0;
    size_t mind_words = 0;      // words of the min-depth pyramid (cleared at the head of the frame)
    uint32_t tl = 16;  // sibling tiles per wave in the tile kernel (16 or 64)
    bool full = false;  // tape uses transcendental / modulo ops -> FULL kernel variants
    bool asm_points = false;  // leaf stage on the assembly interpreters
    bool asm_points_t = false;  // ... on fh_columns_t (tapes with transcendental / modulo / rng opcodes)
    bool asm_normals = false;   // normals by the assembly gradient interpreter fh_normals (gen_normals.py): footprints of leaves of <= 32 registers
    bool split = false;       // 3D tile stage as setup / evaluate+prune / push kernels
    bool asm_tiles = false;   // ... with the evaluate+prune step in assembly (fh_tiles)
    bool asm_tiles_t = false; // ... by the *_t variants (transcendental opcodes)
    uint32_t group_regs = 0, group_choices = 0;  // bounds over the tape's groups
    size_t lds_tiles_group = 0;
    bool groups = false;      // ... and level 0 evaluated as the tape's independent groups (tape parallelism)
    bool prune1 = false;      // ... and, on the first exp_levels levels, the prune as one wave per child (fh_prune1)
    bool prune2 = false;      // ... by the linked prune (prune2.hip k_prune2: visits only the ops a child keeps) where the tape qualifies
    const uint64_t* d_links = nullptr;
    const uint64_t* d_ctab = nullptr;
    size_t lds_prune2 = 0;
    bool prune2_l1 = false;   // ... and level 1 by the same kernel, on the links the level-0 launch leaves in front of every child tape (option
                              // prune2_l1, off: fh_tiles_v64's forward pass alone takes 0.13 ms of its 0.39, but one wave per 32^3 child - 6 120 of them,
                              // 2 to a SIMD, each bound by scalar issue - takes 0.75 ms where the lockstep sweep takes 0.26; profiles/r03o)
    size_t lds_prune2_l1 = 0;
    uint32_t exp_levels = 0;
    uint32_t col_slots = 0, col_depmask = 0, col_flags = 0;   // 3D: axis slots x | y << 8 | z << 16 (0xFF none), inputs varying along a pixel column, bit 16 projective
    bool zrep = false;        // ... column-invariant parents are evaluated for one z-layer only (k_tape_flags)
    bool big_hbm = false;     // the root-sized register files live in HBM (S.gscratch): hbm_waves workgroups per root-sized launch
    uint32_t hbm_waves = 0;
};

static fhip_status bind_inputs(fhip_ctx* ctx, const fhip_tape* tape, const int32_t* axis_slots, const uint64_t* keys,
                               const float* vals, uint32_t n, FhRender& P) {
    const fh::HostTape& t = tape->t;
    std::vector<char> bound(FH_MAX_INPUTS, 0);
    for (uint32_t s = 0; s < FH_MAX_INPUTS; s++) { P.in_kind[s] = 3; P.in_value[s] = 0.0f; }
    for (int a = 0; a < 3; a++) {
        const int s = axis_slots ? axis_slots[a] : t.vars.axis[a];
        if (s >= 0 && s < FH_MAX_INPUTS) { P.in_kind[s] = (uint32_t)a; bound[s] = 1; }
    }
    for (uint32_t i = 0; i < n; i++) {
        // graph-built tapes: keys are Var::V indices; bytecode tapes (axis_slots given): keys are slots
        const int s = axis_slots ? (int)keys[i] : t.vars.slot_of(3, keys[i]);
        if (s >= 0 && s < FH_MAX_INPUTS) { P.in_value[s] = vals[i]; bound[s] = 1; }
    }
    for (uint32_t s = 0; s < t.n_vars; s++)
        if (!bound[s]) return fail(ctx, FHIP_ERR_MISSING_VAR, "a variable of the shape has no value");
    return FHIP_OK;
}

// fidget-raster/src/lib.rs:59-66
static std::vector<uint32_t> trim_tiles(const uint32_t* tiles, uint32_t n, uint32_t max_size) {
    uint32_t i = n;
    for (uint32_t k = 0; k < n; k++) if (tiles[k] < max_size) { i = k; break; }
    i = i ? i - 1 : 0;
    return std::vector<uint32_t>(tiles + i, tiles + n);
}

static std::vector<uint32_t> hip_tiles_3d(uint32_t max_size, bool vm_tiles) {
    std::vector<uint32_t> v = trim_tiles(VM_TILES_3D, 5, max_size);
    if (vm_tiles) return v;  // diagnostics: the reference's own subdivision
    std::vector<uint32_t> out{v[0]};
    for (uint32_t t = v[0]; t > 8;) { t = std::max<uint32_t>(t / 4, 8); out.push_back(t); }
    return out;
}

// 2D hint of the HIP shape: 128 -> 16 with 16 x 16 pixel leaves - what fidget-jit uses (fidget-jit/src/lib.rs:984-986); a fan-out
// of 64 children per parent fills a wavefront of the tile-stage kernels (the VM's 128 / 32 / 8 fans out by 16)
static const uint32_t HIP_TILES_2D[] = {128, 16};
static bool tape_is_full(const fh::HostTape& t) {
    for (uint64_t w : t.ops) {
        const uint32_t op = FH_W_OP((uint32_t)w);
        if ((op >= FH_SIN && op <= FH_LN) || op == FH_ATAN2_RR || op == FH_ATAN2_RI || op == FH_ATAN2_IR ||
            op == FH_MOD_RR || op == FH_MOD_RI || op == FH_MOD_IR)
            return true;
    }
    return false;
}

// medium LDS layout of the tile stage (pre-pass levels below the root): 48 KB, three waves per CU
static const uint32_t MID_REGS = 64, MID_CHOICES = 768;
static size_t tiles_lds(uint32_t regs, uint32_t choices, uint32_t TL) {
    size_t b = (size_t)regs * TL * 8 + (size_t)((choices + 15) / 16) * TL * 4 + (size_t)regs * TL + 256;
    return (b + 15) & ~(size_t)15;
}

// Which part of the volume a render covers (multi-GPU): root-tile columns round robin (index % n_shards == shard, full
// depth), or a block of an nx x ny x nz split of the root-tile grid and of the z-slabs (octants: 2 x 2 x 2)
struct PartSpec {
    uint32_t shard = 0, n_shards = 1;
    uint32_t ix = 0, nx = 1, iy = 0, ny = 1, iz = 0, nz = 1;
};
static fhip_status prepare(fhip_ctx* ctx, const fhip_tape* tape, bool is3d, const std::vector<uint32_t>& ts,
                           const PartSpec& part, RenderSetup& R) {
    const uint32_t shard = part.shard, n_shards = part.n_shards;
    FhRenderState& S = R.S;
    FhRender& P = S.P;
    const fh::HostTape& t = tape->t;
    if (t.n_outputs != 1) return fail(ctx, FHIP_ERR_BAD_TAPE, "shape tapes have exactly one output");
    if (ts.empty() || ts.size() > FH_MAX_LEVELS) return fail(ctx, FHIP_ERR_UNSUPPORTED, "1..8 tile levels supported");
    P.n_levels = (uint32_t)ts.size();
    uint32_t fanout = 1;
    for (size_t i = 0; i < ts.size(); i++) {
        P.tiles[i] = ts[i];
        if (i) {
            if (ts[i - 1] <= ts[i] || ts[i - 1] % ts[i]) return fail(ctx, FHIP_ERR_UNSUPPORTED, "bad tile size list");
            const uint32_t n = ts[i - 1] / ts[i];
            fanout = std::max(fanout, is3d ? n * n * n : n * n);
        }
    }
    if (fanout > 64) return fail(ctx, FHIP_ERR_UNSUPPORTED, "tile fan-out above 64 children");
    const uint32_t TL = R.tl = fanout > 16 ? 64 : 16;
    if (is3d && ts.back() != 8) return fail(ctx, FHIP_ERR_UNSUPPORTED, "3D leaves must be 8^3 (one 8x8 footprint per wave)");
    // (register numbers are 12-bit fields of a tape word.  The device prunes keep old -> new register maps in bytes with 0xFF =
    // dead: a CHILD tape has 255 registers at most - one that would need more keeps its parent's tape; the root tape may have
    // more, its register file then lives in HBM: gscratch below)
    if (t.n_regs >= FH_MAX_REGS) return fail(ctx, FHIP_ERR_UNSUPPORTED, "renders support up to 4095 registers");
    if (t.ops.size() >= (1u << 24)) return fail(ctx, FHIP_ERR_UNSUPPORTED, "renders support tapes of up to 2^24 ops");   // (FhLeafRef packs length | registers << 24)
    P.max_regs = std::max<uint32_t>(t.n_regs, 1);
    P.max_choices = t.n_choices;
    P.roots_x = (P.width + ts[0] - 1) / ts[0];
    P.roots_y = (P.height + ts[0] - 1) / ts[0];
    // z-slabs: the per-slab chains (tile stage, leaf kernel, tail) take `slab_layers` root-tile layers per step when the coarse
    // levels are evaluated for the whole volume up front (the length of the tile chain is its number of steps: every step's
    // launches leave most of the machine idle); one layer per step otherwise
    const uint32_t n_layers = is3d ? (P.depth + ts[0] - 1) / ts[0] : 1;
    const bool prepass_ok = is3d && ts.size() >= 3 && n_layers <= FH_MAX_SLABS;
    uint32_t SL = prepass_ok ? (uint32_t)std::max(1, std::min(8, ctx->opt.slab_layers)) : 1u;
    // (the leaf table: <= 64 eight-voxel layers per slab; at least two slabs, so that the tile stage of one still runs beside the
    // leaf kernel of the other - bear.vm at 512^3, four layers: 3.68 ms per frame as two slabs, 3.77 as one)
    while (SL > 1 && (ts[0] * SL / 8 > 64 || SL * 2 > n_layers)) SL >>= 1;
    P.slab = ts[0] * SL;
    R.n_slabs = is3d ? (P.depth + P.slab - 1) / P.slab : 1;
    R.n_layers = n_layers;
    R.full = tape_is_full(t);
    // assembly leaf kernels: supported opcodes only (any 4x4 screen-to-model matrix, projective ones included)
    R.asm_points = ctx->use_asm && is3d && (tape_asm_ok(t) || !ctx->opt.no_columns_t);
    R.asm_points_t = R.asm_points && !tape_asm_ok(t);   // transcendental / modulo / rng opcodes: the variant that calls the compiled routines
    // (fh_normals_t has the transcendental, rng and atan2 handlers; a modulo's gradient - div_euclid - keeps the C++ kernel)
    R.asm_normals = R.asm_points && !ctx->opt.no_asm_normals && (!R.asm_points_t || !tape_has_mod(t));

    // LDS budgets: BIG = bounded by the root tape (children never need more); SMALL = fixed
    R.lds_tiles_big = tiles_lds(P.max_regs, P.max_choices, TL);
    R.lds_tiles_small = tiles_lds(SMALL_REGS, SMALL_CHOICES, TL);
    R.lds_tiles_mid = tiles_lds(MID_REGS, MID_CHOICES, TL);
    R.lds_points_big = (size_t)P.max_regs * WAVE * 4;
    R.lds_normals_big = (size_t)P.max_regs * WAVE * 16;
    R.lds_normals_small = (size_t)32 * WAVE * 16;
    // A register file that does not fit LDS (more than ~160 registers for the gradients, ~280 for the intervals) lives in HBM:
    // the reference spills registers beyond its file to memory slots (compiler/alloc.rs:116-125), this is the device's form of
    // it - the root-sized kernel variants take a region of `gscratch` per workgroup instead of LDS.  A slow path by design
    // (a few hundred workgroups, no pipelining: render3d_part), for tapes the fast paths cannot take anyway.
    S.gscratch = nullptr; S.gscratch_stride = 0;
    R.big_hbm = R.lds_tiles_big > FH_LDS_MAX || R.lds_normals_big > FH_LDS_MAX || R.lds_points_big > FH_LDS_MAX;
    if (R.big_hbm) {
        const size_t stride = (std::max(std::max(R.lds_tiles_big, R.lds_normals_big), R.lds_points_big) + 255) & ~(size_t)255;
        if (stride >= ((size_t)1 << 31)) return fail(ctx, FHIP_ERR_UNSUPPORTED, "register file too large");
        R.hbm_waves = (uint32_t)std::max<size_t>(64, std::min<size_t>((size_t)ctx->n_cu * 4, ((size_t)1 << 30) / stride));
        HIP_TRY(ctx, ctx->gscratch.ensure((size_t)R.hbm_waves * stride));
        S.gscratch = (char*)ctx->gscratch.p; S.gscratch_stride = (uint32_t)stride;
        R.lds_tiles_big = R.lds_normals_big = R.lds_points_big = 0;      // (no dynamic LDS for those launches; grids: blocks_big)
    }

    // pre-pass: with >= 3 levels the two coarsest levels are evaluated for all z-slabs at once
    S.n_slabs = R.n_slabs;
    S.frame_stamp = ++ctx->frame_stamp;
    S.pre_levels = prepass_ok ? 2 : 0;

    // root-tile layers of this part: layer k of the block split belongs to iz = k * nz / n_layers (iz = nz - 1: the front);
    // its z-slabs are those that hold one of its layers (a slab shared with another part has work for this part's layers only)
    uint32_t layer_lo = 0, layer_hi = n_layers;
    if (part.nz > 1) {
        layer_lo = n_layers; layer_hi = 0;
        for (uint32_t k = 0; k < n_layers; k++)
            if ((uint64_t)k * part.nz / n_layers == part.iz) { layer_lo = std::min(layer_lo, k); layer_hi = std::max(layer_hi, k + 1); }
        if (layer_lo >= layer_hi) layer_lo = layer_hi = 0;   // more parts than layers: nothing to do
    }
    R.slab_lo = layer_lo / SL; R.slab_hi = (layer_hi + SL - 1) / SL;
    if (!S.pre_levels) { R.slab_lo = layer_lo; R.slab_hi = layer_hi; }
    // root groups: runs of <= TL root tiles of this part, index = first + lane * stride (one set per slab in pre-pass mode)
    struct Run { uint32_t first, n, stride; };
    std::vector<Run> runs;
    if (part.nx > 1 || part.ny > 1) {       // a block of root-tile columns: per x, the run of its y range (x-major numbering)
        for (uint32_t tx = 0; tx < P.roots_x; tx++) {
            if ((uint64_t)tx * part.nx / P.roots_x != part.ix) continue;
            uint32_t y0 = P.roots_y, y1 = 0;
            for (uint32_t ty = 0; ty < P.roots_y; ty++)
                if ((uint64_t)ty * part.ny / P.roots_y == part.iy) { y0 = std::min(y0, ty); y1 = std::max(y1, ty + 1); }
            for (uint32_t ty = y0; ty < y1; ty += TL) runs.push_back(Run{tx * P.roots_y + ty, std::min<uint32_t>(TL, y1 - ty), 1});
        }
    } else {
        std::vector<uint32_t> mine;
        for (uint32_t ri = shard; ri < P.roots_x * P.roots_y; ri += n_shards) mine.push_back(ri);
        for (size_t i = 0; i < mine.size(); i += TL) runs.push_back(Run{mine[i], (uint32_t)std::min<size_t>(TL, mine.size() - i), n_shards});
    }
    FhTapeRef root{0, (uint32_t)t.ops.size(), (uint16_t)t.n_regs, (uint16_t)t.n_choices};
    const uint32_t q0_layers = S.pre_levels ? layer_hi - layer_lo : 1;
    for (uint32_t k = 0; k < q0_layers; k++)
        for (const Run& r : runs) {
            FhGroup g{};
            g.tape = root;
            g.first = r.first; g.n = r.n; g.stride = r.stride;
            g.z = (layer_hi - 1 - k) * ts[0];  // front layers first
            R.roots.push_back(g);
        }
    R.groups_per_slab = (uint32_t)(R.roots.size() / std::max<uint32_t>(q0_layers, 1));
    if (layer_lo >= layer_hi) { R.roots.clear(); R.groups_per_slab = 0; }

    // capacities (exact upper bounds): queue[l] holds the tiles of size ts[l-1] that can be
    // ambiguous, per slab for the per-slab levels and for the whole volume for pre-pass levels
    uint32_t qcaps[FH_MAX_LEVELS] = {0};
    qcaps[0] = std::max<uint32_t>((uint32_t)R.roots.size(), 1);
    for (size_t l = 1; l < ts.size(); l++) {
        const uint64_t tp = ts[l - 1];
        uint64_t c = (uint64_t)((P.width + tp - 1) / tp) * ((P.height + tp - 1) / tp) * (is3d ? P.slab / tp : 1);
        if (l < S.pre_levels) c *= R.n_slabs;
        qcaps[l] = (uint32_t)std::max<uint64_t>(c, 1);
    }
    const uint64_t tl = ts.back();
    const uint64_t fw = (P.width + tl - 1) / tl, fhh = (P.height + tl - 1) / tl;
    const uint64_t leaf_cap = fw * fhh * (is3d ? P.slab / tl : 1);
    R.table_words = is3d ? (uint32_t)leaf_cap : 0;
    R.n_footprints = (uint32_t)(fw * fhh);

    HIP_TRY(ctx, ctx->state.ensure(4 * sizeof(FhRenderState)));
    { void* const before = ctx->arena.p; HIP_TRY(ctx, ctx->arena.ensure(ctx->arena_bytes)); if (ctx->arena.p != before) ctx->resident_serial = 0; }
    for (size_t l = 0; l < ts.size(); l++) HIP_TRY(ctx, ctx->queue[l].ensure((size_t)qcaps[l] * sizeof(FhGroup)));
    if (S.pre_levels) HIP_TRY(ctx, ctx->squeue.ensure((size_t)qcaps[S.pre_levels] * R.n_slabs * sizeof(FhGroup)));
    HIP_TRY(ctx, ctx->leaves.ensure(leaf_cap * sizeof(FhLeaf)));
    const size_t extra = std::min<uint32_t>(ctx->slab_contexts, std::max<uint32_t>(R.n_slabs, 1)) - 1;      // (slab contexts beyond the first)
    if (is3d) HIP_TRY(ctx, ctx->leaves_b.ensure(extra * leaf_cap * sizeof(FhLeaf)));
    if (is3d) {
        if (P.width > 65535 || P.height > 65535) return fail(ctx, FHIP_ERR_UNSUPPORTED, "3D renders support images up to 65535 x 65535");
        // (the assembly leaf and normals kernels address the z-buffer as base + a 32-bit byte offset of 8 bytes per pixel)
        if ((uint64_t)P.width * P.height >= ((uint64_t)1 << 29)) return fail(ctx, FHIP_ERR_UNSUPPORTED, "3D renders support images of fewer than 2^29 pixels");
        HIP_TRY(ctx, ctx->leaf_table.ensure(leaf_cap * sizeof(FhLeafRef)));
        HIP_TRY(ctx, ctx->leaf_table_b.ensure(extra * leaf_cap * sizeof(FhLeafRef)));
        HIP_TRY(ctx, ctx->zbuf.ensure((size_t)P.width * P.height * 8));
        HIP_TRY(ctx, ctx->normals.ensure((size_t)P.width * P.height * 12));
        HIP_TRY(ctx, ctx->fp_lists.ensure((size_t)R.n_footprints * 4 * 3));
        HIP_TRY(ctx, ctx->fp_lists_b.ensure(extra * (size_t)R.n_footprints * 4 * 3));
        size_t mind_words = 0;
        for (size_t l = 0; l < ts.size(); l++) mind_words += (size_t)((P.width + ts[l] - 1) / ts[l]) * ((P.height + ts[l] - 1) / ts[l]);
        HIP_TRY(ctx, ctx->mind.ensure(mind_words * 4));
        R.mind_words = mind_words;   // (cleared - empty image: nothing occluded - by the frame's first launch, upload_frame)
        uint32_t* mp = (uint32_t*)ctx->mind.p;
        for (size_t l = 0; l < ts.size(); l++) {
            S.mind[l] = mp;
            mp += (size_t)((P.width + ts[l] - 1) / ts[l]) * ((P.height + ts[l] - 1) / ts[l]);
        }
        for (int c = 0; c < 3; c++) S.fp_list[c] = (uint32_t*)ctx->fp_lists.p + (size_t)c * R.n_footprints;
    }
    S.arena = (uint64_t*)ctx->arena.p;
    S.arena_cap = (uint32_t)std::min<size_t>(ctx->arena_bytes / 8 - 64, 0x7FFFFFE0u);  // slack: the interpreters prefetch up to 12 ops past a tape's end
    S.arena_head = S.arena_root_end = (uint32_t)t.ops.size();
    S.arena_overflow = 0;
    for (int l = 0; l < FH_MAX_LEVELS; l++) {
        S.queue[l] = (FhGroup*)ctx->queue[l].p;
        S.count[l] = S.cursor[l] = S.count_big[l] = S.cursor_big[l] = 0;
    }
    S.count_big[0] = (uint32_t)R.roots.size();  // the root tape always takes the large LDS layout
    for (size_t l = 0; l < ts.size(); l++) S.qcap[l] = qcaps[l];
    R.split = ctx->use_split && R.tl == 64 && (is3d || !ctx->opt.no_split_2d);
    // (tapes with sin cos tan asin acos atan exp ln: the *_t variants of the tile kernels, which carry those interval handlers;
    // atan2, mod, mix, rand keep the HIP tile stage)
    R.asm_tiles_t = !tape_asm_ok(t) && tape_tiles_t_ok(t) && !ctx->opt.no_asm_tiles_t;
    // (not with a register file in HBM: the assembly tile kernels - fh_prune1, the groups path and the linked prune with them - keep
    // registers AND choices in LDS, and a tape of few registers can still outgrow it by its choices alone, ~5 600 of them)
    R.asm_tiles = R.split && ctx->use_asm && !ctx->opt.no_asm_tiles && (tape_asm_ok(t) || R.asm_tiles_t) && t.n_regs <= 128 && !R.big_hbm;
    R.asm_tiles_t = R.asm_tiles_t && R.asm_tiles;
    // levels whose forward pass exports its choices to the one-wave-per-child prune (fh_prune1): long tapes, few parents.
    // 3D: of the pre-pass levels, level 0 (measured); 2D: level 0
    {
        const uint32_t p1_levels = (uint32_t)std::max(0, ctx->opt.prune1_levels);
        R.exp_levels = is3d ? std::min(S.pre_levels, p1_levels) : std::min(1u, p1_levels);
    }
    R.prune1 = R.asm_tiles && !R.asm_tiles_t && R.exp_levels > 0 && !ctx->opt.no_prune1;      // (the *_t kernels have no export mode)
    // tape parallelism: level 0 evaluates the root tree's terms as independent groups on different
    // waves, then the tree itself; the prune sees the root tape with its usual choices
    R.groups = R.prune1 && !tape->tgroups.empty() && !ctx->opt.no_tape_groups;
    S.n_tgroups = 0;
    if (R.groups) {
        uint32_t off = (uint32_t)t.ops.size() + 16, mr = 1, mc = 0;
        for (size_t g = 0; g < tape->tgroups.size(); g++) {
            const fh::HostTape& gt = tape->tgroups[g];
            S.tgroup[g] = FhTapeRef{off, (uint32_t)gt.ops.size(), (uint16_t)gt.n_regs, (uint16_t)gt.n_choices};
            off += (uint32_t)gt.ops.size() + 16;  // slack: the interpreters prefetch past a tape's end
            mr = std::max(mr, gt.n_regs); mc = std::max(mc, gt.n_choices);
        }
        R.group_regs = mr; R.group_choices = mc;
        R.lds_tiles_group = tiles_lds(mr, mc, TL);
        if (mr <= 128 && R.lds_tiles_group <= FH_LDS_MAX && (size_t)off * 8 + 4096 <= ctx->arena_bytes) {
            S.n_tgroups = (uint32_t)tape->tgroups.size();
            S.n_terms = tape->plan.n_terms; S.n_top = (uint32_t)tape->plan.top.size(); S.top_chain = tape->plan.chain ? 1 : 0;
            S.troot_len = (uint32_t)t.ops.size(); S.troot_choices = t.n_choices; S.troot_regs = std::max<uint32_t>(t.n_regs, 1);
            S.arena_head = S.arena_root_end = off;
            std::lock_guard<std::mutex> guard(tape->upload_lock);
            if (tape->device >= 0 && tape->device != ctx->device) return fail(ctx, FHIP_ERR_UNSUPPORTED, "this tape's device copies belong to another device: build the tape per device");
            tape->device = ctx->device;
            if (!tape->d_top) {
                static_assert(sizeof(FhTopOp) == sizeof(fh::TopOp), "top op layout");
                HIP_TRY(ctx, hipMalloc((void**)&tape->d_top, tape->plan.top.size() * sizeof(FhTopOp)));
                HIP_TRY(ctx, hipMemcpy(tape->d_top, tape->plan.top.data(), tape->plan.top.size() * sizeof(FhTopOp), hipMemcpyHostToDevice));
                HIP_TRY(ctx, hipMalloc((void**)&tape->d_chsrc, std::max<size_t>(tape->plan.choice_src.size(), 1) * 4));
                HIP_TRY(ctx, hipMemcpy(tape->d_chsrc, tape->plan.choice_src.data(), tape->plan.choice_src.size() * 4, hipMemcpyHostToDevice));
            }
            S.ttop = tape->d_top; S.chsrc = tape->d_chsrc;
            // the linked prune of the root level (option prune2; prune2.hip): links of the root tape, made once with it.  0.275 ms
            // against fh_prune1's 0.344 per 1024^3 frame of prospero.vm (a child of that root tape keeps ~580 ops, up to 1011);
            // fh_prune1 stays behind it for the children it leaves marked (more than 64 registers / FH_P2_MAX_KEPT ops)
            if (ctx->opt.prune2 && !tape->links_tried) {
                tape->links_tried = true;
                std::vector<uint64_t> lk;
                std::vector<uint64_t> cops;
                if (fh::compute_links(t, lk, cops)) {
                    // (published together or not at all: a failure half way must not leave links without their choice table)
                    uint64_t *dl = nullptr, *dc = nullptr;
                    hipError_t e = hipMalloc((void**)&dl, lk.size() * 8);
                    if (e == hipSuccess) e = hipMemcpy(dl, lk.data(), lk.size() * 8, hipMemcpyHostToDevice);
                    if (e == hipSuccess) e = hipMalloc((void**)&dc, std::max<size_t>(cops.size(), 1) * 8);
                    if (e == hipSuccess) e = hipMemcpy(dc, cops.data(), cops.size() * 8, hipMemcpyHostToDevice);
                    if (e != hipSuccess) {
                        if (dl) (void)hipFree(dl);
                        if (dc) (void)hipFree(dc);
                        HIP_TRY(ctx, e);
                    }
                    tape->d_links = dl; tape->d_ctab = dc;
                }
            }
            R.lds_prune2 = (((size_t)t.ops.size() * 8 + 15) & ~(size_t)15) + (size_t)FH_P2_WPB * fh_p2_wave_lds(t.n_choices);
            // (one workgroup of FH_P2_WPB children per CU: beyond two rounds of them - 2048^3 has 4 096 root tiles - the scalar sweep,
            // whose waves all fit the machine at once, is the faster one again: 2.09 against 2.17 ms per frame)
            R.prune2 = tape->d_links && tape->d_ctab && ctx->opt.prune2 && t.ops.size() <= FH_P2_MAX_OPS && t.n_choices <= FH_P2_MAX_CHOICES &&
                       R.lds_prune2 <= FH_LDS_MAX && R.roots.size() * 64 <= (size_t)2 * ctx->n_cu * FH_P2_WPB;      // (a root group = up to 64 root tiles)
            R.d_ctab = tape->d_ctab;
            R.d_links = tape->d_links;
            R.lds_prune2_l1 = (((size_t)FH_P2_L1_OPS * 8 + 15) & ~(size_t)15) + (size_t)FH_P2_L1_WPB * fh_p2_wave_lds(FH_P2_L1_CHOICES, FH_P2_L1_OPS);
            R.prune2_l1 = R.prune2 && is3d && S.pre_levels > 1 && ctx->opt.prune2_l1 && !ctx->opt.no_tiles_v && R.exp_levels <= 1 &&
                          R.lds_prune2_l1 <= FH_LDS_MAX;
            const size_t blocks = qcaps[0];
            HIP_TRY(ctx, ctx->tvals.ensure(blocks * S.n_terms * WAVE * 8));
            HIP_TRY(ctx, ctx->topch.ensure(blocks * S.n_top * WAVE));
            HIP_TRY(ctx, ctx->chwr.ensure(blocks * S.n_tgroups * ((t.n_choices + 15) / 16) * WAVE * 4 + 256));
            S.tvals = (float*)ctx->tvals.p; S.topch = (uint8_t*)ctx->topch.p; S.chwr = (uint32_t*)ctx->chwr.p;
        } else R.groups = false;
    }
    if (R.prune1) {  // choice words of the pre-pass levels' forward passes: [slot][word][lane]
        uint32_t cap = 1;
        for (uint32_t l = 0; l < std::max(S.pre_levels, R.exp_levels); l++) cap = std::max(cap, qcaps[l] * (l == 0 && R.groups ? S.n_tgroups : 1u));
        const size_t words[2] = {(SMALL_CHOICES + 15) / 16, ((size_t)P.max_choices + 15) / 16};
        for (int k = 0; k < 2; k++) {
            HIP_TRY(ctx, ctx->chw[k].ensure(std::max<size_t>(cap * words[k] * 256, 256)));
            S.chw[k] = (uint32_t*)ctx->chw[k].p;
        }
    }
    if (R.split) {
        uint32_t cap = 1;
        for (size_t l = 0; l < ts.size(); l++) cap = std::max(cap, qcaps[l] * (l == 0 && R.groups ? S.n_tgroups : 1u));
        for (int k = 0; k < 2; k++) {
            HIP_TRY(ctx, ctx->slots[k].ensure((size_t)cap * sizeof(FhSlot)));
            S.slots[k] = (FhSlot*)ctx->slots[k].p;
            S.slot_cap[k] = cap;
        }
    }
    S.squeue = (FhGroup*)ctx->squeue.p;
    S.squeue_cap = qcaps[S.pre_levels];
    S.arena_frame_end = S.arena_root_end;
    for (int k = 0; k < FH_MAX_SLABS; k++) S.scount[k] = S.scount_big[k] = 0;
    S.queue_overflow = 0;
    S.leaves = (FhLeaf*)ctx->leaves.p;
    S.leaf_cap = (uint32_t)leaf_cap;
    S.n_leaves = S.leaf_cursor = S.leaf_cursor_big = S.normal_cursor = S.normal_cursor_big = 0;
    S.leaf_table = (FhLeafRef*)ctx->leaf_table.p;
    for (int c = 0; c < 3; c++) S.fp_count[c] = S.fp_cursor[c] = 0;
    S.zbuf = (uint64_t*)ctx->zbuf.p;
    S.normals = (float*)ctx->normals.p;
    S.image2d = nullptr;
    memset(S.stat, 0, sizeof(S.stat));
    memset(S.leaf_stat, 0, sizeof(S.leaf_stat));
    S.want_stats = (ctx->profiling || ctx->probe || ctx->opt.stats) ? 1 : 0;
    if (((size_t)t.ops.size() + 64) * 8 > ctx->arena_bytes) return fail(ctx, FHIP_ERR_UNSUPPORTED, "tape larger than the arena");
    // level-0 groups sit at the back of queue[0] (the "big" half), in reverse order
    std::reverse(R.roots.begin(), R.roots.end());
    return FHIP_OK;
}

static int blocks_for(const fhip_ctx* ctx, size_t lds, int max_per_cu);
// ... of a root-sized launch: as many workgroups as LDS lets run, or the number of HBM register-file regions
static int blocks_big(const fhip_ctx* ctx, const RenderSetup& R, size_t lds, int max_per_cu) {
    return R.big_hbm ? (int)R.hbm_waves : blocks_for(ctx, lds, max_per_cu);
}
static int blocks_for(const fhip_ctx* ctx, size_t lds, int max_per_cu) {
    int per_cu = lds ? (int)std::min<size_t>((size_t)max_per_cu, FH_LDS_MAX / std::max<size_t>(lds, 1)) : max_per_cu;
    per_cu = std::max(per_cu, 1);
    return ctx->n_cu * per_cu;
}

static fhip_status finish_render(fhip_ctx* ctx) {
    HIP_TRY(ctx, hipMemcpyAsync(&ctx->last_state, ctx->state.p, sizeof(FhRenderState), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (uint32_t k = 1; k < ctx->forked; k++) {  // the other slab contexts keep their own counters
        HIP_TRY(ctx, hipMemcpy(&ctx->last_state_b, (char*)ctx->state.p + k * sizeof(FhRenderState), sizeof(FhRenderState), hipMemcpyDeviceToHost));
        ctx->last_state.queue_overflow += ctx->last_state_b.queue_overflow;
        ctx->last_state.arena_overflow += ctx->last_state_b.arena_overflow;
        for (int i = 0; i < 64; i++) ctx->last_state.stat[i] += ctx->last_state_b.stat[i];
        for (int i = 0; i < 8; i++) ctx->last_state.leaf_stat[i] += ctx->last_state_b.leaf_stat[i];
    }
    ctx->have_last_state = true;
    if (ctx->last_state.queue_overflow) return fail(ctx, FHIP_ERR_OVERFLOW, "device work queue overflow");
    return FHIP_OK;
}

struct FrameClear { void* p = nullptr; size_t bytes = 0; uint32_t fill = 0; };   // a buffer the frame starts from cleared (bytes: a multiple of 4)
static fhip_status upload_frame(fhip_ctx* ctx, const fhip_tape* tape, RenderSetup& R, const FrameClear (&clear)[3]) {
    // The frame's state and root groups go through pinned staging slots (a ring of eight, each guarded by an event): a copy from
    // pageable memory would make the host wait for everything queued on the stream before it, i.e. for the previous frame.
    const size_t roots_bytes = R.roots.size() * sizeof(FhGroup);
    fhip_ctx::Staging& sg = ctx->staging[ctx->staging_next++ % 8];
    if (sg.ev && sg.used) HIP_TRY(ctx, hipEventSynchronize(sg.ev));
    if (!sg.ev) HIP_TRY(ctx, hipEventCreateWithFlags(&sg.ev, hipEventDisableTiming));
    if (sg.cap < sizeof(FhRenderState) + roots_bytes) {
        if (sg.p) (void)hipHostFree(sg.p);
        sg.p = nullptr; sg.cap = 0;
        HIP_TRY(ctx, hipHostMalloc(&sg.p, sizeof(FhRenderState) + roots_bytes + 4096, hipHostMallocDefault));
        sg.cap = sizeof(FhRenderState) + roots_bytes + 4096;
    }
    memcpy(sg.p, &R.S, sizeof(FhRenderState));
    if (roots_bytes) memcpy((char*)sg.p + sizeof(FhRenderState), R.roots.data(), roots_bytes);
    // The root tape and its groups sit below arena_root_end, where no frame writes: a shape rendered
    // again finds them there (17 small copies, 0.1 ms of a 4 ms frame, otherwise).
    if (ctx->resident_serial != tape->serial || ctx->resident_groups != R.S.n_tgroups) {
        ctx->resident_serial = 0;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->arena.p, tape->t.ops.data(), tape->t.ops.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        for (uint32_t g = 0; g < R.S.n_tgroups; g++)  // the group tapes follow the root tape
            HIP_TRY(ctx, hipMemcpyAsync((uint64_t*)ctx->arena.p + R.S.tgroup[g].off, tape->tgroups[g].ops.data(),
                                        tape->tgroups[g].ops.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        ctx->resident_serial = tape->serial;
        ctx->resident_groups = R.S.n_tgroups;
    }
    // state, root groups and the cleared buffers in one launch (k_frame_begin reads the pinned slot itself)
    static_assert(sizeof(FhRenderState) % 4 == 0 && sizeof(FhGroup) % 4 == 0, "copied as 32-bit words");
    FhFrameBegin fb;
    memset(&fb, 0, sizeof(fb));
    fb.state_dst = (uint32_t*)ctx->state.p; fb.state_src = (const uint32_t*)sg.p; fb.state_words = (uint32_t)(sizeof(FhRenderState) / 4);
    if (!R.roots.empty()) {
        fb.roots_dst = (uint32_t*)((FhGroup*)ctx->queue[0].p + (R.S.qcap[0] - R.roots.size()));
        fb.roots_src = (const uint32_t*)((const char*)sg.p + sizeof(FhRenderState));
        fb.roots_words = (uint32_t)(roots_bytes / 4);
    }
    size_t most = 0;
    for (int k = 0; k < 3; k++) {
        fb.clear[k] = (uint32_t*)clear[k].p; fb.clear_words[k] = clear[k].bytes / 4; fb.fill[k] = clear[k].fill;
        if (clear[k].p) most = std::max(most, clear[k].bytes);
    }
    const unsigned blocks = (unsigned)std::max<size_t>(2, std::min<size_t>((size_t)ctx->n_cu * 8, (most + 256 * 64 - 1) / (256 * 64)));
    hipLaunchKernelGGL(k_frame_begin, dim3(blocks), dim3(256), 0, ctx->stream, fb);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(sg.ev, ctx->stream));
    sg.used = true;
    for (auto& e : ctx->prof_events) { (void)hipEventDestroy(e.second.first); (void)hipEventDestroy(e.second.second); }
    ctx->prof_events.clear();
    for (auto& e : ctx->asm_events) { (void)hipEventDestroy(e.second.first); (void)hipEventDestroy(e.second.second); }
    ctx->asm_events.clear();
    return FHIP_OK;
}

// One level of the tile hierarchy: the small-LDS variant for the bulk of the groups and the
// root-sized variant for the few large tapes (both always launched; empty queues exit at once).
#define FH_LAUNCH_TILES(IS3D, FULL, BIG, grid, lds)                                                                  \
    do {                                                                                                            \
        if (R.tl == 64) hipLaunchKernelGGL((k_tiles<IS3D, FULL, BIG, 64>), dim3(grid), dim3(WAVE), lds, ctx->stream, dS, level); \
        else hipLaunchKernelGGL((k_tiles<IS3D, FULL, BIG, 16>), dim3(grid), dim3(WAVE), lds, ctx->stream, dS, level);            \
    } while (0)
// 3D tile stage of one level as three kernels (see kernels.hip "Split 3D tile stage")
static void launch_tiles_split(fhip_ctx* ctx, const RenderSetup& R, FhRenderState* dS, int level, bool is3d) {
    // Persistent waves with a static round robin over the parents.  (FHIP_ONE_EACH_TILES=1: one short
    // workgroup per parent instead - measured slower in the pipelined frame: the tile stage then
    // takes more of the machine from the leaf kernel it overlaps with.)
    const uint32_t one_each = ctx->opt.one_each_tiles ? std::min<uint32_t>(R.S.qcap[level], 1u << 20) : 0u;
    const int gs = one_each ? (int)one_each : blocks_for(ctx, R.lds_tiles_small, 8);
    const int gb = one_each && !R.big_hbm ? (int)one_each : blocks_big(ctx, R, R.lds_tiles_big, 8);
    const int gp = one_each ? (int)one_each : ctx->n_cu * 8;
    launch(ctx, FHIP_K_TILES, [&] {
        // (pre-pass levels below the root: the children of a parent shared out over several slots - tsetup_body; option
        // l1_split: 0 chosen on the device from the number of parents, 1 off, 2 / 4 / 8 fixed)
        const uint32_t csplit = (level > 0 && (uint32_t)level < R.S.pre_levels && R.asm_tiles) ? (uint32_t)std::max(0, std::min(8, ctx->opt.l1_split)) : 1u;
        if (is3d) hipLaunchKernelGGL(k_tsetup3d, dim3(gp), dim3(WAVE), 0, ctx->stream, dS, level, csplit);
        else hipLaunchKernelGGL(k_tsetup2d, dim3(gp), dim3(WAVE), 0, ctx->stream, dS, level);
    });
    if (R.groups && level == 0) {
        // Tape parallelism: the root tree's terms by independent groups, one wave per (block of root
        // tiles, group) -> the tree over the terms (result, marks, arena) -> the root tape's choice words
        // gathered from both -> one wave per ambiguous child prunes the root tape -> push.
        launch(ctx, FHIP_K_TILES, [&] {
            struct { FhRenderState* S; uint32_t level, big, max_regs, max_choices, n_waves, flags, skip_regs, skip_choices; } ka;
            const int gg = blocks_for(ctx, R.lds_tiles_group, 8);
            ka.S = dS; ka.level = 0; ka.big = 1; ka.max_regs = R.group_regs; ka.max_choices = R.group_choices;
            ka.n_waves = (uint32_t)gg; ka.flags = (ctx->probe ? 1u : 0u) | 2u | 4u | 8u; ka.skip_regs = ka.skip_choices = 0;
            (void)launch_asm(ctx, FH_ASM_TILES, (uint32_t)gg, &ka, sizeof(ka), R.lds_tiles_group);
            const uint32_t blocks = R.S.qcap[0], root_words = (R.S.troot_choices + 15) / 16, group_words = (R.group_choices + 15) / 16;
            if (R.S.top_chain) hipLaunchKernelGGL(k_tchain3d, dim3(WAVE, blocks), dim3(WAVE), 0, ctx->stream, dS);
            else hipLaunchKernelGGL(k_ttop3d, dim3(blocks), dim3(WAVE), 0, ctx->stream, dS);
            hipLaunchKernelGGL(k_tmark3d, dim3(blocks), dim3(WAVE), 0, ctx->stream, dS);
            if (root_words) hipLaunchKernelGGL(k_tscatter3d, dim3(root_words, blocks), dim3(WAVE), 0, ctx->stream, dS, group_words, root_words);
            if (R.prune2) {
                hipEvent_t ea = nullptr, eb = nullptr;      // (timed under the fh_prune1 slot of the per-kernel profile: it replaces that launch)
                if (ctx->profiling) { (void)hipEventCreate(&ea); (void)hipEventCreate(&eb); (void)hipEventRecord(ea, ctx->stream); }
                hipLaunchKernelGGL(k_prune2, dim3(blocks * FH_P2_PER_SLOT), dim3(FH_P2_WPB * 64), R.lds_prune2, ctx->stream, dS, 0u, 1u, 2u, root_words,
                                   (const uint2*)R.d_links, (const uint2*)R.d_ctab, (R.prune2_l1 ? 1u : 0u) | (ctx->opt.prune2_probe_level == 0 ? 2u : 0u), R.S.troot_len, R.S.troot_choices,
                                   (uint32_t)FH_P2_MAX_KEPT);
                // ... and the scalar sweep behind it for the children it left marked (more than 64 registers or FH_P2_MAX_KEPT kept ops:
                // none for the models here; a wave whose child is done leaves at once)
                struct { FhRenderState* S; uint32_t level, big, max_choices, mode; } kp = {dS, 0, 1, R.S.troot_choices, 2};
                size_t kp_bytes = sizeof(kp);
                void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &kp, HIP_LAUNCH_PARAM_BUFFER_SIZE, &kp_bytes, HIP_LAUNCH_PARAM_END};
                (void)hipModuleLaunchKernel(ctx->asm_fn[FH_ASM_PRUNE1], blocks * 64, 1, 1, WAVE, 1, 1, 0, ctx->stream, nullptr, extra);
                if (ctx->profiling) { (void)hipEventRecord(eb, ctx->stream); ctx->asm_events.push_back({FH_ASM_PRUNE1, {ea, eb}}); }
            } else {
                struct { FhRenderState* S; uint32_t level, big, max_choices, mode; } kp = {dS, 0, 1, R.S.troot_choices, 2};
                (void)launch_asm(ctx, FH_ASM_PRUNE1, blocks * 64, &kp, sizeof(kp));
            }
        });
    } else if (R.asm_tiles) {
        launch(ctx, FHIP_K_TILES, [&] {
            // pre-pass levels: long tapes, few parents -> the forward pass exports its choices and
            // the prune runs as one wave per child (fh_prune1)
            const bool exp = R.prune1 && (uint32_t)level < R.exp_levels;      // level 0 only: 8 parents, 6363-op tape (measured)
            const int K_TILES = R.asm_tiles_t ? FH_ASM_TILES_T : FH_ASM_TILES;
            struct { FhRenderState* S; uint32_t level, big, max_regs, max_choices, n_waves, flags, skip_regs, skip_choices; } ka;
            ka.S = dS; ka.level = (uint32_t)level; ka.flags = (ctx->probe ? 1u : 0u) | (exp ? 2u : 0u);
            ka.skip_regs = ka.skip_choices = 0;
            // Pre-pass levels below the root: the small-layout parents and the others are different slot lists;
            // their launches run side by side (second stream) instead of one after the other.
            // (Only for a frame alone, whose coarse levels are on the caller's stream: in a pipelined frame they are off the critical
            // path, and the side stream carries the previous frame's tile chains, where this frame's level-1 kernel sat for 170 us
            // of every frame - 1.64 -> 1.60 ms without the fork.  Forking to the tail stream instead: 2.0 ms; to streams of their
            // own, also for the per-slab levels' nearly always empty big-list launches: 3.5 ms - streams beyond four share
            // hardware queues (GPU_MAX_HW_QUEUES) and serialise against each other.)
            // (A fifth stream for the per-slab levels' nearly always empty big-list launches, with GPU_MAX_HW_QUEUES=8 in the
            // environment: 2.3 ms per frame instead of 1.03 - more than four streams in flight cost far more than two kernel
            // boundaries per slab, whatever the number of hardware queues.)
            hipStream_t const rest_stream = ctx->stream2;
            const bool side = level > 0 && (uint32_t)level < R.S.pre_levels && ctx->use_pipeline && !ctx->profiling && rest_stream &&
                              ctx->stream != rest_stream && ctx->stream != ctx->stream_pre && !ctx->opt.pipe_serial && is3d;
            hipStream_t const big_stream = side ? rest_stream : nullptr;
            // Tapes of <= 32 registers / 256 choices (the small slot list: every parent of the leaf level) and, from the other
            // list, those of <= 64 / 512 go to the kernels that keep the interval file, the choices and the prune's register
            // map in VGPRs (fh_tiles_v32: 16 waves per CU, fh_tiles_v64: 8; no LDS); what is left takes the LDS layouts.
            const bool use_v = !ctx->opt.no_tiles_v;
            const bool vk = use_v && !exp;
            bool both_lists = false;
            if (level > 0) {
                ka.big = 0; ka.max_regs = SMALL_REGS; ka.max_choices = SMALL_CHOICES; ka.n_waves = (uint32_t)gs;
                if (side) {
                    (void)hipEventRecord(ctx->ev_rest_fork, ctx->stream);
                    (void)hipStreamWaitEvent(rest_stream, ctx->ev_rest_fork, 0);
                }
                // (a pre-pass level has a few hundred parents in the two lists together: fh_tiles_v64 takes both in ONE launch
                // below - the level's time is its slowest parent's either way, and a launch of its own for the small list put
                // another 130 us on the coarse levels' chain)
                both_lists = vk && (uint32_t)level < R.S.pre_levels && !side && !ctx->opt.no_both_lists;
                if (vk && !both_lists) {
                    const int v32_waves = ctx->opt.v32_waves;
                    ka.n_waves = one_each ? one_each : (uint32_t)(ctx->n_cu * v32_waves);
                    (void)launch_asm(ctx, R.asm_tiles_t ? FH_ASM_TILES_V32_T : FH_ASM_TILES_V32, ka.n_waves, &ka, sizeof(ka));
                } else if (vk) {
                } else
                    (void)launch_asm(ctx, K_TILES, (uint32_t)gs, &ka, sizeof(ka), R.lds_tiles_small);
            }
            ka.big = 1;
            if (exp) ka.flags |= ((R.S.P.max_choices + 15) / 16) << 16;  // one stride in chw[1] for the medium and the large layout
            bool rest = true;   // anything left for the root-sized LDS layout?
            // (leaving the per-slab levels' big-list parents to the root-sized LDS launch alone - one launch less on the slab's tile
            // chain - was measured: 1.02 vs 1.04 ms per frame, within the noise; not done)
            if (vk && level > 0) {
                const int v64_waves = ctx->opt.v64_waves;
                // (per-slab levels: the parents' tapes fit fh_tiles_v32 but for a rare one - an empty launch of 2048 waves of 176
                // VGPRs each, queued behind the leaf kernel of the slab in front, was measured to hold the tile chain up for
                // 130 us: a small persistent grid there)
                const int v64_slab_waves = ctx->opt.v64_slab_waves;
                const bool per_slab = (uint32_t)level >= R.S.pre_levels && R.S.pre_levels > 0;
                ka.max_regs = V64_REGS; ka.max_choices = V64_CHOICES;
                ka.n_waves = one_each ? one_each : (per_slab ? (uint32_t)v64_slab_waves : (uint32_t)(ctx->n_cu * v64_waves));
                if (both_lists) ka.flags |= 16u;
                // (level 1 with the linked prune: parents whose tape carries this frame's links get their choices exported - chw[1]
                // with this stride, chw[0] with 16 words - and their children marked for k_prune2 below; the others are pruned here)
                const bool linked = R.prune2_l1 && !per_slab;
                const uint32_t plain_flags = ka.flags;
                if (linked) ka.flags = (ka.flags & 0xFFFFu) | 2u | (((R.S.P.max_choices + 15) / 16) << 16);
                (void)launch_asm(ctx, R.asm_tiles_t ? FH_ASM_TILES_V64_T : FH_ASM_TILES_V64, ka.n_waves, &ka, sizeof(ka), 0, 1, big_stream);
                ka.flags = plain_flags & ~16u;
                ka.skip_regs = V64_REGS; ka.skip_choices = V64_CHOICES;
                rest = R.S.P.max_regs > V64_REGS || R.S.P.max_choices > V64_CHOICES;
            }
            // Pre-pass levels below the root: a few hundred parents whose tapes are far smaller than the
            // root's.  With the root-sized LDS layout only one wave fits a CU (256 at a time); a medium
            // layout takes those that fit it three to a CU, the root-sized launch takes the rest.
            const bool mid = !(vk && level > 0) && level > 0 && (uint32_t)level < R.S.pre_levels && R.lds_tiles_mid * 2 <= R.lds_tiles_big && !ctx->opt.no_mid;
            if (mid) {
                const int gm = one_each ? (int)one_each : blocks_for(ctx, R.lds_tiles_mid, 8);
                ka.max_regs = MID_REGS; ka.max_choices = MID_CHOICES; ka.n_waves = (uint32_t)gm;
                (void)launch_asm(ctx, K_TILES, (uint32_t)gm, &ka, sizeof(ka), R.lds_tiles_mid, 1, big_stream);
                ka.skip_regs = MID_REGS; ka.skip_choices = MID_CHOICES;
            }
            ka.max_regs = R.S.P.max_regs; ka.max_choices = R.S.P.max_choices; ka.n_waves = (uint32_t)gb;
            if (rest) (void)launch_asm(ctx, K_TILES, (uint32_t)gb, &ka, sizeof(ka), R.lds_tiles_big, 1, big_stream);
            if (side) {
                (void)hipEventRecord(ctx->ev_rest_join, rest_stream);
                (void)hipStreamWaitEvent(ctx->stream, ctx->ev_rest_join, 0);
            }
            if (R.prune2_l1 && use_v && level > 0 && (uint32_t)level < R.S.pre_levels) {
                hipEvent_t ea = nullptr, eb = nullptr;      // (slot 7 of the per-kernel profile)
                if (ctx->profiling) { (void)hipEventCreate(&ea); (void)hipEventCreate(&eb); (void)hipEventRecord(ea, ctx->stream); }
                hipLaunchKernelGGL(k_prune2, dim3(ctx->n_cu * 2), dim3(FH_P2_L1_WPB * 64), R.lds_prune2_l1, ctx->stream, dS, (uint32_t)level, 2u, 0u,
                                   (R.S.P.max_choices + 15) / 16, (const uint2*)nullptr, (const uint2*)nullptr, ctx->opt.prune2_probe_level == 1 ? 2u : 0u,
                                   (uint32_t)FH_P2_L1_OPS, (uint32_t)FH_P2_L1_CHOICES, (uint32_t)FH_P2_L1_OPS);
                if (ctx->profiling) { (void)hipEventRecord(eb, ctx->stream); ctx->asm_events.push_back({7, {ea, eb}}); }
            }
            if (exp) {
                struct { FhRenderState* S; uint32_t level, big, max_choices, pad; } kp = {dS, (uint32_t)level, 0, SMALL_CHOICES, 0};
                const uint32_t bound = R.S.qcap[level] * 64;  // 64 waves per possible parent; unmarked children exit at once
                if (level > 0) (void)launch_asm(ctx, FH_ASM_PRUNE1, bound, &kp, sizeof(kp));
                kp.big = 1; kp.max_choices = R.S.P.max_choices;
                (void)launch_asm(ctx, FH_ASM_PRUNE1, bound, &kp, sizeof(kp));
            }
        });
    } else
    launch(ctx, FHIP_K_TILES, [&] {
        if (level > 0) {
            if (R.full) hipLaunchKernelGGL((k_teval3d<true, false>), dim3(gs), dim3(WAVE), R.lds_tiles_small, ctx->stream, dS, level);
            else hipLaunchKernelGGL((k_teval3d<false, false>), dim3(gs), dim3(WAVE), R.lds_tiles_small, ctx->stream, dS, level);
        }
        if (R.full) hipLaunchKernelGGL((k_teval3d<true, true>), dim3(gb), dim3(WAVE), R.lds_tiles_big, ctx->stream, dS, level);
        else hipLaunchKernelGGL((k_teval3d<false, true>), dim3(gb), dim3(WAVE), R.lds_tiles_big, ctx->stream, dS, level);
    });
    // (last level: fewer waves, several parents each - one leaf reservation per wave)
    const int push_mul = ctx->opt.push_waves;
    const int gpush = (level + 1 == (int)R.S.P.n_levels && !one_each) ? ctx->n_cu * push_mul : gp;
    launch(ctx, FHIP_K_TILES, [&] {
        if (is3d) hipLaunchKernelGGL(k_tpush3d, dim3(gpush), dim3(WAVE), 0, ctx->stream, dS, level);
        else {
            hipLaunchKernelGGL(k_tpush2d, dim3(gpush), dim3(WAVE), 0, ctx->stream, dS, level);
            const uint32_t slots_max = R.S.qcap[level] * ((level == 0 && R.groups) ? R.S.n_tgroups : 1u);
            hipLaunchKernelGGL(k_tfill2d, dim3(64, slots_max), dim3(256), 0, ctx->stream, dS, level);
        }
    });
}

static void launch_tiles(fhip_ctx* ctx, const RenderSetup& R, FhRenderState* dS, int level, bool is3d) {
    if (R.split) return launch_tiles_split(ctx, R, dS, level, is3d);
    const int gs = blocks_for(ctx, R.lds_tiles_small, 8), gb = blocks_big(ctx, R, R.lds_tiles_big, 8);
    launch(ctx, FHIP_K_TILES, [&] {
        if (is3d) { if (R.full) FH_LAUNCH_TILES(true, true, true, gb, R.lds_tiles_big); else FH_LAUNCH_TILES(true, false, true, gb, R.lds_tiles_big); }
        else { if (R.full) FH_LAUNCH_TILES(false, true, true, gb, R.lds_tiles_big); else FH_LAUNCH_TILES(false, false, true, gb, R.lds_tiles_big); }
    });
    if (level > 0)
        launch(ctx, FHIP_K_TILES, [&] {
            if (is3d) { if (R.full) FH_LAUNCH_TILES(true, true, false, gs, R.lds_tiles_small); else FH_LAUNCH_TILES(true, false, false, gs, R.lds_tiles_small); }
            else { if (R.full) FH_LAUNCH_TILES(false, true, false, gs, R.lds_tiles_small); else FH_LAUNCH_TILES(false, false, false, gs, R.lds_tiles_small); }
        });
}

fhip_status fhip_render2d(fhip_ctx* ctx, const fhip_tape* tape, const fhip_render2d_config* cfg, float* out,
                          int out_is_device) {
    if (ctx->cancelled.load()) return fail(ctx, FHIP_ERR_CANCELLED, "cancelled");
    RenderSetup R;
    memset(&R.S, 0, sizeof(R.S));
    FhRender& P = R.S.P;
    P.width = cfg->width; P.height = cfg->height; P.depth = 0; P.z = cfg->z; P.pixel_perfect = cfg->pixel_perfect ? 1 : 0;
    fhip_status st = bind_inputs(ctx, tape, cfg->axis_slots, cfg->var_keys, cfg->var_values, cfg->n_vars, P);
    if (st) return st;
    // mat = world_to_model * screen_to_world, lifted to 4x4 preserving Z (pixel.rs:122-124, 281-285)
    const uint32_t size[2] = {cfg->width, cfg->height};
    float s2w[9], m3[9];
    const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    fhip_screen_to_world(size, 2, s2w);
    mat_product(cfg->world_to_model ? cfg->world_to_model : ident, s2w, 3, m3);
    const float m4[16] = {m3[0], m3[1], 0, m3[2], m3[3], m3[4], 0, m3[5], 0, 0, 1, 0, m3[6], m3[7], 0, m3[8]};
    memcpy(P.mat, m4, sizeof(m4));
    const std::vector<uint32_t> ts = cfg->tile_sizes ? trim_tiles(cfg->tile_sizes, cfg->n_tile_sizes, std::max(cfg->width, cfg->height))
                                                     : (ctx->opt.vm_tiles ? trim_tiles(VM_TILES_2D, 3, std::max(cfg->width, cfg->height))
                                                                                : trim_tiles(HIP_TILES_2D, 2, std::max(cfg->width, cfg->height)));
    st = prepare(ctx, tape, false, ts, PartSpec{}, R);
    if (st) return st;
    const size_t npix = (size_t)cfg->width * cfg->height;
    float* d_out = out;
    if (!out_is_device) { HIP_TRY(ctx, ctx->tmp_out.ensure(npix * 4)); d_out = (float*)ctx->tmp_out.p; }
    R.S.image2d = d_out;
    FhRenderState* dS = (FhRenderState*)ctx->state.p;
    const FrameClear no_clear[3] = {};
    st = upload_frame(ctx, tape, R, no_clear);
    if (st) return st;
    for (uint32_t l = 0; l < P.n_levels; l++) {
        if (ctx->cancelled.load()) return fail(ctx, FHIP_ERR_CANCELLED, "cancelled");
        launch_tiles(ctx, R, dS, (int)l, false);
    }
    launch(ctx, FHIP_K_POINTS, [&] {
        if (R.full) hipLaunchKernelGGL((k_pixels2d<32, true>), dim3(ctx->n_cu * 8), dim3(WAVE), 0, ctx->stream, dS);
        else hipLaunchKernelGGL((k_pixels2d<32, false>), dim3(ctx->n_cu * 16), dim3(WAVE), 0, ctx->stream, dS);
    });
    if (P.max_regs > 32)
        launch(ctx, FHIP_K_POINTS, [&] {
            const int g = blocks_big(ctx, R, R.lds_points_big, 16);
            if (R.full) hipLaunchKernelGGL((k_pixels2d<0, true>), dim3(g), dim3(WAVE), R.lds_points_big, ctx->stream, dS);
            else hipLaunchKernelGGL((k_pixels2d<0, false>), dim3(g), dim3(WAVE), R.lds_points_big, ctx->stream, dS);
        });
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_done, ctx->stream));     // (a later pipelined 3D frame that takes this buffer set waits for it)
    ctx->ev_done_valid = true;
    if (!out_is_device) {
        HIP_TRY(ctx, hipMemcpyAsync(out, d_out, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
        return finish_render(ctx);
    }
    return FHIP_OK;
}

static fhip_status render3d_part(fhip_ctx* ctx, const fhip_tape* tape, const fhip_render3d_config* cfg, void* out,
                                 int out_is_device, const PartSpec& part) {
    if (ctx->cancelled.load()) return fail(ctx, FHIP_ERR_CANCELLED, "cancelled");
    (void)hipSetDevice(ctx->device);
    RenderSetup R;
    memset(&R.S, 0, sizeof(R.S));
    FhRender& P = R.S.P;
    P.width = cfg->width; P.height = cfg->height; P.depth = cfg->depth; P.z = 0; P.pixel_perfect = 0;
    fhip_status st = bind_inputs(ctx, tape, cfg->axis_slots, cfg->var_keys, cfg->var_values, cfg->n_vars, P);
    if (st) return st;
    const uint32_t size[3] = {cfg->width, cfg->height, cfg->depth};
    float s2w[16];
    const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    fhip_screen_to_world(size, 3, s2w);
    mat_product(cfg->world_to_model ? cfg->world_to_model : ident, s2w, 4, P.mat);  // voxel.rs:107-109
    const std::vector<uint32_t> ts = cfg->tile_sizes ? trim_tiles(cfg->tile_sizes, cfg->n_tile_sizes, std::max(cfg->width, cfg->height))
                                                     : hip_tiles_3d(std::max(cfg->width, cfg->height), ctx->opt.vm_tiles != 0);
    // Frame pipelining (asynchronous renders): this frame takes the buffer set the previous frame did not use, and everything up
    // to and including its coarse levels is queued on a stream of its own - it depends on nothing the previous frame does, so it
    // runs beside that frame's slabs.  The slabs' tile chains follow on the side stream (after the previous frame's), the leaf
    // chains and the final image on the caller's stream as before.
    hipStream_t const main_stream = ctx->stream;
    // (a tape whose register files live in HBM takes the slow path: one region per workgroup, shared by the launches of a frame, so
    // nothing of the frame runs beside anything else)
    const bool huge = (size_t)std::max<uint32_t>(tape->t.n_regs, 1) * WAVE * 16 > FH_LDS_MAX || tiles_lds(std::max<uint32_t>(tape->t.n_regs, 1), tape->t.n_choices, 64) > FH_LDS_MAX;
    const bool fpipe = ctx->frame_pipeline && ctx->use_pipeline && !ctx->profiling && out_is_device && !ctx->opt.pipe_serial && !huge;
    struct StreamGuard { fhip_ctx* c; hipStream_t s; ~StreamGuard() { c->stream = s; } } stream_guard{ctx, main_stream};
    if (fpipe) {
        // (rotate: the current set goes to the back of the ring, the set used longest ago comes forward)
        for (uint32_t i = 0; i < ctx->extra_sets; i++) std::swap(static_cast<FrameBufs&>(*ctx), ctx->others[i]);
        ctx->stream = ctx->stream_pre;
        if (ctx->ev_done_valid) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream_pre, ctx->ev_done, 0));   // the set's previous frame has left it
    }
    st = prepare(ctx, tape, true, ts, part, R);
    if (st) return st;
    {   // input slots of the axes, and which inputs change along a pixel column (a z coefficient in the axis' matrix row, or a projective matrix)
        uint32_t u[16];
        memcpy(u, P.mat, sizeof(u));
        const bool proj = (((u[12] | u[13] | u[14]) & 0x7FFFFFFFu) | (u[15] ^ 0x3F800000u)) != 0;
        int slot[3] = {-1, -1, -1};
        for (int sl = 0; sl < FH_MAX_INPUTS; sl++) if (P.in_kind[sl] < 3) slot[P.in_kind[sl]] = sl;   // (the last slot of an axis)
        for (int ax = 0; ax < 3; ax++) {
            R.col_slots |= (uint32_t)(slot[ax] < 0 ? 0xFF : slot[ax]) << (8 * ax);
            const bool dep = proj || (u[4 * ax + 2] & 0x7FFFFFFFu) != 0;
            if (dep && slot[ax] >= 0) R.col_depmask |= 1u << slot[ax];
            if (dep) R.col_flags |= 0x20000u << ax;     // (bits 17 .. 19: this axis of the model changes along a pixel column - from the camera alone)
        }
        R.col_flags |= proj ? 0x10000u : 0u;
        // tiles of a tape that reads nothing varying along z repeat along z: worth looking for when x and y do not vary with it
        const bool xy_fixed = !proj && (slot[0] < 0 || !((R.col_depmask >> slot[0]) & 1)) && (slot[1] < 0 || !((R.col_depmask >> slot[1]) & 1));
        // (FHIP_NO_COLUMN_INV=1, diagnostics / bench: no column-invariance short cut anywhere - every input counts as varying
        // along z - which is what a model with z in every tape gets)
        const bool no_inv = ctx->opt.no_column_inv != 0;
        if (no_inv) R.col_depmask = 0xFFFFFFFFu;
        R.zrep = R.split && R.S.pre_levels > 0 && xy_fixed && !no_inv && !ctx->opt.no_zrep;
    }
    const size_t npix = (size_t)cfg->width * cfg->height;
    FhGeometryPixel* d_out = (FhGeometryPixel*)out;
    if (!out_is_device) { HIP_TRY(ctx, ctx->tmp_out.ensure(npix * sizeof(FhGeometryPixel))); d_out = (FhGeometryPixel*)ctx->tmp_out.p; }
    FhRenderState* dS = (FhRenderState*)ctx->state.p;
    // (FHIP_DEBUG_ZFILL, diagnostics: every pixel already at the far depth - the front slab's leaf kernel then finds all its
    // leaves but nothing pending, which times its per-workgroup and per-leaf set-up without the interpretation)
    const FrameClear clear3[3] = {{ctx->zbuf.p, npix * 8, ctx->opt.debug_zfill ? 0xFFFFFFFFu : 0u}, {ctx->normals.p, npix * 12, 0u},
                                  {ctx->mind.p, R.mind_words * 4, 0u}};
    st = upload_frame(ctx, tape, R, clear3);
    if (st) return st;
    const uint32_t n_groups = R.groups_per_slab;
    const uint32_t pre = R.S.pre_levels;
    const int reset_blocks = (int)std::max<uint32_t>(1, std::min<uint32_t>(1024, (std::max(R.table_words, n_groups) + 255) / 256));
    const int class_blocks = (int)((R.n_footprints + 255) / 256);
    // Pipelined frames: the root level stays on the pre-pass stream, the level below it moves to the head of this frame's tile
    // chains on the side stream.  The two coarse levels of a frame are one dependent chain of ~0.9 ms that, on one stream, set
    // the frame rate; split, the root level of frame n + 1 runs beside level 1 and the slabs of frame n, and the side stream
    // carries level 1 + the (now few) slab steps of its own frame.  (A frame alone sees no difference: the same chain.)
    const bool l1_side = fpipe && ctx->opt.l1_on_side && pre > 1 && ctx->stream2 && !ctx->opt.pipe_serial &&
                         ctx->use_pipeline && R.slab_hi - R.slab_lo > 1 && n_groups > 0;
    if (pre && n_groups) {  // coarse levels of every slab in one go
        for (uint32_t l = 0; l < pre; l++) {
            if (l == 1 && l1_side) {
                HIP_TRY(ctx, hipEventRecord(ctx->ev_l0, ctx->stream_pre));
                HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_l0, 0));
                ctx->stream = ctx->stream2;
            }
            if (R.zrep && l > 0) launch(ctx, FHIP_K_OTHER, [&] { hipLaunchKernelGGL(k_tape_flags, dim3(ctx->n_cu * 4), dim3(WAVE), 0, ctx->stream, dS, (int)l, R.col_depmask, 0); });
            launch_tiles(ctx, R, dS, (int)l, true);
        }
        if (R.zrep) launch(ctx, FHIP_K_OTHER, [&] { hipLaunchKernelGGL(k_tape_flags, dim3(ctx->n_cu * 8), dim3(WAVE), 0, ctx->stream, dS, (int)pre, R.col_depmask, 1); });
        launch(ctx, FHIP_K_OTHER, [&] { hipLaunchKernelGGL(k_mark_frame, dim3(1), dim3(1), 0, ctx->stream, dS); });
    }
    // Two-stream pipeline over the z-slabs: the tile stage of a slab runs on the side stream while
    // the leaves of the slab in front of it are evaluated on the caller's stream.  The occlusion
    // pyramid is then one slab stale, which is still exact (depths only grow).  Two slab contexts
    // (dS, dS + 1) alternate; each owns its leaves, leaf table, footprint lists and arena half.
    FhRenderState* const dS0 = dS;
    const bool pipe = ctx->use_pipeline && !ctx->profiling && R.slab_hi - R.slab_lo > 1 && n_groups > 0 && !R.big_hbm;
    hipStream_t const side_stream = ctx->opt.pipe_serial ? main_stream : ctx->stream2;  // diagnostics
    const uint32_t NC = pipe ? std::min<uint32_t>(ctx->slab_contexts, R.slab_hi - R.slab_lo) : 1;     // (no more contexts than slabs: each takes its share of the arena)
    ctx->forked = pipe ? NC : 0;
    if (pipe) {
        hipLaunchKernelGGL(k_fork_state, dim3(1), dim3(1), 0, ctx->stream, dS0, NC, (FhLeaf*)ctx->leaves_b.p,
                           (FhLeafRef*)ctx->leaf_table_b.p, (uint32_t*)ctx->fp_lists_b.p, (size_t)R.S.leaf_cap, (size_t)R.n_footprints);
        HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(side_stream, ctx->ev_fork, 0));
    }
    if (fpipe) {      // the rest of the frame is the caller's stream's (and the side stream's, which waits for the fork above)
        HIP_TRY(ctx, hipEventRecord(ctx->ev_pre, ctx->stream));     // (the stream the last coarse-level kernel went to)
        HIP_TRY(ctx, hipStreamWaitEvent(main_stream, ctx->ev_pre, 0));
        ctx->stream = main_stream;
    }
    int last_tail_idx = -1;
    // Where a slab's tile chain goes: the side stream, or (option tiles_stream = 1, pipelined frames of at most as many slabs
    // as there are slab contexts) the tail stream, every slab's chain queued there BEFORE the tail work of the first slab - the
    // side stream then carries level 1 of the coarse levels alone, the pre-pass stream the root level, and the three chains
    // of consecutive frames run beside each other.
    // (2, the default: there when the ROOT tape reads no input that changes along a pixel column - then no tape of the frame does,
    // the leaf stage is light and the tail stream has room; a frame whose leaf kernels fill the machine wants its tile chains on
    // the high-priority side stream: prospero.vm 1024^3 0.77 -> 0.64 ms per frame there, the same frames with the column-invariance
    // short cuts off 1.86 -> 2.01)
    bool root_invariant = !ctx->opt.no_column_inv && R.col_depmask != 0xFFFFFFFFu;
    for (uint64_t w : tape->t.ops)
        if (FH_W_OP((uint32_t)w) == FH_INPUT && ((R.col_depmask >> ((uint32_t)(w >> 32) & 31u)) & 1u)) { root_invariant = false; break; }
    const bool tiles_first = pipe && l1_side && (ctx->opt.tiles_stream == 1 || (ctx->opt.tiles_stream == 2 && root_invariant)) && ctx->stream3 &&
                             ctx->opt.tail_stream == 1 && R.asm_points && R.slab_hi - R.slab_lo <= NC;
    hipStream_t const tile_stream = tiles_first ? ctx->stream3 : side_stream;
    if (tiles_first) HIP_TRY(ctx, hipStreamWaitEvent(tile_stream, ctx->ev_fork, 0));
    auto tile_step = [&](int k, int idx) -> fhip_status {
        dS = dS0 + (pipe ? (uint32_t)idx % NC : 0u);
        if (pipe) {
            ctx->stream = tile_stream;
            if (idx >= (int)NC) HIP_TRY(ctx, hipStreamWaitEvent(tile_stream, ctx->ev_leaves[idx - (int)NC], 0));  // context free again
        }
        launch(ctx, FHIP_K_OTHER, [&] {
            // the usual pyramid (three levels, 4 x 4 each, 8 x 8 leaf tiles) has a kernel of its own
            // (up to 1024 x 1024: at 2048 x 2048 it was measured SLOWER than the generic kernel - 11.2 vs 8.2 ms per frame)
            const bool pyr3 = P.n_levels == 3 && P.tiles[2] == 8 && P.tiles[1] == 32 && P.tiles[0] == 128 &&
                              ((P.width + 31) / 32) * ((P.height + 31) / 32) <= 1024 && !ctx->opt.old_pyr;
            const bool rebuild = k != (int)R.slab_hi - 1;  // the first slab sees an empty image (pyramid pre-zeroed)
            if (rebuild && pyr3 && pre == 2 && !ctx->opt.no_slab_begin) {
                const uint32_t n1 = ((P.width + 31) / 32) * ((P.height + 31) / 32);
                hipLaunchKernelGGL(k_slab_begin3, dim3(n1 + reset_blocks), dim3(256), 0, ctx->stream, dS, n1, R.table_words, (uint32_t)k, n_groups);
                return;
            }
            hipLaunchKernelGGL(k_reset_slab, dim3(reset_blocks), dim3(256), 0, ctx->stream, dS, R.table_words, (uint32_t)k, n_groups,
                               (pyr3 && rebuild) ? 1u : 0u);
            if (rebuild && pyr3) {
                const uint32_t n1 = ((P.width + 31) / 32) * ((P.height + 31) / 32);
                hipLaunchKernelGGL(k_minpyramid3, dim3(n1), dim3(256), 0, ctx->stream, dS);
            } else if (rebuild)
                hipLaunchKernelGGL(k_minpyramid, dim3(P.roots_x * P.roots_y), dim3(256), 0, ctx->stream, dS);
        });
        for (uint32_t l = pre; l < P.n_levels; l++) launch_tiles(ctx, R, dS, (int)l, true);
        if (pipe) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_tiles[idx], tile_stream));
            ctx->stream = main_stream;
        }
        return FHIP_OK;
    };
    if (tiles_first)
        for (int k = (int)R.slab_hi - 1; k >= (int)R.slab_lo && n_groups; k--) {
            const fhip_status ts_ = tile_step(k, (int)R.slab_hi - 1 - k);
            if (ts_) { ctx->stream = main_stream; return ts_; }
        }
    for (int k = (int)R.slab_hi - 1; k >= (int)R.slab_lo && n_groups; k--) {  // front to back (voxel.rs:252-261)
        if (ctx->cancelled.load()) { ctx->stream = main_stream; return fail(ctx, FHIP_ERR_CANCELLED, "cancelled"); }
        const int idx = (int)R.slab_hi - 1 - k;
        if (!tiles_first) {
            const fhip_status ts_ = tile_step(k, idx);
            if (ts_) { ctx->stream = main_stream; return ts_; }
        }
        dS = dS0 + (pipe ? (uint32_t)idx % NC : 0u);
        // (diagnostics, FHIP_LEAF_STREAMS=2: leaf kernels of consecutive slabs on two streams, so that the tail of one overlaps
        // the head of the next - any interleaving gives the same image - at the price of lanes that no longer see the hits in front)
        const bool tail1 = ctx->opt.tail_stream == 1;
        hipStream_t const leaf_stream = (pipe && tail1 && R.asm_points && ctx->stream3 && ctx->stream_leaf2 && (idx & 1)) ? ctx->stream_leaf2 : main_stream;
        if (pipe) HIP_TRY(ctx, hipStreamWaitEvent(leaf_stream, ctx->ev_tiles[idx], 0));
        // The leaf kernel is the slab's critical chain.  What surrounds it - the footprint lists (needed by the normals and the
        // LDS-class leaves only), those leaves (any order with the others: atomic-max z-buffer) and the normals of the slab's
        // hits - are small launches that leave the machine mostly idle, so in the pipelined frame they run on a third stream
        // beside the leaf kernel of the NEXT slab: the normals kernel only takes hits of its own slab's depth range, and a hit
        // behind them can never replace them.  (Measured with three slab contexts, ms per frame: everything on the caller's stream 2.44, the normals only on the third stream 2.30, lists + normals 2.16 - once the min-depth pyramid kernel of the tile chain ran in blocks of four waves: its 16-wave blocks found no room beside a leaf kernel that is never interrupted, 166 us instead of 10.  FHIP_TAIL_STREAM=0 / 2 / 1.)
        const int tail_mode = ctx->opt.tail_stream;   // 0: off, 1: lists + normals, 2: normals only
        const bool tail = pipe && ctx->stream3 && tail_mode > 0 && R.asm_points;   // (the HIP leaf kernels walk the footprint lists)
        const uint32_t z_lo = (uint32_t)k * P.slab, z_hi = z_lo + P.slab;
        auto classify_work = [&] {
            launch(ctx, FHIP_K_OTHER, [&] { hipLaunchKernelGGL(k_classify3d, dim3(class_blocks), dim3(256), 0, ctx->stream, dS, R.asm_points ? 1 : 0); });
            if (P.max_regs > 32)
                launch(ctx, FHIP_K_POINTS, [&] {
                    const int g = blocks_big(ctx, R, R.lds_points_big, 16);
                    if (R.full) hipLaunchKernelGGL((k_leaves3d<2, 0, 1, true>), dim3(g), dim3(WAVE), R.lds_points_big, ctx->stream, dS);
                    else hipLaunchKernelGGL((k_leaves3d<2, 0, 1, false>), dim3(g), dim3(WAVE), R.lds_points_big, ctx->stream, dS);
                });
        };
        auto normals_work = [&] {
            launch(ctx, FHIP_K_NORMALS, [&] {
                const int gs = blocks_for(ctx, R.lds_normals_small, 8), gb = blocks_big(ctx, R, R.lds_normals_big, 8);
                if (R.asm_normals) {
                    // (list 0 of k_classify3d holds every footprint whose leaves need <= 32 registers: the assembly interpreter's file)
                    struct { FhRenderState* S; uint32_t n_waves, slots, z_lo, z_hi, pad[2]; } kn = {dS, (uint32_t)(ctx->n_cu * std::max(1, ctx->opt.normals_waves)), R.col_slots, z_lo, z_hi, {0, 0}};
                    (void)launch_asm(ctx, R.asm_points_t ? FH_ASM_NORMALS_T : FH_ASM_NORMALS, kn.n_waves, &kn, sizeof(kn));
                }
                else if (R.full) hipLaunchKernelGGL((k_normals3d<true, false>), dim3(gs), dim3(WAVE), R.lds_normals_small, ctx->stream, dS, z_lo, z_hi);
                else hipLaunchKernelGGL((k_normals3d<false, false>), dim3(gs), dim3(WAVE), R.lds_normals_small, ctx->stream, dS, z_lo, z_hi);
                if (P.max_regs > 32) {
                    if (R.full) hipLaunchKernelGGL((k_normals3d<true, true>), dim3(gb), dim3(WAVE), R.lds_normals_big, ctx->stream, dS, z_lo, z_hi);
                    else hipLaunchKernelGGL((k_normals3d<false, true>), dim3(gb), dim3(WAVE), R.lds_normals_big, ctx->stream, dS, z_lo, z_hi);
                }
            });
        };
        if (tail && tail_mode == 1) {
            ctx->stream = ctx->stream3;
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream3, ctx->ev_tiles[idx], 0));
            classify_work();
            ctx->stream = main_stream;
        } else classify_work();
        launch(ctx, FHIP_K_POINTS, [&] {
            // class 0: <= 16 registers, 4 voxels per lane; class 1: <= 32 registers, 2 per lane; class 2: LDS file
            if (R.asm_points) {
                // one launch for classes 0 and 1: 128 VGPRs -> 4 waves per SIMD
                // one workgroup per block of 4 footprints of one 8-voxel layer, front layers first
                // (FHIP_COL_WAVES=n: n persistent waves per CU instead, diagnostics)
                const uint32_t col_waves = (uint32_t)std::max(0, ctx->opt.col_waves);
                // per-frame constants of the leaf kernel (gen_interp.py gen_columns): input slots of the axes, the inputs that change
                // along a pixel column (a z coefficient in the axis' matrix row, or a projective matrix), projective flag
                struct { FhRenderState* S; uint32_t n_waves, slots, depmask, flags, pad[2]; } ka = {dS, (uint32_t)ctx->n_cu * col_waves, R.col_slots, R.col_depmask, R.col_flags, {0, 0}};
                const int which = R.asm_points_t ? FH_ASM_COLUMNS_T : FH_ASM_COLUMNS;
                if (col_waves) (void)launch_asm(ctx, which, ka.n_waves, &ka, sizeof(ka), 0, 1, leaf_stream);
                else {
                    const uint32_t blk = 1u << ctx->opt.col_blkl;   // footprints per workgroup: gen_interp.py BLKL
                    (void)launch_asm(ctx, which, (R.n_footprints + blk - 1) / blk, &ka, sizeof(ka), 0, P.slab / 8, leaf_stream);
                }
            } else if (R.full) {
                hipLaunchKernelGGL((k_leaves3d<0, 16, 4, true>), dim3(ctx->n_cu * 8), dim3(WAVE), 0, ctx->stream, dS);
                hipLaunchKernelGGL((k_leaves3d<1, 32, 2, true>), dim3(ctx->n_cu * 8), dim3(WAVE), 0, ctx->stream, dS);
            } else {
                hipLaunchKernelGGL((k_leaves3d<0, 16, 4, false>), dim3(ctx->n_cu * 16), dim3(WAVE), 0, ctx->stream, dS);
                hipLaunchKernelGGL((k_leaves3d<1, 32, 2, false>), dim3(ctx->n_cu * 16), dim3(WAVE), 0, ctx->stream, dS);
            }
        });
        if (tail) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_aux[idx], leaf_stream));          // the slab's leaf kernel is through
            ctx->stream = ctx->stream3;
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream3, ctx->ev_aux[idx], 0));
            normals_work();
            HIP_TRY(ctx, hipEventRecord(ctx->ev_leaves[idx], ctx->stream3));       // slab context free again; the last one: image complete
            ctx->stream = main_stream;
            last_tail_idx = idx;
            continue;
        }
        normals_work();
        if (pipe) HIP_TRY(ctx, hipEventRecord(ctx->ev_leaves[idx], main_stream));
    }
    if (last_tail_idx >= 0) HIP_TRY(ctx, hipStreamWaitEvent(main_stream, ctx->ev_leaves[last_tail_idx], 0));   // the third stream is serial: the last slab's normals
    launch(ctx, FHIP_K_OTHER, [&] { hipLaunchKernelGGL(k_finish3d, dim3(ctx->n_cu * 4), dim3(256), 0, ctx->stream, dS0, d_out, std::max<uint32_t>(ctx->forked, 1u), (uint32_t*)ctx->sticky.p); });
    HIP_TRY(ctx, hipGetLastError());
    if (ctx->launch_failed) { ctx->launch_failed = false; return FHIP_ERR_HIP; }   // (message in fhip_last_error)
    ctx->async_pending = out_is_device != 0;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_done, main_stream));     // (a later pipelined frame that takes this set waits for it)
    ctx->ev_done_valid = true;
    if (!out_is_device) {
        HIP_TRY(ctx, hipMemcpyAsync(out, d_out, npix * sizeof(FhGeometryPixel), hipMemcpyDeviceToHost, ctx->stream));
        return finish_render(ctx);
    }
    return FHIP_OK;
}
fhip_status fhip_render3d(fhip_ctx* ctx, const fhip_tape* tape, const fhip_render3d_config* cfg, void* out,
                          int out_is_device) {
    return render3d_part(ctx, tape, cfg, out, out_is_device, PartSpec{});
}
fhip_status fhip_render3d_shard(fhip_ctx* ctx, const fhip_tape* tape, const fhip_render3d_config* cfg, void* out,
                                int out_is_device, uint32_t shard, uint32_t n_shards) {
    if (n_shards == 0 || shard >= n_shards) return fail(ctx, FHIP_ERR_UNSUPPORTED, "bad shard");
    PartSpec p;
    p.shard = shard; p.n_shards = n_shards;
    return render3d_part(ctx, tape, cfg, out, out_is_device, p);
}
// Octant-style shards: block `index` = ix + nx * (iy + ny * iz) of an nx x ny x nz split of the volume (root-tile
// columns in x and y, z-slabs in z; iz = nz - 1 is the front).  Pixels outside the block's columns stay {0,0,0,0}.
fhip_status fhip_render3d_block(fhip_ctx* ctx, const fhip_tape* tape, const fhip_render3d_config* cfg, void* out,
                                int out_is_device, uint32_t index, const uint32_t split[3]) {
    if (!split || !split[0] || !split[1] || !split[2] || index >= split[0] * split[1] * split[2]) return fail(ctx, FHIP_ERR_UNSUPPORTED, "bad block");
    PartSpec p;
    p.nx = split[0]; p.ny = split[1]; p.nz = split[2];
    p.ix = index % p.nx; p.iy = (index / p.nx) % p.ny; p.iz = index / (p.nx * p.ny);
    return render3d_part(ctx, tape, cfg, out, out_is_device, p);
}
// Merge of two partial images of the same pixels from different z ranges (the stitch rule of voxel.rs:527-550 applied
// across shards): the larger depth wins, a tie goes to `front` (the range nearer the camera: a hit there carries the
// normal, the other side's equal depth is a filled tile's z + T + 1 with no normal); then the clamp depth >= D - 1 ->
// (D, [0, 0, 1]).  In place on `front`; device pointers; n pixels.
fhip_status fhip_merge_depth(fhip_ctx* ctx, void* front, const void* back, uint64_t n_pixels, uint32_t image_depth) {
    if (!n_pixels) return FHIP_OK;
    (void)hipSetDevice(ctx->device);
    hipLaunchKernelGGL(k_merge_depth, dim3((unsigned)std::min<uint64_t>((n_pixels + 255) / 256, 65535)), dim3(256), 0, ctx->stream,
                       (FhGeometryPixel*)front, (const FhGeometryPixel*)back, (size_t)n_pixels, image_depth);
    HIP_TRY(ctx, hipGetLastError());
    return FHIP_OK;
}

// ---- effects (fidget-raster/src/effects.rs) ---------------------------------------------------
// Inputs and outputs are device pointers when `on_device` != 0 (asynchronous on the context's stream: the
// usual case, the image was just rendered there); otherwise host buffers, staged through the context.
struct FxStage {
    fhip_ctx* ctx;
    int on_device;
    std::vector<std::pair<void*, std::pair<void*, size_t>>> outs;   // host ptr <- device ptr, bytes
    const void* in(DevBuf& b, const void* host, size_t bytes, hipError_t& e) {
        if (on_device || !host) return host;
        if ((e = b.ensure(bytes)) != hipSuccess) return nullptr;
        e = hipMemcpyAsync(b.p, host, bytes, hipMemcpyHostToDevice, ctx->stream);
        return b.p;
    }
    void* out(DevBuf& b, void* host, size_t bytes, hipError_t& e) {
        if (on_device) return host;
        if ((e = b.ensure(bytes)) != hipSuccess) return nullptr;
        outs.push_back({host, {b.p, bytes}});
        return b.p;
    }
    fhip_status finish() {
        HIP_TRY(ctx, hipGetLastError());
        for (auto& o : outs) HIP_TRY(ctx, hipMemcpyAsync(o.first, o.second.first, o.second.second, hipMemcpyDeviceToHost, ctx->stream));
        if (!on_device) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return FHIP_OK;
    }
};
static dim3 fx_grid(uint32_t w, uint32_t h) { return dim3((w + 15) / 16, (h + 15) / 16); }

fhip_status fhip_denoise_normals(fhip_ctx* ctx, const void* image, uint32_t width, uint32_t height, void* out, int on_device) {
    if (!width || !height) return FHIP_OK;
    (void)hipSetDevice(ctx->device);
    FxStage st{ctx, on_device, {}};
    hipError_t e = hipSuccess;
    const size_t bytes = (size_t)width * height * sizeof(FhGeometryPixel);
    const void* di = st.in(ctx->io_a, image, bytes, e); HIP_TRY(ctx, e);
    void* dout = st.out(ctx->io_b, out, bytes, e); HIP_TRY(ctx, e);
    hipLaunchKernelGGL(fhfx::k_fx_denoise, fx_grid(width, height), dim3(16, 16), 0, ctx->stream, (const FhGeometryPixel*)di, (int)width, (int)height,
                       (FhGeometryPixel*)dout);
    return st.finish();
}
fhip_status fhip_compute_ssao(fhip_ctx* ctx, const void* image, uint32_t width, uint32_t height, uint32_t depth, const float* kernel,
                              uint32_t n_kernel, const float* noise, uint32_t n_noise, float* out, int on_device) {
    if (!width || !height) return FHIP_OK;
    if (!n_kernel || !n_noise || !depth) return fail(ctx, FHIP_ERR_UNSUPPORTED, "empty SSAO kernel / noise or zero depth");
    (void)hipSetDevice(ctx->device);
    FxStage st{ctx, on_device, {}};
    hipError_t e = hipSuccess;
    const void* di = st.in(ctx->io_a, image, (size_t)width * height * sizeof(FhGeometryPixel), e); HIP_TRY(ctx, e);
    const void* dk = st.in(ctx->io_c, kernel, (size_t)n_kernel * 12, e); HIP_TRY(ctx, e);
    const void* dn = st.in(ctx->io_d, noise, (size_t)n_noise * 8, e); HIP_TRY(ctx, e);
    void* dout = st.out(ctx->io_b, out, (size_t)width * height * 4, e); HIP_TRY(ctx, e);
    hipLaunchKernelGGL(fhfx::k_fx_ssao, fx_grid(width, height), dim3(16, 16), 0, ctx->stream, (const FhGeometryPixel*)di, (int)width, (int)height,
                       (int)depth, (const float*)dk, (int)n_kernel, (const float*)dn, (int)n_noise, (float*)dout);
    return st.finish();
}
fhip_status fhip_blur_ssao(fhip_ctx* ctx, const float* ssao, uint32_t width, uint32_t height, float* out, int on_device) {
    if (!width || !height) return FHIP_OK;
    (void)hipSetDevice(ctx->device);
    FxStage st{ctx, on_device, {}};
    hipError_t e = hipSuccess;
    const void* di = st.in(ctx->io_a, ssao, (size_t)width * height * 4, e); HIP_TRY(ctx, e);
    void* dout = st.out(ctx->io_b, out, (size_t)width * height * 4, e); HIP_TRY(ctx, e);
    hipLaunchKernelGGL(fhfx::k_fx_blur, fx_grid(width, height), dim3(16, 16), 0, ctx->stream, (const float*)di, (int)width, (int)height, (float*)dout);
    return st.finish();
}
fhip_status fhip_apply_shading(fhip_ctx* ctx, const void* image, uint32_t width, uint32_t height, uint32_t depth, const float* ssao,
                               uint8_t* out_rgb, int on_device) {
    if (!width || !height) return FHIP_OK;
    if (!depth) return fail(ctx, FHIP_ERR_UNSUPPORTED, "zero depth");
    (void)hipSetDevice(ctx->device);
    FxStage st{ctx, on_device, {}};
    hipError_t e = hipSuccess;
    const void* di = st.in(ctx->io_a, image, (size_t)width * height * sizeof(FhGeometryPixel), e); HIP_TRY(ctx, e);
    const void* ds = st.in(ctx->io_c, ssao, (size_t)width * height * 4, e); HIP_TRY(ctx, e);
    void* dout = st.out(ctx->io_b, out_rgb, (size_t)width * height * 3, e); HIP_TRY(ctx, e);
    hipLaunchKernelGGL(fhfx::k_fx_shade, fx_grid(width, height), dim3(16, 16), 0, ctx->stream, (const FhGeometryPixel*)di, (int)width, (int)height,
                       (int)depth, (const float*)ds, (uint8_t*)dout);
    return st.finish();
}
fhip_status fhip_to_rgba(fhip_ctx* ctx, const float* image, uint32_t width, uint32_t height, int mode, uint8_t* out_rgba, int on_device) {
    if (mode < 0 || mode > 3) return fail(ctx, FHIP_ERR_UNSUPPORTED, "colour map 0..3");
    const size_t n = (size_t)width * height;
    if (!n) return FHIP_OK;
    (void)hipSetDevice(ctx->device);
    FxStage st{ctx, on_device, {}};
    hipError_t e = hipSuccess;
    const void* di = st.in(ctx->io_a, image, n * 4, e); HIP_TRY(ctx, e);
    void* dout = st.out(ctx->io_b, out_rgba, n * 4, e); HIP_TRY(ctx, e);
    hipLaunchKernelGGL(fhfx::k_fx_rgba, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const float*)di, n, mode, (uchar4*)dout);
    return st.finish();
}

// ---- meshing: the evaluation side of fidget_mesh::Octree::build (fidget-mesh/src/octree.rs) --------------------------------
// CELL_TO_VERT_TO_EDGES of fidget-mesh/build.rs:26-160: per corner mask, the inside -> outside edges grouped into cell vertices
// by connected region (filled regions first, then empty ones, each in ascending order of their corner sets)
static void build_mdc_table(FhMdcTable& T) {
    auto next = [](int a) { return (a << 1) > 4 ? 1 : (a << 1); };
    for (int i = 0; i < 256; i++) {
        int region_of[2][8];
        for (int pass = 0; pass < 2; pass++) {
            int* r = region_of[pass];
            for (int j = 0; j < 8; j++) r[j] = 1 << j;
            for (bool changed = true; changed;) {
                changed = false;
                for (int f = 0; f < 8; f++) {
                    if ((((i >> f) & 1) != 0) != (pass == 0)) continue;
                    for (int axis : {1, 2, 4}) {
                        const int g = f ^ axis;
                        if ((((i >> g) & 1) != 0) != (pass == 0)) continue;
                        const int v = r[f] | r[g];
                        if (r[f] != v || r[g] != v) { r[f] = v; r[g] = v; changed = true; }
                    }
                }
            }
        }
        std::vector<int> fr, er;
        for (int j = 0; j < 8; j++) ((i >> j) & 1 ? fr : er).push_back(region_of[(i >> j) & 1 ? 0 : 1][j]);
        for (auto* v : {&fr, &er}) { std::sort(v->begin(), v->end()); v->erase(std::unique(v->begin(), v->end()), v->end()); }
        int regions[8], ri = 0;
        for (auto* rs : {&fr, &er})
            for (int r : *rs) { for (int j = 0; j < 8; j++) if (r & (1 << j)) regions[j] = ri; ri++; }
        std::vector<std::pair<int, std::vector<std::pair<int, int>>>> verts;
        for (int rev = 0; rev < 2; rev++)
            for (int t : {1, 2, 4}) {
                const int u = next(t), v = next(u);
                for (int b = 0; b < 2; b++)
                    for (int a = 0; a < 2; a++) {
                        int start = (a * u) | (b * v), end = start | t;
                        if (rev) std::swap(start, end);
                        if (!(((i >> start) & 1) && !((i >> end) & 1))) continue;
                        auto it = std::find_if(verts.begin(), verts.end(), [&](auto& kv) { return kv.first == regions[start]; });
                        if (it == verts.end()) { verts.push_back({regions[start], {}}); it = verts.end() - 1; }
                        it->second.push_back({start, end});
                    }
            }
        std::sort(verts.begin(), verts.end(), [](auto& a, auto& b) { return a.first < b.first; });
        T.n_verts[i] = (uint8_t)verts.size();
        int ne = 0;
        for (int k = 0; k < 4; k++) T.per_vert[i][k] = 0;
        for (size_t vi = 0; vi < verts.size(); vi++) {
            T.per_vert[i][vi] = (uint8_t)verts[vi].second.size();
            for (auto& e : verts[vi].second) { T.edge[i][ne][0] = (uint8_t)e.first; T.edge[i][ne][1] = (uint8_t)e.second; ne++; }
        }
        T.n_edges[i] = (uint8_t)ne;
    }
}
struct fhip_mesh {
    // leaf records, in pinned host memory (the device writes them there in chunks while the leaf kernel is still running)
    struct PinnedLeaves {
        FhMeshLeaf* p = nullptr;
        size_t n = 0;
        bool borrowed = false;      // the context's cached area (fhip_mesh_build): not kept with the mesh
        // fhip_mesh_merge: the records stay where the parts' buffers hold them; segment k covers records seg_start[k] .. seg_start[k + 1] - 1
        std::vector<const FhMeshLeaf*> seg_p;
        std::vector<size_t> seg_start;
        const FhMeshLeaf& operator[](size_t i) const {
            if (seg_p.empty()) return p[i];
            size_t k = 0;
            while (k + 1 < seg_p.size() && i >= seg_start[k + 1]) k++;
            return seg_p[k][i - seg_start[k]];
        }
        const FhMeshLeaf* data() const { return p; }
        size_t size() const { return n; }
        ~PinnedLeaves() { if (p && !borrowed) (void)hipHostFree(p); }
    } leaves;
    uint64_t cells_evaluated = 0, full = 0, empty = 0, ambiguous_leaves = 0;
    std::vector<uint64_t> per_level;   // cells evaluated at each depth
    // per level, per evaluated cell: class (1 empty 2 full 3 ambiguous) and, for ambiguous cells, their index among the level's
    // ambiguous cells (= parent index of their children / leaf record index)
    std::vector<std::vector<uint8_t>> cls;
    std::vector<std::vector<uint32_t>> slot;
    fhmesh::VertVec vertices;                            // fhip_mesh_build: Mesh::vertices
    fhmesh::TriVec triangles;                            // ... Mesh::triangles
    uint64_t octree_cells = 0, octree_verts = 0;
    uint32_t depth = 0, part = 0, n_parts = 1;           // fhip_mesh_sample_part: which of the root's octants this one covers
};
// Assembly of the octree from the device's results, as Octree::recurse unwinds (octree.rs:556-583), then Octree::walk_dual
struct MeshAssembler {
    const fhip_mesh& M;
    uint32_t depth;
    fhmesh::Octree o;
    fhmesh::Cell build(uint32_t d, size_t i, const float* b, fhmesh::Hermite* hermite) {
        fhmesh::Cell res;
        const uint8_t c = M.cls[d][i];
        if (c == 2) { res.kind = fhmesh::C_FULL; return res; }
        if (c == 1) { res.kind = fhmesh::C_EMPTY; return res; }
        const uint32_t s = M.slot[d][i];
        if (d == depth) {       // leaf() (octree.rs:590-862) with the device's samples
            const FhMeshLeaf& lf = M.leaves[s];
            if (lf.mask == 0) { res.kind = fhmesh::C_EMPTY; return res; }
            if (lf.mask == 255) { res.kind = fhmesh::C_FULL; return res; }
            const fhmesh::Tables& T = fhmesh::tables();
            uint32_t ii = 0, vi = 0;
            for (auto& vs : T.v2e[lf.mask]) {
                bool forced = false;
                for (auto& e : vs) {
                    const uint32_t k = std::min<uint32_t>(ii, 11);
                    const float* g = lf.grad[k];
                    if (g[0] != g[0] || g[1] != g[1] || g[2] != g[2] || g[3] != g[3]) { forced = true; hermite->qef_err = fhmesh::QEF_ERR_INVALID; break; }
                    fhmesh::LeafIntersection& li = hermite->inter[fhmesh::to_undirected(e.first, e.second)];
                    li.pos[0] = lf.pos[k][0]; li.pos[1] = lf.pos[k][1]; li.pos[2] = lf.pos[k][2]; li.pos[3] = 1.0f;
                    for (int q = 0; q < 4; q++) li.grad[q] = g[q];
                    ii++;
                }
                if (!forced) hermite->qef_err = lf.qef_err[vi];
                vi++;
            }
            res.kind = fhmesh::C_LEAF; res.mask = (uint8_t)lf.mask; res.index = (uint32_t)o.verts.size();
            for (uint32_t v = 0; v < lf.n_verts; v++) o.verts.push_back(fhmesh::V3{lf.vert[v][0], lf.vert[v][1], lf.vert[v][2]});
            for (uint32_t e = 0; e < lf.n_edges; e++) o.verts.push_back(fhmesh::V3{lf.pos[e][0], lf.pos[e][1], lf.pos[e][2]});
            return res;
        }
        const size_t index = o.cells.size();
        o.cells.push_back(std::array<fhmesh::Cell, 8>());
        fhmesh::Hermite hc[8];
        for (int corner = 0; corner < 8; corner++) {
            float cb[6];
            for (int k = 0; k < 3; k++) {
                const float mid = (b[2 * k] + b[2 * k + 1]) / 2.0f;        // cell.rs:184-194
                if (corner & (1 << k)) { cb[2 * k] = mid; cb[2 * k + 1] = b[2 * k + 1]; } else { cb[2 * k] = b[2 * k]; cb[2 * k + 1] = mid; }
            }
            const fhmesh::Cell ch = build(d + 1, (size_t)s * 8 + corner, cb, &hc[corner]);
            o.cells[index][corner] = ch;
        }
        return o.check_done(b, index, hc, hermite);
    }
};
// The same assembly by independent subtrees on the host's threads, with the sequential recursion's result cell for cell and
// vertex for vertex (as Octree::build_inner_mt does with its thread pool, octree.rs:94-210, but spliced in recursion order):
// the ambiguous cells of level L are built each into an octree of its own, then the levels above them are assembled
// sequentially and take the subtrees in the order the recursion reaches them (cell / vertex indices shifted to where the
// recursion would have put them - check_done's bookkeeping only ever looks at the end of the arrays, which a subtree owns).
struct ParallelMeshAssembler {
    const fhip_mesh& M;
    uint32_t depth, L;
    fhmesh::Octree o;
    struct Task { size_t i; float b[6]; };
    struct Sub { fhmesh::Octree o; fhmesh::Cell root; fhmesh::Hermite h; size_t co = 0, vo = 0; };
    static fhmesh::Cell shift(fhmesh::Cell x, size_t co, size_t vo) {
        if (x.kind == fhmesh::C_BRANCH) x.index += (uint32_t)co;
        else if (x.kind == fhmesh::C_LEAF) x.index += (uint32_t)vo;
        return x;
    }
    std::vector<Task> tasks;
    std::vector<Sub> subs;
    size_t next = 0;
    static void child_bounds(const float* b, int corner, float* cb) {
        for (int k = 0; k < 3; k++) {
            const float mid = (b[2 * k] + b[2 * k + 1]) / 2.0f;        // cell.rs:184-194
            if (corner & (1 << k)) { cb[2 * k] = mid; cb[2 * k + 1] = b[2 * k + 1]; } else { cb[2 * k] = b[2 * k]; cb[2 * k + 1] = mid; }
        }
    }
    void plan(uint32_t d, size_t i, const float* b) {
        if (M.cls[d][i] != 3) return;
        if (d == L) { Task t; t.i = i; for (int k = 0; k < 6; k++) t.b[k] = b[k]; tasks.push_back(t); return; }
        const uint32_t s = M.slot[d][i];
        for (int corner = 0; corner < 8; corner++) { float cb[6]; child_bounds(b, corner, cb); plan(d + 1, (size_t)s * 8 + corner, cb); }
    }
    fhmesh::Cell top(uint32_t d, size_t i, const float* b, fhmesh::Hermite* hermite) {
        fhmesh::Cell res;
        const uint8_t c = M.cls[d][i];
        if (c == 2) { res.kind = fhmesh::C_FULL; return res; }
        if (c == 1) { res.kind = fhmesh::C_EMPTY; return res; }
        if (d == L) {       // splice the subtree
            Sub& S = subs[next++];
            // (room now, contents later and in parallel: nothing above this level ever reads inside a subtree)
            const size_t co = o.cells.size(), vo = o.verts.size();
            S.co = co; S.vo = vo;
            o.cells.resize(co + S.o.cells.size());
            o.verts.resize(vo + S.o.verts.size());
            *hermite = S.h;
            return shift(S.root, co, vo);
        }
        const uint32_t s = M.slot[d][i];
        const size_t index = o.cells.size();
        o.cells.push_back(std::array<fhmesh::Cell, 8>());
        fhmesh::Hermite hc[8];
        for (int corner = 0; corner < 8; corner++) {
            float cb[6];
            child_bounds(b, corner, cb);
            const fhmesh::Cell ch = top(d + 1, (size_t)s * 8 + corner, cb, &hc[corner]);
            o.cells[index][corner] = ch;
        }
        return o.check_done(b, index, hc, hermite);
    }
    fhmesh::Cell run(const float* rb, fhmesh::Hermite* h) {
        fhmesh::tables();
        const bool times = getenv("FHIP_MESH_TIMES") != nullptr;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t0 = now();
        plan(0, 0, rb);
        subs.resize(tasks.size());
        fhmesh::parallel_for(tasks.size(), [&](size_t k) {
            MeshAssembler A{M, depth, {}};
            subs[k].root = A.build(L, tasks[k].i, tasks[k].b, &subs[k].h);
            subs[k].o = std::move(A.o);
        });
        const double t1 = now();
        size_t total_c = 64, total_v = 64;
        for (auto& S : subs) { total_c += S.o.cells.size() + 1; total_v += S.o.verts.size(); }
        o.cells.reserve(total_c + 600 * tasks.size() / 512 + 4096);
        o.verts.reserve(total_v + 4096);
        const fhmesh::Cell root = top(0, 0, rb, h);
        const double t2 = now();
        fhmesh::parallel_for(subs.size(), [&](size_t k) {
            Sub& S = subs[k];
            // (a top-level collapse may have cut the arrays back below this subtree: then it is unreachable and not copied)
            if (S.co + S.o.cells.size() <= o.cells.size())
                for (size_t i = 0; i < S.o.cells.size(); i++) for (int q = 0; q < 8; q++) o.cells[S.co + i][q] = shift(S.o.cells[i][q], S.co, S.vo);
            if (S.vo + S.o.verts.size() <= o.verts.size() && !S.o.verts.empty())
                memcpy(&o.verts[S.vo], S.o.verts.data(), S.o.verts.size() * sizeof(fhmesh::V3));
            S.o = fhmesh::Octree();
        });
        if (times) fprintf(stderr, "fhip mesh assembly: %zu subtrees below level %u %.4f s, levels above + room %.4f s, splice %.4f s\n", tasks.size(), L, t1 - t0, t2 - t1, now() - t2);
        return root;
    }
};
// the root's octants part `part` of `n_parts` evaluates: octant o belongs to part o * n_parts / 8 (8 parts: one octant each, as
// Octree::build_inner_mt hands the root's children to its workers, octree.rs:109-123; 2 parts: the z halves)
static uint32_t mesh_part_mask(uint32_t part, uint32_t n_parts) {
    uint32_t m = 0;
    for (uint32_t o = 0; o < 8; o++) if (o * n_parts / 8 == part) m |= 1u << o;
    return m;
}
struct MeshTimes { bool on; double t_start, t_cells, t_leaf, t_copy; uint32_t n_leaf_cells; };
static void mesh_assemble(fhip_ctx* ctx, fhip_mesh* M, uint32_t depth, bool has_mat, const float* mat, MeshTimes& T);
// The octree assembled on the device (mesh_collapse.hpp oct_assemble; kernels in mesh.hip): arrays in HBM, one launch per pass and level
struct OctDevX {
    hipStream_t st;
    std::vector<void*> owned;
    hipError_t err = hipSuccess;
    void chk(hipError_t e) { if (e != hipSuccess && err == hipSuccess) err = e; }
    void* alloc(size_t b) {
        void* p = nullptr;
        const hipError_t e = hipMalloc(&p, b ? b : 4);
        if (e != hipSuccess) { chk(e); return nullptr; }
        owned.push_back(p);
        return p;
    }
    void zero(void* p, size_t b) { chk(hipMemsetAsync(p, 0, b, st)); }
    void read(void* d, const void* s, size_t b) { chk(hipMemcpyAsync(d, s, b, hipMemcpyDeviceToHost, st)); chk(hipStreamSynchronize(st)); }
    void kind(const fhmesh::OctLevel& D, const fhmesh::OctLevel& C, const fhmesh::OctLeaves& L, const FhMdcTable* T, uint32_t* counter, uint32_t n) {
        if (!n) return;
        hipLaunchKernelGGL(fhm::k_oct_kind, dim3((n + 255) / 256), dim3(256), 0, st, D, C, L, T, counter, n);
        chk(hipGetLastError());
    }
    void collapse(const fhmesh::OctLevel& D, const fhmesh::OctLevel& C, const fhmesh::OctLeaves& L, const FhMdcTable* T, uint32_t n) {
        if (!n) return;
        hipLaunchKernelGGL(fhm::k_oct_collapse, dim3((n + 63) / 64), dim3(64), 0, st, D, C, L, T, n);
        chk(hipGetLastError());
    }
    void place(const fhmesh::OctLevel& D, const fhmesh::OctLevel& C, const fhmesh::OctLeaves& L, const FhMdcTable* T, fhmesh::Cell* cells, fhmesh::V3* verts, const float* mat, uint32_t n) {
        if (!n) return;
        hipLaunchKernelGGL(fhm::k_oct_place, dim3((n + 255) / 256), dim3(256), 0, st, D, C, L, T, cells, verts, mat, n);
        chk(hipGetLastError());
    }
    void leaf_verts(const fhmesh::OctLeaves& L, fhmesh::V3* verts, const float* mat, uint32_t n) {
        if (!n) return;
        hipLaunchKernelGGL(fhm::k_oct_leaf_verts, dim3((n + 255) / 256), dim3(256), 0, st, L, verts, mat, n);
        chk(hipGetLastError());
    }
    void release() { for (void* p : owned) (void)hipFree(p); owned.clear(); }
};
static hipError_t mesh_assemble_device(fhip_ctx* ctx, fhip_mesh* M, uint32_t depth, std::vector<fhmesh::OctLevel>& lv, const FhMeshLeaf* rec, uint32_t n_rec, const FhMdcTable* table,
                                       bool has_mat, const float* mat, MeshTimes& T, std::string& why);
enum MeshMode { MESH_SAMPLE, MESH_BUILD, MESH_PART };
static fhip_status mesh_run(fhip_ctx* ctx, const fhip_tape* tape, uint32_t depth, const float* world_to_model, const int32_t* axis_slots,
                            const uint64_t* var_keys, const float* var_values, uint32_t n_vars, MeshMode mode, uint32_t part, uint32_t n_parts, fhip_mesh** out) {
    if (!out) return FHIP_ERR_BAD_TAPE;
    *out = nullptr;
    const bool assemble = mode == MESH_BUILD;
    // fhip_mesh_build assembles the octree on the device: the levels' arrays and the leaf records stay in HBM, the host gets the finished
    // octree for the dual walk.  (Option mesh_device_assembly 0: on the host's threads from copies of both, as fhip_mesh_merge does.)
    const bool dev_asm = assemble && n_parts == 1 && ctx->opt.mesh_device_assembly;
    const bool keep = mode != MESH_SAMPLE && !dev_asm;
    if (depth > 20) return fail(ctx, FHIP_ERR_UNSUPPORTED, "octree depth above 20");
    if (n_parts < 1 || n_parts > 8 || part >= n_parts) return fail(ctx, FHIP_ERR_UNSUPPORTED, "mesh parts: 1..8, part < n_parts");
    const fh::HostTape& t = tape->t;
    if (t.n_outputs != 1) return fail(ctx, FHIP_ERR_BAD_TAPE, "shape tapes have exactly one output");
    (void)hipSetDevice(ctx->device);
    { fhip_status ts_ = tape_to_device(ctx, tape); if (ts_) return ts_; }
    FhRender R;
    memset(&R, 0, sizeof(R));
    fhip_status st = bind_inputs(ctx, tape, axis_slots, var_keys, var_values, n_vars, R);
    if (st) return st;
    FhMeshParams P;
    memset(&P, 0, sizeof(P));
    P.tape = tape->d_ops; P.len = (uint32_t)t.ops.size(); P.n_regs = std::max<uint32_t>(t.n_regs, 1);
    bool ident = true;
    if (world_to_model) for (int i = 0; i < 16; i++) { P.mat[i] = world_to_model[i]; ident &= world_to_model[i] == ((i % 5 == 0) ? 1.0f : 0.0f); }
    P.has_mat = (world_to_model && !ident) ? 1 : 0;     // octree.rs:487-492: no transform at all for the identity
    for (int s = 0; s < FH_MAX_INPUTS; s++) { P.in_kind[s] = R.in_kind[s]; P.in_value[s] = R.in_value[s]; }
    const size_t lds_iv = (size_t)P.n_regs * WAVE * 8, lds_leaf = (size_t)P.n_regs * WAVE * 16;
    if (lds_leaf + 1024 > FH_LDS_MAX) return fail(ctx, FHIP_ERR_UNSUPPORTED, "register file exceeds LDS");
    {   // (function attributes are per device; contexts on several host threads may arrive here together)
        static std::mutex attr_lock;
        static bool attr_done[64] = {};
        std::lock_guard<std::mutex> guard(attr_lock);
        const int d = ctx->device & 63;
        if (!attr_done[d]) {
            (void)hipFuncSetAttribute((const void*)fhm::k_mesh_cells, hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS_MAX);
            (void)hipFuncSetAttribute((const void*)fhm::k_mesh_leaf, hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS_MAX - 2048);
            (void)hipFuncSetAttribute((const void*)fhm::k_mesh_corners, hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS_MAX);
            (void)hipFuncSetAttribute((const void*)fhm::k_mesh_edges, hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS_MAX);
            (void)hipFuncSetAttribute((const void*)fhm::k_mesh_grads, hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS_MAX);
            attr_done[d] = true;
        }
    }
    fhip_mesh* M = new fhip_mesh();
    M->depth = depth; M->part = part; M->n_parts = n_parts;
    const bool times = getenv("FHIP_MESH_TIMES") != nullptr;       // diagnostic: phase wall times on stderr
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = now();
    double t_cells = 0, t_leaf = 0, t_copy = 0;
    DevBuf bufs[2], counters, table, leaves, d_cls, d_slot, edge_list, edge_count, edge_br, edge_vars, edge_vals;
    std::vector<DevBuf> lv_cls, lv_slot, lv_amb;        // dev_asm: every level's classes, slots and ambiguous cells stay
    if (dev_asm) { lv_cls.resize(depth + 1); lv_slot.resize(depth + 1); lv_amb.resize(depth + 1); }
    std::vector<uint32_t> lv_n_amb;
    auto cleanup = [&] {
        bufs[0].release(); bufs[1].release(); counters.release(); table.release(); leaves.release(); d_cls.release(); d_slot.release();
        edge_list.release(); edge_count.release(); edge_br.release(); edge_vars.release(); edge_vals.release();
        for (auto* v : {&lv_cls, &lv_slot, &lv_amb}) for (DevBuf& b : *v) b.release();
    };
#define MESH_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { cleanup(); delete M; return fail(ctx, FHIP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } } while (0)
    MESH_TRY(counters.ensure(16));
    FhMeshCell root;
    for (int k = 0; k < 3; k++) { root.b[2 * k] = -1.0f; root.b[2 * k + 1] = 1.0f; }     // CellBounds::new (cell.rs:171-176)
    root.path = 1;
    MESH_TRY(bufs[0].ensure(sizeof(FhMeshCell)));
    MESH_TRY(hipMemcpyAsync(bufs[0].p, &root, sizeof(root), hipMemcpyHostToDevice, ctx->stream));
    uint32_t n_in = 1;      // cells in bufs[cur] to evaluate (level 0) or whose 8 children to evaluate
    int cur = 0;
    uint32_t n_leaf_cells = 0;
    for (uint32_t d = 0; d <= depth; d++) {
        const uint64_t n64 = d == 0 ? 1 : (uint64_t)n_in * 8;
        if (n64 > (1ull << 30)) { cleanup(); delete M; return fail(ctx, FHIP_ERR_OVERFLOW, "octree level above 2^30 cells"); }
        const uint32_t n = (uint32_t)n64;
        DevBuf& out_cells = dev_asm ? lv_amb[d] : bufs[cur ^ 1];
        const void* in_cells = (dev_asm && d > 0) ? lv_amb[d - 1].p : bufs[cur].p;
        MESH_TRY(out_cells.ensure((size_t)n * sizeof(FhMeshCell)));
        MESH_TRY(hipMemsetAsync(counters.p, 0, 16, ctx->stream));
        if (keep) { MESH_TRY(d_cls.ensure(n)); MESH_TRY(d_slot.ensure((size_t)n * 4)); }
        if (dev_asm) { MESH_TRY(lv_cls[d].ensure(n)); MESH_TRY(lv_slot[d].ensure((size_t)n * 4)); }
        uint8_t* const cls_p = dev_asm ? (uint8_t*)lv_cls[d].p : (keep ? (uint8_t*)d_cls.p : nullptr);
        uint32_t* const slot_p = dev_asm ? (uint32_t*)lv_slot[d].p : (keep ? (uint32_t*)d_slot.p : nullptr);
        const uint32_t child_mask = (d == 1 && n_parts > 1) ? mesh_part_mask(part, n_parts) : 0xFFu;      // (level 1 = the root's 8 children)
        hipLaunchKernelGGL(fhm::k_mesh_cells, dim3((n + WAVE - 1) / WAVE), dim3(WAVE), lds_iv, ctx->stream, P, (const FhMeshCell*)in_cells, n, d == 0 ? 0 : 1,
                           (FhMeshCell*)out_cells.p, (uint32_t*)counters.p, n, cls_p, slot_p, child_mask);
        MESH_TRY(hipGetLastError());
        uint32_t c[4];
        MESH_TRY(hipMemcpyAsync(c, counters.p, 16, hipMemcpyDeviceToHost, ctx->stream));
        if (keep) {
            M->cls.emplace_back(n); M->slot.emplace_back(n);
            MESH_TRY(hipMemcpyAsync(M->cls.back().data(), d_cls.p, n, hipMemcpyDeviceToHost, ctx->stream));
            MESH_TRY(hipMemcpyAsync(M->slot.back().data(), d_slot.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        }
        MESH_TRY(hipStreamSynchronize(ctx->stream));
        const uint32_t n_here = child_mask == 0xFFu ? n : (uint32_t)__builtin_popcount(child_mask);
        M->cells_evaluated += n_here; M->full += c[1]; M->empty += c[2];
        M->per_level.push_back(n_here);
        cur ^= 1;
        n_in = c[0];
        lv_n_amb.push_back(c[0]);
        if (d == depth) n_leaf_cells = c[0];
        if (n_in == 0) break;
    }
    M->ambiguous_leaves = n_leaf_cells;
    t_cells = now() - t_start;
    FhMdcTable mdc;
    if (n_leaf_cells || dev_asm) {
        build_mdc_table(mdc);
        MESH_TRY(table.ensure(sizeof(mdc)));
        MESH_TRY(hipMemcpyAsync(table.p, &mdc, sizeof(mdc), hipMemcpyHostToDevice, ctx->stream));
    }
    // one chunk of leaf cells sampled into records: as passes in which every lane has a point of its own (corners, the edge search over the
    // chunk's list of edges, gradients), or - FHIP_MESH_LEAF_PASSES=0 - one wavefront per cell (k_mesh_leaf); then the cell vertices' QEFs
    const uint32_t LEAF_CH = 1u << 19;
    const char* const lp_env = getenv("FHIP_MESH_LEAF_PASSES");        // diagnostic: 0 = k_mesh_leaf, the kernel the passes are checked against
    const bool leaf_passes = !(lp_env && lp_env[0] == '0');
    const char* const be_env = getenv("FHIP_MESH_BULK_EDGES");          // diagnostic: 0 = the edge search by k_mesh_edges (the generic interpreter)
    const bool bulk_edges = leaf_passes && ctx->use_asm && P.n_regs <= 32 && !(be_env && be_env[0] == '0');
    uint32_t n_slots = std::max<uint32_t>(t.n_vars, 1);
    for (uint32_t sl = 0; sl < FH_MAX_INPUTS; sl++) if (P.in_kind[sl] < 3) n_slots = std::max(n_slots, sl + 1);
    const size_t lds_f32 = (size_t)P.n_regs * WAVE * 4;
    auto sample_chunk = [&](const FhMeshCell* cells, FhMeshLeaf* recs, uint32_t cnt) -> hipError_t {
        hipError_t e = hipSuccess;
        auto ck = [&](hipError_t x) { if (x != hipSuccess && e == hipSuccess) e = x; };
        if (leaf_passes && cnt < (1u << 28)) {
            ck(edge_list.ensure((size_t)LEAF_CH * 12 * 4));
            ck(edge_count.ensure(4));
            if (e != hipSuccess) return e;
            ck(hipMemsetAsync(edge_count.p, 0, 4, ctx->stream));
            hipLaunchKernelGGL(fhm::k_mesh_corners, dim3((cnt + 7) / 8), dim3(WAVE), lds_f32, ctx->stream, P, cells, cnt, (const FhMdcTable*)table.p, recs,
                               (uint32_t*)edge_count.p, (uint32_t*)edge_list.p);
            ck(hipGetLastError());
            uint32_t n_edges = 0;
            ck(hipMemcpyAsync(&n_edges, edge_count.p, 4, hipMemcpyDeviceToHost, ctx->stream));
            ck(hipStreamSynchronize(ctx->stream));
            if (e == hipSuccess && n_edges && bulk_edges) {
                // the four rounds as passes over the chunk's edges, the samples' values from the assembly bulk interpreter (mesh_edges.hpp)
                const uint32_t n = n_edges * 16u;
                ck(edge_br.ensure((size_t)n_edges * sizeof(fhmesh::EdgeBracket)));
                ck(edge_vars.ensure((size_t)n_slots * n * 4));
                ck(edge_vals.ensure((size_t)n * 4));
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(fhm::k_mesh_edge_begin, dim3((n_edges + 255) / 256), dim3(256), 0, ctx->stream, (const FhMdcTable*)table.p, (const FhMeshLeaf*)recs,
                                   (const uint32_t*)edge_list.p, n_edges, (fhmesh::EdgeBracket*)edge_br.p);
                ck(hipGetLastError());
                for (uint32_t sl = 0; sl < n_slots; sl++)
                    if (P.in_kind[sl] >= 3) {
                        hipLaunchKernelGGL(fhm::k_mesh_fill, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (float*)edge_vars.p + (size_t)sl * n, P.in_value[sl], n);
                        ck(hipGetLastError());
                    }
                struct { const uint64_t* tape; const float* vars; float* out; uint32_t len, n; } ka = {tape->d_ops, (const float*)edge_vars.p, (float*)edge_vals.p, P.len, n};
                const bool plain = tape_asm_ok(t);
                const uint32_t per = P.n_regs <= 16 ? 256 : 128;
                const int which = P.n_regs <= 16 ? (plain ? FH_ASM_FLOAT_16x4 : FH_ASM_FLOAT_16x4_T) : (plain ? FH_ASM_FLOAT_32x2 : FH_ASM_FLOAT_32x2_T);
                for (int round = 0; round < 4 && e == hipSuccess; round++) {
                    hipLaunchKernelGGL(fhm::k_mesh_edge_points, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, P, (const FhMeshLeaf*)recs, (const uint32_t*)edge_list.p,
                                       (const fhmesh::EdgeBracket*)edge_br.p, n_edges, (float*)edge_vars.p, n);
                    ck(hipGetLastError());
                    ck(launch_asm(ctx, which, (n + per - 1) / per, &ka, sizeof(ka)));
                    hipLaunchKernelGGL(fhm::k_mesh_edge_narrow, dim3((n_edges + 255) / 256), dim3(256), 0, ctx->stream, (fhmesh::EdgeBracket*)edge_br.p, (const float*)edge_vals.p, n_edges);
                    ck(hipGetLastError());
                }
                hipLaunchKernelGGL(fhm::k_mesh_edge_end, dim3((n_edges + 255) / 256), dim3(256), 0, ctx->stream, recs, (const uint32_t*)edge_list.p, (const fhmesh::EdgeBracket*)edge_br.p, n_edges);
                ck(hipGetLastError());
                hipLaunchKernelGGL(fhm::k_mesh_grads, dim3((n_edges + WAVE - 1) / WAVE), dim3(WAVE), lds_leaf, ctx->stream, P, recs, (const uint32_t*)edge_list.p, n_edges);
                ck(hipGetLastError());
            } else if (e == hipSuccess && n_edges) {
                hipLaunchKernelGGL(fhm::k_mesh_edges, dim3((n_edges + 3) / 4), dim3(WAVE), lds_f32, ctx->stream, P, (const FhMdcTable*)table.p, recs, (const uint32_t*)edge_list.p, n_edges);
                ck(hipGetLastError());
                hipLaunchKernelGGL(fhm::k_mesh_grads, dim3((n_edges + WAVE - 1) / WAVE), dim3(WAVE), lds_leaf, ctx->stream, P, recs, (const uint32_t*)edge_list.p, n_edges);
                ck(hipGetLastError());
            }
        } else {
            hipLaunchKernelGGL(fhm::k_mesh_leaf, dim3(cnt), dim3(WAVE), lds_leaf, ctx->stream, P, cells, cnt, (const FhMdcTable*)table.p, recs);
            ck(hipGetLastError());
        }
        hipLaunchKernelGGL(fhm::k_mesh_leaf_qef, dim3((cnt + WAVE - 1) / WAVE), dim3(WAVE), 0, ctx->stream, (const FhMdcTable*)table.p, recs, cnt);
        ck(hipGetLastError());
        return e;
    };
    if (n_leaf_cells && dev_asm) {      // the records stay in HBM
        MESH_TRY(leaves.ensure((size_t)n_leaf_cells * sizeof(FhMeshLeaf)));
        const uint32_t CH = LEAF_CH;
        const void* const leaf_cells = lv_amb[depth].p;
        for (uint32_t off = 0; off < n_leaf_cells; off += CH) {
            const uint32_t cnt = std::min<uint32_t>(CH, n_leaf_cells - off);
            const hipError_t se = sample_chunk((const FhMeshCell*)leaf_cells + off, (FhMeshLeaf*)leaves.p + off, cnt);
            MESH_TRY(se);
        }
        if (times) { MESH_TRY(hipStreamSynchronize(ctx->stream)); t_leaf = now() - t_start - t_cells; }
    } else if (n_leaf_cells) {
        MESH_TRY(leaves.ensure((size_t)n_leaf_cells * sizeof(FhMeshLeaf)));
        // in chunks: the records of chunk k travel to the host (second stream) while chunk k + 1 is sampled
        const size_t leaf_bytes = (size_t)n_leaf_cells * sizeof(FhMeshLeaf);
        if (assemble) {     // the records are only needed until the octree is assembled: the context's cached landing area
            if (ctx->mesh_pinned_cap < leaf_bytes) {
                if (ctx->mesh_pinned) (void)hipHostFree(ctx->mesh_pinned);
                ctx->mesh_pinned = nullptr; ctx->mesh_pinned_cap = 0;
                MESH_TRY(hipHostMalloc(&ctx->mesh_pinned, leaf_bytes + leaf_bytes / 8, hipHostMallocDefault));
                ctx->mesh_pinned_cap = leaf_bytes + leaf_bytes / 8;
            }
            M->leaves.p = (FhMeshLeaf*)ctx->mesh_pinned;
            M->leaves.borrowed = true;
        } else
            MESH_TRY(hipHostMalloc((void**)&M->leaves.p, leaf_bytes, hipHostMallocDefault));
        M->leaves.n = n_leaf_cells;
        const uint32_t CH = LEAF_CH;
        std::vector<hipEvent_t> evs;
        hipStream_t const copy_stream = ctx->stream2 ? ctx->stream2 : ctx->stream;
        bool ok = true;
        hipError_t first_err = hipSuccess;
        auto chk = [&](hipError_t e) { if (e != hipSuccess && ok) { ok = false; first_err = e; } };
        for (uint32_t off = 0; off < n_leaf_cells && ok; off += CH) {
            const uint32_t cnt = std::min<uint32_t>(CH, n_leaf_cells - off);
            chk(sample_chunk((const FhMeshCell*)bufs[cur].p + off, (FhMeshLeaf*)leaves.p + off, cnt));
            hipEvent_t ev = nullptr;
            chk(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            if (ev) evs.push_back(ev);
            chk(hipEventRecord(ev, ctx->stream));
            chk(hipStreamWaitEvent(copy_stream, ev, 0));
            chk(hipMemcpyAsync(M->leaves.p + off, (FhMeshLeaf*)leaves.p + off, (size_t)cnt * sizeof(FhMeshLeaf), hipMemcpyDeviceToHost, copy_stream));
        }
        if (times) { chk(hipStreamSynchronize(ctx->stream)); t_leaf = now() - t_start - t_cells; }
        chk(hipStreamSynchronize(ctx->stream));
        chk(hipStreamSynchronize(copy_stream));
        for (hipEvent_t e : evs) (void)hipEventDestroy(e);
        MESH_TRY(first_err);
    }
    if (dev_asm) {
        std::vector<fhmesh::OctLevel> lv(lv_n_amb.size());
        for (size_t d = 0; d < lv.size(); d++) {
            lv[d].cls = (const uint8_t*)lv_cls[d].p; lv[d].slot = (const uint32_t*)lv_slot[d].p;
            lv[d].amb = (const FhMeshCell*)lv_amb[d].p; lv[d].n_amb = lv_n_amb[d];
        }
        MeshTimes MT{times, t_start, t_cells, t_leaf, 0.0, n_leaf_cells};
        std::string why;
        const hipError_t ae = mesh_assemble_device(ctx, M, depth, lv, (const FhMeshLeaf*)leaves.p, n_leaf_cells, (const FhMdcTable*)table.p, P.has_mat != 0, P.mat, MT, why);
        if (ae != hipSuccess && !why.empty()) { cleanup(); delete M; return fail(ctx, FHIP_ERR_OVERFLOW, why); }
        MESH_TRY(ae);
        cleanup();
        *out = M;
        return FHIP_OK;
    }
#undef MESH_TRY
    cleanup();
    t_copy = now() - t_start - t_cells - t_leaf;
    MeshTimes MT{times, t_start, t_cells, t_leaf, t_copy, n_leaf_cells};
    if (assemble) mesh_assemble(ctx, M, depth, P.has_mat != 0, P.mat, MT);
    else if (times)
        fprintf(stderr, "fhip mesh depth %u (part %u of %u): cells %.4f s (%llu evaluated), leaf kernel %.4f s (%u leaves), copies %.4f s\n", depth, part, n_parts,
                t_cells, (unsigned long long)M->cells_evaluated, t_leaf, n_leaf_cells, t_copy);
    *out = M;
    return FHIP_OK;
}
// Octree assembly (cell collapse included) and dual walk on the host's threads, from the classes / slots / leaf records in M
static void mesh_cache_release(void* octree) { delete (fhmesh::Octree*)octree; }
static void mesh_assemble(fhip_ctx* ctx, fhip_mesh* M, uint32_t depth, bool has_mat, const float* mat, MeshTimes& T) {
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    double t_asm = 0, t_walk = 0;
    {
        const float rb[6] = {-1.0f, 1.0f, -1.0f, 1.0f, -1.0f, 1.0f};
        fhmesh::Hermite h;
        // (a level the recursion never reached - everything above it was decided - has no arrays: only levels 0 .. cls.size()-1 are indexed)
        const uint32_t split = std::min<uint32_t>(depth, getenv("FHIP_MESH_SPLIT") ? (uint32_t)atoi(getenv("FHIP_MESH_SPLIT")) : 5u);
        const bool par = split >= 1 && M->cls.size() > split && fhmesh::mesh_threads() > 1;
        struct { fhmesh::Octree o; } A;
        if (par) {
            ParallelMeshAssembler PA{*M, depth, split, {}, {}, {}, 0};
            if (ctx && ctx->mesh_octree_cache) {       // the arrays of the last build: their room, not their contents
                PA.o = std::move(*(fhmesh::Octree*)ctx->mesh_octree_cache);
                PA.o.cells.clear(); PA.o.verts.clear(); PA.o.root = fhmesh::Cell();
            }
            PA.o.root = PA.run(rb, &h);
            A.o = std::move(PA.o);
        } else {
            MeshAssembler SA{*M, depth, {}};
            SA.o.root = SA.build(0, 0, rb, &h);
            A.o = std::move(SA.o);
        }
        if (has_mat)       // octree.rs:58-65: vertices back to model space (nalgebra transform_point)
            for (auto& v : A.o.verts) {
                const float x = v.x, y = v.y, z = v.z;
                const float n = ((mat[12] * x + mat[13] * y) + mat[14] * z) + mat[15];
                float a = ((mat[0] * x + mat[1] * y) + mat[2] * z) + mat[3];
                float b = ((mat[4] * x + mat[5] * y) + mat[6] * z) + mat[7];
                float c = ((mat[8] * x + mat[9] * y) + mat[10] * z) + mat[11];
                if (n != 0.0f) { a = a / n; b = b / n; c = c / n; }
                v.x = a; v.y = b; v.z = c;
            }
        t_asm = now() - t0;
        M->leaves.p = nullptr; M->leaves.n = 0;      // (borrowed from the context or from the parts' buffers: gone with the assembly)
        M->leaves.seg_p.clear(); M->leaves.seg_start.clear();
        fhmesh::ParallelWalker W(A.o);
        if (ctx) { W.scratch = &ctx->mesh_first; W.scratch_cap = &ctx->mesh_first_cap; }
        W.run();
        t_walk = now() - t0 - t_asm;
        M->octree_cells = A.o.cells.size(); M->octree_verts = A.o.verts.size();
        M->vertices.swap(W.vertices);
        M->triangles.swap(W.triangles);
        if (ctx && par) {
            if (!ctx->mesh_octree_cache) ctx->mesh_octree_cache = new fhmesh::Octree();
            *(fhmesh::Octree*)ctx->mesh_octree_cache = std::move(A.o);
        }
    }
    if (T.on)
        fprintf(stderr, "fhip mesh depth %u: cells %.4f s (%llu evaluated), leaf kernel %.4f s (%u leaves), copies %.4f s, assembly %.4f s, dual walk %.4f s, total %.4f s\n",
                depth, T.t_cells, (unsigned long long)M->cells_evaluated, T.t_leaf, T.n_leaf_cells, T.t_copy, t_asm, t_walk, now() - T.t_start);
}
// fhip_mesh_build's second half: the octree assembled in HBM (check_done / collapse / places, mesh_collapse.hpp), its blocks of cells copied
// to the context's pinned landing area, Octree::walk_dual on the host's threads over them, and the mesh's vertices - the walk knows which
// of the octree's they are - gathered on the device.  Neither the leaf records (528 bytes each) nor the octree's vertices (at depth 10:
// 191 M, of which the mesh uses 7.5 M) leave the device.
static hipError_t mesh_assemble_device(fhip_ctx* ctx, fhip_mesh* M, uint32_t depth, std::vector<fhmesh::OctLevel>& lv, const FhMeshLeaf* rec, uint32_t n_rec, const FhMdcTable* table,
                                       bool has_mat, const float* mat, MeshTimes& T, std::string& why) {
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    OctDevX x{ctx->stream, {}, hipSuccess};
    auto give_up = [&](hipError_t e) { (void)hipStreamSynchronize(ctx->stream); x.release(); return e; };
    float* d_mat = nullptr;
    if (has_mat) {
        d_mat = (float*)x.alloc(64);
        if (d_mat) x.chk(hipMemcpyAsync(d_mat, mat, 64, hipMemcpyHostToDevice, ctx->stream));
    }
    fhmesh::OctOut oo;
    const int rc = x.err != hipSuccess ? (int)fhmesh::OCT_NO_MEMORY : fhmesh::oct_assemble(x, depth, lv.data(), (uint32_t)lv.size(), rec, n_rec, table, d_mat, &oo);
    if (rc == fhmesh::OCT_TOO_MANY_VERTICES) { why = "the octree has more than 2^32 vertices"; return give_up(hipErrorInvalidValue); }
    if (rc != fhmesh::OCT_OK || x.err != hipSuccess) return give_up(x.err != hipSuccess ? x.err : hipErrorOutOfMemory);
    const size_t cell_bytes = (size_t)oo.n_blocks * 8 * sizeof(fhmesh::Cell);
    if (ctx->mesh_pinned_cap < cell_bytes + 256) {
        if (ctx->mesh_pinned) (void)hipHostFree(ctx->mesh_pinned);
        ctx->mesh_pinned = nullptr; ctx->mesh_pinned_cap = 0;
        const size_t room = cell_bytes + cell_bytes / 8 + 256;
        const hipError_t e = hipHostMalloc(&ctx->mesh_pinned, room, hipHostMallocDefault);
        if (e != hipSuccess) return give_up(e);
        ctx->mesh_pinned_cap = room;
    }
    fhmesh::Octree o;
    o.root = oo.root;
    o.cells_view = (const std::array<fhmesh::Cell, 8>*)ctx->mesh_pinned; o.n_cells_view = oo.n_blocks;
    o.verts_view = nullptr; o.n_verts_view = oo.n_verts;      // (never read: the walk gathers through the device)
    if (cell_bytes) x.chk(hipMemcpyAsync(ctx->mesh_pinned, oo.cells, cell_bytes, hipMemcpyDeviceToHost, ctx->stream));
    x.chk(hipStreamSynchronize(ctx->stream));
    if (x.err != hipSuccess) return give_up(x.err);
    const double t_asm = now() - t0;
    fhmesh::ParallelWalker W(o);
    W.scratch = &ctx->mesh_first; W.scratch_cap = &ctx->mesh_first_cap;
    W.gather = [&](const uint32_t* idx, size_t n, fhmesh::V3* out) {
        if (!n) return true;
        uint32_t* d_idx = (uint32_t*)x.alloc(n * 4);
        fhmesh::V3* d_out = (fhmesh::V3*)x.alloc(n * sizeof(fhmesh::V3));
        if (!d_idx || !d_out) return false;
        x.chk(hipMemcpyAsync(d_idx, idx, n * 4, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(fhm::k_oct_gather, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const fhmesh::V3*)oo.verts, (const uint32_t*)d_idx, d_out, (uint32_t)n);
        x.chk(hipGetLastError());
        x.chk(hipMemcpyAsync(out, d_out, n * sizeof(fhmesh::V3), hipMemcpyDeviceToHost, ctx->stream));
        x.chk(hipStreamSynchronize(ctx->stream));
        return x.err == hipSuccess;
    };
    W.run();
    const double t_walk = now() - t0 - t_asm;
    if (W.gather_failed) return give_up(x.err != hipSuccess ? x.err : hipErrorOutOfMemory);
    x.release();
    M->octree_cells = oo.n_blocks; M->octree_verts = oo.n_verts;
    M->vertices.swap(W.vertices);
    M->triangles.swap(W.triangles);
    if (T.on)
        fprintf(stderr, "fhip mesh depth %u: cells %.4f s (%llu evaluated), leaf kernel %.4f s (%u leaves), assembly on the device + cells to the host %.4f s (%u blocks, %u vertices), "
                        "dual walk + the mesh's vertices gathered %.4f s, total %.4f s\n",
                depth, T.t_cells, (unsigned long long)M->cells_evaluated, T.t_leaf, T.n_leaf_cells, t_asm, oo.n_blocks, oo.n_verts, t_walk, now() - T.t_start);
    return hipSuccess;
}
fhip_status fhip_mesh_sample(fhip_ctx* ctx, const fhip_tape* tape, uint32_t depth, const float* world_to_model, const int32_t* axis_slots,
                             const uint64_t* var_keys, const float* var_values, uint32_t n_vars, fhip_mesh** out) {
    return mesh_run(ctx, tape, depth, world_to_model, axis_slots, var_keys, var_values, n_vars, MESH_SAMPLE, 0, 1, out);
}
// Octree::build + Octree::walk_dual (octree.rs:48-68, 219-225): fhip_mesh_sample, then the octree assembled from the device's
// results (cell collapse included) and the dual walk on the host
fhip_status fhip_mesh_build(fhip_ctx* ctx, const fhip_tape* tape, uint32_t depth, const float* world_to_model, const int32_t* axis_slots,
                            const uint64_t* var_keys, const float* var_values, uint32_t n_vars, fhip_mesh** out) {
    return mesh_run(ctx, tape, depth, world_to_model, axis_slots, var_keys, var_values, n_vars, MESH_BUILD, 0, 1, out);
}
// ---- the build sharded by the root's octants (Octree::build_inner_mt, octree.rs:94-210, across GPUs): every part runs the
// device side for its octants; the parts' results travel as flat buffers to one place, where fhip_mesh_merge puts the level
// arrays together (slots of later parts shifted by the ambiguous cells before them) and runs assembly and dual walk
fhip_status fhip_mesh_sample_part(fhip_ctx* ctx, const fhip_tape* tape, uint32_t depth, const float* world_to_model, const int32_t* axis_slots,
                                  const uint64_t* var_keys, const float* var_values, uint32_t n_vars, uint32_t part, uint32_t n_parts, fhip_mesh** out) {
    return mesh_run(ctx, tape, depth, world_to_model, axis_slots, var_keys, var_values, n_vars, MESH_PART, part, n_parts, out);
}
namespace {
struct MeshPartHeader {       // followed by n_levels u64 level sizes, then per level {cls bytes padded to 8, slot words padded to 8}, then the leaf records
    uint32_t magic, version, depth, part, n_parts, n_levels, leaf_size, pad;
    uint64_t n_leaves, cells_evaluated, full, empty;
};
constexpr uint32_t MESH_PART_MAGIC = 0x504d4846u;     // "FHMP"
inline uint64_t pad8(uint64_t n) { return (n + 7) & ~7ull; }
}
uint64_t fhip_mesh_part_bytes(const fhip_mesh* m) {
    uint64_t n = sizeof(MeshPartHeader) + 8ull * m->cls.size();
    for (auto& c : m->cls) n += pad8(c.size()) + pad8(4ull * c.size());
    return n + (uint64_t)m->leaves.size() * sizeof(FhMeshLeaf);
}
void fhip_mesh_part_export(const fhip_mesh* m, void* out) {
    char* p = (char*)out;
    MeshPartHeader h;
    memset(&h, 0, sizeof(h));
    h.magic = MESH_PART_MAGIC; h.version = 1; h.depth = m->depth; h.part = m->part; h.n_parts = m->n_parts; h.n_levels = (uint32_t)m->cls.size();
    h.leaf_size = (uint32_t)sizeof(FhMeshLeaf); h.n_leaves = m->leaves.size(); h.cells_evaluated = m->cells_evaluated; h.full = m->full; h.empty = m->empty;
    memcpy(p, &h, sizeof(h)); p += sizeof(h);
    for (auto& c : m->cls) { const uint64_t n = c.size(); memcpy(p, &n, 8); p += 8; }
    for (size_t d = 0; d < m->cls.size(); d++) {
        const size_t n = m->cls[d].size();
        memset(p, 0, pad8(n)); memcpy(p, m->cls[d].data(), n); p += pad8(n);
        memset(p, 0, pad8(4 * n)); memcpy(p, m->slot[d].data(), 4 * n); p += pad8(4 * n);
    }
    if (m->leaves.size()) memcpy(p, m->leaves.data(), m->leaves.size() * sizeof(FhMeshLeaf));
}
fhip_status fhip_mesh_merge(fhip_ctx* ctx, const void* const* parts, const uint64_t* part_bytes, uint32_t n_parts, const float* world_to_model, fhip_mesh** out) {
    if (!out) return FHIP_ERR_BAD_TAPE;
    *out = nullptr;
    if (!parts || !part_bytes || n_parts < 1 || n_parts > 8) return fail(ctx, FHIP_ERR_UNSUPPORTED, "mesh merge: 1..8 parts");
    struct View { MeshPartHeader h; const uint64_t* level_n; std::vector<const uint8_t*> cls; std::vector<const uint32_t*> slot; const FhMeshLeaf* leaves; };
    std::vector<View> V(n_parts);
    for (uint32_t k = 0; k < n_parts; k++) {       // part k of the array must BE part k
        const char* p = (const char*)parts[k];
        if (!p || part_bytes[k] < sizeof(MeshPartHeader)) return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: short part");
        View& v = V[k];
        memcpy(&v.h, p, sizeof(v.h));
        if (v.h.magic != MESH_PART_MAGIC || v.h.version != 1 || v.h.leaf_size != sizeof(FhMeshLeaf)) return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: not a mesh part of this library");
        if (v.h.n_parts != n_parts || v.h.part != k || v.h.depth != V[0].h.depth || v.h.n_levels < 1 || v.h.n_levels > 21)
            return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: parts do not belong together (part index, part count or depth)");
        uint64_t need = sizeof(MeshPartHeader) + 8ull * v.h.n_levels;
        if (part_bytes[k] < need) return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: short part");
        v.level_n = (const uint64_t*)(p + sizeof(MeshPartHeader));
        const char* q = p + need;
        for (uint32_t d = 0; d < v.h.n_levels; d++) {
            const uint64_t n = v.level_n[d];
            if (n > (1ull << 30)) return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: level size");
            need += pad8(n) + pad8(4 * n);
            if (part_bytes[k] < need) return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: short part");
            v.cls.push_back((const uint8_t*)q); q += pad8(n);
            v.slot.push_back((const uint32_t*)q); q += pad8(4 * n);
        }
        if (part_bytes[k] < need + v.h.n_leaves * sizeof(FhMeshLeaf)) return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: short part");
        v.leaves = (const FhMeshLeaf*)q;
    }
    const uint32_t depth = V[0].h.depth;
    fhip_mesh* M = new fhip_mesh();
    M->depth = depth;
    // the root: evaluated by every part, with the same result
    for (uint32_t k = 0; k < n_parts; k++)
        if (V[k].level_n[0] != 1 || V[k].cls[0][0] != V[0].cls[0][0]) { delete M; return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: the parts disagree about the root cell"); }
    const bool whole = n_parts == 1 || V[0].h.n_levels == 1;       // nothing below the root (decided, or a leaf at depth 0): part 0 has it all
    const uint32_t np = whole ? 1 : n_parts;
    uint32_t levels = 0;
    for (uint32_t k = 0; k < np; k++) levels = std::max(levels, V[k].h.n_levels);
    M->cells_evaluated = 1; M->full = 0; M->empty = 0;
    for (uint32_t k = 0; k < np; k++) { M->cells_evaluated += V[k].h.cells_evaluated - 1; M->full += V[k].h.full; M->empty += V[k].h.empty; }
    if (!whole && V[0].cls[0][0] != 3) { delete M; return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: levels below a decided root"); }
    std::vector<uint64_t> shift(np, 0);       // slots of part k at the level before: + shift[k]
    M->cls.resize(levels); M->slot.resize(levels);
    for (uint32_t d = 0; d < levels; d++) {
        std::vector<uint8_t>& C = M->cls[d];
        std::vector<uint32_t>& S = M->slot[d];
        std::vector<uint64_t> amb(np, 0);
        if (d == 0) { C.assign(1, V[0].cls[0][0]); S.assign(1, V[0].slot[0][0]); if (!whole) S[0] = 0; amb.assign(np, 0); }
        else if (d == 1 && !whole) {     // the root's eight children, each from the part that owns it
            C.assign(8, 0); S.assign(8, 0xFFFFFFFFu);
            for (uint32_t k = 0; k < np; k++) {
                if (V[k].h.n_levels < 2 || V[k].level_n[1] != 8) { delete M; return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: a part without the root's children"); }
                const uint32_t mask = mesh_part_mask(k, n_parts);
                for (uint32_t o = 0; o < 8; o++) {
                    const uint8_t c = V[k].cls[1][o];
                    if (((mask >> o) & 1u) != (c != 0 ? 1u : 0u)) { delete M; return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: a part covers the wrong octants"); }
                    if (c == 3) amb[k]++;
                }
            }
            uint64_t off = 0;
            for (uint32_t k = 0; k < np; k++) {
                for (uint32_t o = 0; o < 8; o++) if (V[k].cls[1][o]) { C[o] = V[k].cls[1][o]; S[o] = V[k].cls[1][o] == 3 ? (uint32_t)(off + V[k].slot[1][o]) : 0xFFFFFFFFu; }
                shift[k] = off; off += amb[k];
            }
            continue;
        } else {
            // children of the level above's ambiguous cells: part k's array sits at 8 * (its slots' shift at the level above)
            uint64_t total = 0;
            for (uint32_t k = 0; k < np; k++) total += V[k].h.n_levels > d ? V[k].level_n[d] : 0;
            C.resize(total); S.resize(total);
            // (two passes over the parts, each on the host's threads: the ambiguous cells of every part, then the copies with
            //  the slots shifted by the ambiguous cells of the parts before)
            std::vector<uint64_t> n_of(np, 0), at_of(np, 0), next_shift(np, 0);
            uint64_t at = 0;
            for (uint32_t k = 0; k < np; k++) {
                n_of[k] = V[k].h.n_levels > d ? V[k].level_n[d] : 0;
                if (at != shift[k] * 8 && n_of[k]) { delete M; return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: level arrays do not line up"); }
                at_of[k] = at; at += n_of[k];
            }
            fhmesh::parallel_for(np, [&](size_t k) {
                uint64_t a = 0;
                const uint8_t* c = n_of[k] ? V[k].cls[d] : nullptr;
                for (uint64_t i = 0; i < n_of[k]; i++) a += c[i] == 3;
                amb[k] = a;
            });
            uint64_t off = 0;
            for (uint32_t k = 0; k < np; k++) { next_shift[k] = off; off += amb[k]; }
            constexpr uint64_t CHUNK = 1u << 20;
            std::vector<std::array<uint64_t, 3>> jobs;      // part, first cell, cells
            for (uint32_t k = 0; k < np; k++) for (uint64_t i = 0; i < n_of[k]; i += CHUNK) jobs.push_back({k, i, std::min(CHUNK, n_of[k] - i)});
            fhmesh::parallel_for(jobs.size(), [&](size_t j) {
                const uint32_t k = (uint32_t)jobs[j][0];
                const uint8_t* c = V[k].cls[d] + jobs[j][1];
                const uint32_t* sl = V[k].slot[d] + jobs[j][1];
                uint8_t* co = C.data() + at_of[k] + jobs[j][1];
                uint32_t* so = S.data() + at_of[k] + jobs[j][1];
                const uint32_t sh = (uint32_t)next_shift[k];
                for (uint64_t i = 0; i < jobs[j][2]; i++) { co[i] = c[i]; so[i] = c[i] == 3 ? sh + sl[i] : 0xFFFFFFFFu; }
            });
            shift = next_shift;
            continue;
        }
    }
    // leaf records: in part order (= slot order at the leaf depth), left where they are
    uint64_t n_leaves = 0;
    M->leaves.seg_start.push_back(0);
    for (uint32_t k = 0; k < np; k++) {
        M->leaves.seg_p.push_back(V[k].leaves);
        n_leaves += V[k].h.n_leaves;
        M->leaves.seg_start.push_back(n_leaves);
    }
    M->leaves.n = n_leaves;
    M->ambiguous_leaves = n_leaves;
    for (uint32_t d = 0; d < levels; d++) M->per_level.push_back(M->cls[d].size());
    float mat[16];
    bool ident = true;
    if (world_to_model) for (int i = 0; i < 16; i++) { mat[i] = world_to_model[i]; ident &= world_to_model[i] == ((i % 5 == 0) ? 1.0f : 0.0f); }
    MeshTimes MT{getenv("FHIP_MESH_TIMES") != nullptr, std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(), 0, 0, 0, (uint32_t)n_leaves};
    mesh_assemble(ctx, M, depth, world_to_model && !ident, mat, MT);
    *out = M;
    return FHIP_OK;
}
void fhip_debug_walk_dual(const uint32_t* cells, uint64_t n_cells, const uint32_t* root, const float* verts, uint64_t n_verts, int parallel,
                          uint64_t counts[2], uint64_t* tris, float* verts_out) {
    fhmesh::Octree o;
    auto cell = [](const uint32_t* w) { fhmesh::Cell c; c.kind = (uint8_t)w[0]; c.mask = (uint8_t)w[1]; c.index = w[2]; return c; };
    o.root = cell(root);
    o.cells.resize(n_cells);
    for (uint64_t i = 0; i < n_cells; i++) for (int k = 0; k < 8; k++) o.cells[i][k] = cell(cells + (i * 8 + k) * 3);
    o.verts.resize(n_verts);
    for (uint64_t i = 0; i < n_verts; i++) o.verts[i] = fhmesh::V3{verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]};
    fhmesh::TriVec t;
    fhmesh::VertVec v;
    if (parallel == 2) {     // as fhip_mesh_build runs it: the cells through a view, the octree's vertices never read - the mesh's are gathered afterwards
        fhmesh::Octree w;
        w.root = o.root;
        w.cells_view = o.cells.data(); w.n_cells_view = o.cells.size(); w.n_verts_view = o.verts.size();
        fhmesh::ParallelWalker W(w);
        W.gather = [&](const uint32_t* idx, size_t n, fhmesh::V3* out) { for (size_t i = 0; i < n; i++) out[i] = o.verts[idx[i]]; return true; };
        W.run();
        t.swap(W.triangles); v.swap(W.vertices);
    } else if (parallel) { fhmesh::ParallelWalker W(o); W.run(); t.swap(W.triangles); v.swap(W.vertices); }
    else { fhmesh::Walker W(o); W.cell(fhmesh::CellRef()); t.swap(W.triangles); v.swap(W.vertices); }
    counts[0] = t.size(); counts[1] = v.size();
    if (tris) memcpy(tris, t.data(), t.size() * 24);
    if (verts_out) memcpy(verts_out, v.data(), v.size() * 12);
}
void fhip_mesh_vertices(const fhip_mesh* m, float* out) { memcpy(out, m->vertices.data(), m->vertices.size() * 12); }
void fhip_mesh_triangles(const fhip_mesh* m, uint64_t* out) { memcpy(out, m->triangles.data(), m->triangles.size() * 24); }
void fhip_mesh_free(fhip_mesh* m) { delete m; }
// out = {cells evaluated (= interval evaluations), Full, Empty, ambiguous cells at the leaf depth (= calls of leaf()), bytes per leaf record, levels}
void fhip_mesh_counts(const fhip_mesh* m, uint64_t out[8]) {
    out[0] = m->cells_evaluated; out[1] = m->full; out[2] = m->empty; out[3] = m->ambiguous_leaves; out[4] = sizeof(FhMeshLeaf);
    out[5] = m->per_level.size(); out[6] = m->vertices.size(); out[7] = m->triangles.size();
}
void fhip_mesh_leaves(const fhip_mesh* m, void* out) { memcpy(out, m->leaves.data(), m->leaves.size() * sizeof(FhMeshLeaf)); }

// ---- profiling -------------------------------------------------------------------------
void fhip_profile_enable(fhip_ctx* ctx, int on) { ctx->profiling = on != 0; }
fhip_status fhip_profile_read(fhip_ctx* ctx, double ms[4], uint32_t launches[4]) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 4; i++) { ms[i] = 0; launches[i] = 0; }
    for (auto& e : ctx->prof_events) {
        float t = 0;
        if (hipEventElapsedTime(&t, e.second.first, e.second.second) == hipSuccess) { ms[e.first] += t; launches[e.first]++; }
    }
    return FHIP_OK;
}
fhip_status fhip_profile_read_kernels(fhip_ctx* ctx, double ms[8], uint32_t launches[8]) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 8; i++) { ms[i] = 0; launches[i] = 0; }
    for (auto& e : ctx->asm_events) {
        float t = 0;
        if (e.first < 8 && hipEventElapsedTime(&t, e.second.first, e.second.second) == hipSuccess) { ms[e.first] += t; launches[e.first]++; }
    }
    return FHIP_OK;
}
fhip_status fhip_render_counters(fhip_ctx* ctx, uint64_t out[8]) {
    fhip_status st = finish_render(ctx);
    if (st != FHIP_OK && st != FHIP_ERR_OVERFLOW) return st;
    const FhRenderState& S = ctx->last_state;
    out[0] = S.arena_head; out[1] = S.arena_overflow; out[2] = S.n_leaves; out[3] = S.queue_overflow;
    for (int i = 0; i < 4; i++) out[4 + i] = S.count[i + 1];
    return FHIP_OK;
}

// Diagnostics: per-kernel-kind wave busy statistics of the last render (see WaveProbe)
fhip_status fhip_debug_stats(fhip_ctx* ctx, uint64_t out[64]) {
    fhip_status st = finish_render(ctx);
    if (st != FHIP_OK && st != FHIP_ERR_OVERFLOW) return st;
    for (int i = 0; i < 64; i++) out[i] = ctx->last_state.stat[i];
    return FHIP_OK;
}
// Diagnostics: the links of a tape as the linked prune gets them (host_graph.hpp compute_links); 0: the tape does not qualify
uint32_t fhip_debug_tape_links(const fhip_tape* tape, uint64_t* out, uint32_t cap) {
    std::vector<uint64_t> lk;
    std::vector<uint64_t> cops;
    if (!fh::compute_links(tape->t, lk, cops)) return 0;
    for (size_t i = 0; i < lk.size() && i < cap; i++) out[i] = lk[i];
    return (uint32_t)lk.size();
}
// ... and the leaf stage's counters of the last (profiled) 3D frame: render_state.h leaf_stat
fhip_status fhip_debug_leaf_stats(fhip_ctx* ctx, uint64_t out[8]) {
    fhip_status st = finish_render(ctx);
    if (st != FHIP_OK && st != FHIP_ERR_OVERFLOW) return st;
    for (int i = 0; i < 8; i++) out[i] = ctx->last_state.leaf_stat[i];
    return FHIP_OK;
}

// Diagnostics: the leaves (24-byte FhLeaf records) of the last slab of the last 3D frame
uint32_t fhip_debug_leaves(fhip_ctx* ctx, void* out, uint32_t cap) {
    if (finish_render(ctx) != FHIP_OK) return 0;
    const uint32_t n = std::min(std::min(ctx->last_state.n_leaves, ctx->last_state.leaf_cap), cap);
    if (hipMemcpy(out, ctx->last_state.leaves, (size_t)n * sizeof(FhLeaf), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
}

// Diagnostics: the work-queue entries (36-byte FhGroup records) the last 3D frame left behind: kind 0 = queue of
// tile level `index`, kind 1 = parked queue of z-slab `index`.  counts[0] = entries of the small-layout half (written
// first), counts[1] = of the other half.  Returns the number of records written.
uint32_t fhip_debug_groups(fhip_ctx* ctx, int kind, uint32_t index, void* out, uint32_t cap, uint32_t counts[2]) {
    counts[0] = counts[1] = 0;
    if (finish_render(ctx) != FHIP_OK) return 0;
    const FhRenderState& S = ctx->last_state;
    const FhGroup* base; uint32_t ns, nb, qcap;
    if (kind == 0) {
        if (index >= FH_MAX_LEVELS || !S.queue[index]) return 0;
        base = S.queue[index]; ns = S.count[index]; nb = S.count_big[index]; qcap = S.qcap[index];
    } else {
        if (index >= FH_MAX_SLABS || !S.squeue) return 0;
        base = S.squeue + (size_t)index * S.squeue_cap; ns = S.scount[index]; nb = S.scount_big[index]; qcap = S.squeue_cap;
    }
    ns = std::min(ns, qcap); nb = std::min(nb, qcap - ns);
    const uint32_t n0 = std::min(ns, cap), n1 = std::min(nb, cap - n0);
    if (n0 && hipMemcpy(out, base, (size_t)n0 * sizeof(FhGroup), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    if (n1 && hipMemcpy((FhGroup*)out + n0, base + (qcap - nb), (size_t)n1 * sizeof(FhGroup), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    counts[0] = n0; counts[1] = n1;
    return n0 + n1;
}

// Diagnostics: the ISA probe kernel (gen_interp.py gen_probe): 16 rows of 64 floats
fhip_status fhip_debug_probe(fhip_ctx* ctx, float* out) {
    HIP_TRY(ctx, ctx->io_a.ensure(16 * 256));
    struct { void* p; } ka = {ctx->io_a.p};
    if (launch_asm(ctx, FH_ASM_PROBE, 1, &ka, sizeof(ka)) != hipSuccess) return FHIP_ERR_HIP;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, ctx->io_a.p, 16 * 256, hipMemcpyDeviceToHost));
    return FHIP_OK;
}

// Diagnostics: instruction-cost micro-benchmark `test` (gen_ubench.py) on `n_waves` single-wave workgroups; out[w] = shader
// clocks per pattern for wave w
fhip_status fhip_debug_ubench(fhip_ctx* ctx, uint32_t test, uint32_t iters, uint32_t n_waves, float* out) {
    HIP_TRY(ctx, ctx->io_a.ensure((size_t)n_waves * 4));
    HIP_TRY(ctx, hipMemsetAsync(ctx->io_a.p, 0, (size_t)n_waves * 4, ctx->stream));
    struct { void* p; uint32_t test, iters; } ka = {ctx->io_a.p, test, iters};
    if (launch_asm(ctx, FH_ASM_UBENCH, n_waves, &ka, sizeof(ka), 64) != hipSuccess) return FHIP_ERR_HIP;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, ctx->io_a.p, (size_t)n_waves * 4, hipMemcpyDeviceToHost));
    return FHIP_OK;
}

// Diagnostics: accuracy of transcendental opcode `op` (0 sin 1 cos 2 tan 3 asin 4 acos 5 atan 6 exp 7 ln) against `ref` (host
// libm results for the floats with bit patterns first + i * stride): out = {max ulp, differing, > 1 ulp, input bits of the worst}
fhip_status fhip_debug_math_sweep(fhip_ctx* ctx, int op, uint32_t first, uint32_t stride, uint64_t n, const float* ref, uint64_t out[4]) {
    (void)hipSetDevice(ctx->device);
    HIP_TRY(ctx, ctx->io_a.ensure(n * 4));
    HIP_TRY(ctx, ctx->io_b.ensure(64));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->io_a.p, ref, n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->io_b.p, 0, 64, ctx->stream));
    hipLaunchKernelGGL(k_math_sweep, dim3(ctx->n_cu * 16), dim3(256), 0, ctx->stream, op, first, stride, (size_t)n, (const float*)ctx->io_a.p,
                       (unsigned long long*)ctx->io_b.p);
    HIP_TRY(ctx, hipGetLastError());
    unsigned long long r[4];
    HIP_TRY(ctx, hipMemcpyAsync(r, ctx->io_b.p, 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    out[0] = r[0] >> 32; out[1] = r[1]; out[2] = r[2]; out[3] = r[0] & 0xFFFFFFFFull;
    return FHIP_OK;
}

// Diagnostics: `n` ops of the tape arena starting at op `off` (the tapes the last frame left there)
uint32_t fhip_debug_arena(fhip_ctx* ctx, uint32_t off, uint32_t n, uint64_t* out) {
    if ((size_t)(off + (size_t)n) * 8 > ctx->arena_bytes) return 0;
    if (hipMemcpy(out, (const uint64_t*)ctx->arena.p + off, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
}

// Diagnostics: time `reps` passes of the point interpreter over `tape` in `n_waves` waves.
// variant: 0 = VGPR file 16 regs x 4, 1 = VGPR 32 x 2, 2 = LDS file, 3 = VGPR 32 x 1
fhip_status fhip_debug_bench(fhip_ctx* ctx, const fhip_tape* tape, uint32_t n_waves, uint32_t reps, int variant, double* ms) {
    fhip_status st = tape_to_device(ctx, tape);
    if (st) return st;
    HIP_TRY(ctx, ctx->state.ensure(2 * sizeof(FhRenderState)));
    FhRenderState S;
    memset(&S, 0, sizeof(S));
    for (int i = 0; i < FH_MAX_INPUTS; i++) S.P.in_kind[i] = i % 3;
    HIP_TRY(ctx, hipMemcpy(ctx->state.p, &S, sizeof(S), hipMemcpyHostToDevice));
    HIP_TRY(ctx, ctx->io_a.ensure((size_t)n_waves * WAVE * 4));
    hipEvent_t a, b;
    HIP_TRY(ctx, hipEventCreate(&a));
    HIP_TRY(ctx, hipEventCreate(&b));
    const uint32_t len = (uint32_t)tape->t.ops.size();
    FhRenderState* dS = (FhRenderState*)ctx->state.p;
    for (int it = 0; it < 2; it++) {
        HIP_TRY(ctx, hipEventRecord(a, ctx->stream));
        if (variant == 0) hipLaunchKernelGGL((k_bench_points<16, 4>), dim3(n_waves), dim3(WAVE), 0, ctx->stream, dS, tape->d_ops, len, reps, (float*)ctx->io_a.p);
        else if (variant == 1) hipLaunchKernelGGL((k_bench_points<32, 2>), dim3(n_waves), dim3(WAVE), 0, ctx->stream, dS, tape->d_ops, len, reps, (float*)ctx->io_a.p);
        else if (variant == 3) hipLaunchKernelGGL((k_bench_points<32, 1>), dim3(n_waves), dim3(WAVE), 0, ctx->stream, dS, tape->d_ops, len, reps, (float*)ctx->io_a.p);
        else hipLaunchKernelGGL((k_bench_points<0, 1>), dim3(n_waves), dim3(WAVE), (size_t)std::max<uint32_t>(tape->t.n_regs, 1) * WAVE * 4, ctx->stream, dS, tape->d_ops, len, reps, (float*)ctx->io_a.p);
        HIP_TRY(ctx, hipEventRecord(b, ctx->stream));
        HIP_TRY(ctx, hipEventSynchronize(b));
    }
    float t = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&t, a, b));
    *ms = t;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return FHIP_OK;
}

// ---- host graph ------------------------------------------------------------------------
static const int UNARY_MAP[] = {FH_NEG, FH_ABS, FH_RECIP, FH_SQRT, FH_SQUARE, FH_FLOOR, FH_CEIL, FH_ROUND, FH_SIN,
                                FH_COS, FH_TAN, FH_ASIN, FH_ACOS, FH_ATAN, FH_EXP, FH_LN, FH_NOT, FH_RAND};
// BinaryOpcode order (context/op.rs:35-48): Add Sub Mul Div Atan Min Max Compare Mod And Or Mix
static const int BINARY_MAP[] = {FH_ADD_RR, FH_SUB_RR, FH_MUL_RR, FH_DIV_RR, FH_ATAN2_RR, FH_MIN_RR, FH_MAX_RR,
                                 FH_COMPARE_RR, FH_MOD_RR, FH_AND_RR, FH_OR_RR, FH_MIX_RR};
fhip_graph* fhip_graph_new(void) { return new fhip_graph(); }
void fhip_graph_free(fhip_graph* g) { delete g; }
uint32_t fhip_graph_len(const fhip_graph* g) { return (uint32_t)g->g.nodes.size(); }
uint32_t fhip_graph_var(fhip_graph* g, int kind, uint64_t index) { return g->g.var((uint8_t)kind, kind < 3 ? 0 : index); }
uint32_t fhip_graph_constant(fhip_graph* g, float v) { return g->g.constant(v); }
uint32_t fhip_graph_unary(fhip_graph* g, int opcode, uint32_t a) {
    if (opcode < 0 || opcode >= 18) return fh::NO_NODE;
    return g->g.unary(UNARY_MAP[opcode], a);
}
uint32_t fhip_graph_binary(fhip_graph* g, int opcode, uint32_t a, uint32_t b) {
    if (opcode < 0 || opcode >= 12) return fh::NO_NODE;
    return g->g.binary(BINARY_MAP[opcode], a, b);
}
uint32_t fhip_graph_from_text(fhip_graph* g, const char* text) {
    std::string err;
    return g->g.parse(text, err);
}

}  // extern "C"
