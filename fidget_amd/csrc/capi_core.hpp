// Fragment of capi.hip (context, device buffers, error plumbing, kernel launch helpers: everything the sections of the C ABI share (before `extern "C"`)); not a stand-alone header: included by capi.hip only.
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
    mutable std::atomic<uint32_t> input_slots{0x80000000u};   // the input slots the tape reads, bit per slot (bit 31: not looked up yet)
    mutable uint32_t n_chain = 0;          // ... and, behind that table, the root chain's ops in evaluation order: choice ordinal | op index << 16 (plan.chain)
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
enum { FH_ASM_COLUMNS = 0, FH_ASM_FLOAT_16x4, FH_ASM_FLOAT_32x2, FH_ASM_TILES, FH_ASM_PRUNE1, FH_ASM_TILES_V32, FH_ASM_TILES_V64, FH_ASM_PROBE, FH_ASM_UBENCH, FH_ASM_COLUMNS_T, FH_ASM_NORMALS, FH_ASM_NORMALS_T, FH_ASM_TILES_T, FH_ASM_TILES_V32_T, FH_ASM_TILES_V64_T, FH_ASM_FLOAT_16x4_T, FH_ASM_FLOAT_32x2_T, FH_ASM_TRANS_PROBE, FH_ASM_COUNT };
static const char* const FH_ASM_NAMES[FH_ASM_COUNT] = {"fh_columns", "fh_float_eval_16x4", "fh_float_eval_32x2", "fh_tiles", "fh_prune1",
                                                       "fh_tiles_v32", "fh_tiles_v64", "fh_probe", "fh_ubench", "fh_columns_t", "fh_normals", "fh_normals_t", "fh_tiles_t", "fh_tiles_v32_t", "fh_tiles_v64_t", "fh_float_eval_16x4_t", "fh_float_eval_32x2_t", "fh_trans_probe"};
// register-file shapes of the VGPR tile kernels (gen_tilesv.py): registers, choices
static const uint32_t V32_REGS = 32, V32_CHOICES = 256, V64_REGS = 64, V64_CHOICES = 512;

// Behaviour switches of a context - diagnostics and tuning, none is needed in normal use.  They are part of the context, not of
// the process: read ONCE from the environment when the context is created (FHIP_<NAME IN CAPITALS>, for runs of unmodified
// programs under a switch) and changed afterwards only through fhip_ctx_set_option(ctx, "<name>", value) - nothing in a
// render's launch path looks at the environment.  name, default; DESIGN.md section 5 says what each one selects.
#define FH_OPTION_LIST(X)                                                                                                       \
    /* kernel selection (each falls back to the HIP C++ path of the same stage, which tapes outside the assembly set take anyway) */ \
    X(no_asm, 0) X(no_split, 0) X(no_asm_tiles, 0) X(no_tiles_v, 0) X(no_asm_tiles_t, 0) X(no_columns_t, 0) X(no_asm_normals, 0)   \
    X(no_tape_groups, 0) X(prune2, 1)                                                                                           \
    /* short cuts: column invariance off everywhere; no_zrep 1 = no sharing of tiles along z at all, 2 = only not at the root level, 3 = every slab rendered;     \
       column_walk: the leaf kernel by footprint columns - 1 in frames whose tapes read no z, 0 never, 2 always; column_group: every other frame by groups of 2^g layers of a column (0: by blocks of four footprints of a layer) */                  \
    X(no_column_inv, 0) X(no_zrep, 0) X(root32_max, 4096) X(column_walk, 1) X(column_group, 2)                                                    \
    /* pipelining and resources */                                                                                             \
    X(no_pipeline, 0) X(frame_lanes, 4) X(lanes_tune, 1) X(lanes_fail, 0) X(slab_layers, 4) X(arena_mb, 4096)  \
    /* mesher */                                                                                                               \
    X(mesh_device_assembly, 1) X(mesh_device_walk, 1) X(mesh_simplify_min_ops, 256)                                             \
    /* diagnostics (stats: bit 0 device counters and clocks, bit 1 the host thread's time per frame to stderr) */                                                                                                          \
    X(probe, 0) X(stats, 0)
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

// Everything one frame of a render owns on the device.  A context holds several sets (4): an asynchronous 3D render
// takes the set used longest ago, so that its coarse levels (which keep a few hundred waves busy for most of a millisecond)
// run on a stream of their own beside the previous frames' slabs (frame pipelining; option no_pipeline turns all pipelining off).
struct FrameBufs {
    DevBuf state, arena, leaves, leaf_table, zbuf, normals, fp_lists, mind, squeue, slots[2], leaves_b, leaf_table_b, fp_lists_b, chw[2], tvals, topch, chwr, gscratch, rare_scratch;
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
                          &leaves_b, &leaf_table_b, &fp_lists_b, &chw[0], &chw[1], &tvals, &topch, &chwr, &gscratch, &rare_scratch};
        for (DevBuf* b : bufs) b->release();
        for (auto& q : queue) q.release();
        if (ev_done) (void)hipEventDestroy(ev_done);
        ev_done = nullptr;
    }
};
struct fhip_ctx : FrameBufs {
    FhOptions opt;                  // behaviour switches (FH_OPTION_LIST): environment at creation, fhip_ctx_set_option later
    // the sets of the frames before the current one: a frame takes the set used longest ago (ring of four).  A set is free again when its frame is complete, and a pipelined frame takes ~1.5 ms from its first
    // coarse-level kernel to its image: the period cannot be shorter than that over the number of sets.  Three were enough while the tail
    // stream bounded the pipeline; since round 4 (capi_render.hpp tail_on_main) three give frames of 0.44 / 0.59 / 0.48 ms in turn, four
    // 0.49 each, five the same (profiles/r04k)
#define FH_EXTRA_SETS 4
    FrameBufs others[FH_EXTRA_SETS];
    uint32_t extra_sets = 3;     // (four sets: more did not raise the pipelined rate, fewer lowered it - round 4)
    bool frame_pipeline = true;
    hipStream_t stream_pre = nullptr;   // coarse levels of a pipelined frame
    uint32_t pre_turn = 0;              // ... frames whose root levels alternate between it and the tail stream: whose turn
    hipEvent_t ev_rest_fork = nullptr, ev_rest_join = nullptr;
    hipEvent_t ev_pre = nullptr, ev_l0 = nullptr, ev_l1 = nullptr;
    hipStream_t post_v64_stream = nullptr;   // launch_tiles_split: where the launches behind level 1's fh_tiles_v64 go (side_only_l1), or null
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
    // A frame lane (a child context) reads its PARENT's flag: fhip_cancel comes from another thread and must not walk `lanes`, which the
    // render thread grows, clears and frees (run_on_lane_, lanes_release); a parent outlives its lanes, so the pointer is always good.
    const std::atomic<int>* cancel_src = nullptr;
    // ... and a flag of the CALLER's, one byte, non-zero = cancel (fhip_cancel_watch): fidget_core::render::CancelToken is an Arc<AtomicBool>
    // whose address its into_raw() hands out - the Rust crate passes it for the duration of a render, no watcher thread
    std::atomic<const volatile unsigned char*> watch{nullptr};
    bool is_cancelled() const {
        if ((cancel_src ? cancel_src : &cancelled)->load() != 0) return true;
        const volatile unsigned char* w = watch.load(std::memory_order_relaxed);
        return w && *w != 0;
    }
    DevBuf tmp_out, io_a, io_b, io_c, io_d, io_e;
    DevBuf sticky;      // one word: a queue overflow of ANY asynchronous frame since the last fhip_ctx_sync (k_finish3d latches it)
    struct Staging { void* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool used = false; } staging[8];   // pinned (upload_frame)
    DevBuf mesh_leaves;               // fhip_mesh_build / fhip_mesh_sample: the leaf records in HBM (17 GB at depth 10), kept between builds - allocating and freeing
                                      // them per build cost 10-20 ms of every build and, once in a few, two seconds (profiles/r04a, r04h)
    void* mesh_pinned = nullptr;      // fhip_mesh_build: the leaf records' landing area on the host, kept between calls (pinning 17 GB takes over a second)
    size_t mesh_pinned_cap = 0;
    // ... and the two largest host-side temporaries of the assembly, kept for the same reason (fresh memory of that size is
    // faulted in page by page and handed back page by page): the octree's cell / vertex arrays and the dual walk's first-use table
    void* mesh_octree_cache = nullptr;       // fhmesh::Octree*
    uint32_t* mesh_first = nullptr;
    size_t mesh_first_cap = 0;
    uint32_t staging_next = 0;
    // Tape arena, per buffer set: starts at FH_ARENA_START_MB (prospero.vm at 1024^3 peaks at 79 MB per z-slab context) and is grown
    // twofold - up to option arena_mb, default 4 GiB - when a frame ran out: the frame itself is still right (children keep their parent's
    // tape, tests/test_gpu_parity.py test_render3d_survives_a_full_tape_arena), its last kernel says so in `host_flags` - a pinned word the
    // device writes and the host reads without waiting for anything - and the next render call grows the arena first.  Until round 4 every
    // set held the full 4 GiB: 18.6 GB per context for a peak use of 0.1.
    size_t arena_bytes = (size_t)256 << 20, arena_cap_bytes = (size_t)4 << 30;
    uint64_t hip_tile_frames = 0;               // frames whose tile stage took the HIP C++ kernels without having been told to (capi_render.hpp prepare)
    uint64_t substituted_tiles = 0;             // 3D frames whose caller's tile list was valid but not one the kernels take: rendered with the library's (same image)
    int last_hip_error = 0;                     // the HIP error code behind the last FHIP_ERR_HIP (HIP_TRY)
    volatile uint32_t* host_flags = nullptr;    // pinned: [0] a frame's arena overflowed, [1] peak arena ops of any frame
    bool profiling = false;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> prof_events;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> asm_events;   // ... and per assembly kernel launch
    FhRenderState last_state;
    bool have_last_state = false;
    // Frame lanes (capi_render.hpp run_on_lane; option frame_lanes): child contexts, each on one stream of its own, that take whole
    // frames of a queued sequence in turn when the frames' kernels are the 256-VGPR ones
    std::vector<fhip_ctx*> lanes;
    uint32_t lane_next = 0;
    // Rare mode of a 3D frame (capi_render.hpp): the launches that exist for tapes too large for the assembly kernels' register files - three or
    // four per slab, empty in nearly every frame - folded into launches the slab makes anyway; taken while the last finished frame met no such tape
    bool rare_now = false;            // ... this frame
    uint32_t rare_stride = 0;         // bytes of a rare block's register file in rare_scratch
    uint64_t rare_frames = 0;         // frames rendered that way so far (fhip_debug_rare_frames)
    uint64_t lane_frames = 0;         // frames that went to a lane so far (fhip_debug_lane_frames)
    uint64_t lane_frames_wanted = 0;  // ... not counting the tuner's measuring windows
    hipEvent_t ev_last = nullptr;     // the end of the last 3D frame on the caller's stream ("is the frame before still under way?")
    bool ev_last_valid = false;
    // Which of the two arrangements a queued 3D frame takes is MEASURED, per (tape, image size): consecutive queued frames of one kind
    // run a window under the stage pipeline, then one on the lanes, each timed between two events on the caller's stream, and the
    // faster arrangement is kept (capi_render.hpp lane_mode; option lanes_tune)
    struct LaneTune {
        uint64_t key = 0, used = 0;
        int phase = 0;               // 0 / 1 / 2: windows under the stage pipeline, on the lanes, under the stage pipeline again; 3 waiting for the last window's end; 4 decided
        uint32_t n = 0;              // consecutive queued frames of this kind in the current window
        bool lanes = false;          // (decided) the lanes were faster
        float ms[3] = {0.0f, 0.0f, 0.0f};  // ms per frame of the three windows
        hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    };
    std::vector<LaneTune> lane_tune;
    uint64_t tune_last_key = 0, tune_clock = 0;
    int tune_cur = -1;                // the entry of the frame being queued (frame_queued marks its window), or -1
    // ... and what a context has when it IS a lane: its stream is its own, its image goes to the caller's buffer by a copy on the
    // caller's stream
    bool is_lane = false, lane_stream_owned = false;
    DevBuf lane_img;
    hipEvent_t lane_done = nullptr, lane_copied = nullptr;
    bool lane_copied_valid = false;
};

// The cached forms of some options (what the rest of the driver reads)
#define FH_ARENA_START_MB 128
static void apply_options(fhip_ctx* c) {
    c->use_asm = c->opt.no_asm == 0;
    c->use_split = c->opt.no_split == 0;
    c->probe = c->opt.probe != 0;
    c->use_pipeline = c->opt.no_pipeline == 0;
    c->frame_pipeline = true;
    c->slab_contexts = 4;
    c->arena_cap_bytes = (size_t)std::max(1, c->opt.arena_mb) << 20;
    c->arena_bytes = std::min(c->arena_cap_bytes, std::max(c->arena_bytes, (size_t)FH_ARENA_START_MB << 20));
    c->extra_sets = 3;      // four sets in all
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
        if (e_ != hipSuccess) {                                                                                  \
            if (ctx) (ctx)->last_hip_error = (int)e_;                                                            \
            return fail(ctx, FHIP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));                   \
        }                                                                                                        \
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

// VmData::simplify (vm/data.rs:123-318) on the host: the reverse sweep over `p` with one choice per min / max / and / or op (in tape
// order), dense re-allocation of the registers (lowest free first).  false: a choice was Unknown.  (fhip_simplify; the mesher's tape
// simplification at its split level, capi_mesh.hpp)
static bool simplify_host(const fh::HostTape& p, const uint8_t* choices, fh::HostTape& out) {
    std::vector<int> map(FH_MAX_REGS, -1);
    fh::RegPool pool;
    std::vector<uint64_t> rev;
    uint32_t ci = p.n_choices, kept = 0;
    auto use = [&](uint32_t r) { if (map[r] < 0) map[r] = pool.take(); return (uint32_t)map[r]; };
    for (size_t k = p.ops.size(); k-- > 0;) {
        const uint64_t w = p.ops[k];
        const uint32_t w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
        const uint32_t op = FH_W_OP(w0), ro = FH_W_OUT(w0), ra = FH_W_A(w0), rb = w1;
        const bool is_choice = fh_is_choice(op);
        uint32_t c = FH_CHOICE_BOTH;
        if (is_choice) {
            c = choices[--ci];
            if (c == FH_CHOICE_UNKNOWN) return false;
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
    out.ops.assign(rev.rbegin(), rev.rend());
    out.n_regs = pool.high;
    out.n_choices = kept;
    out.n_outputs = p.n_outputs;
    out.n_vars = p.n_vars;  // children keep the parent's variable slots (vm/data.rs:316)
    out.vars = p.vars;
    return true;
}

// Launches of the library's own __global__ kernels in the render path: through hipModuleLaunchKernel with the arguments packed here,
// on the function handle of the kernel's symbol, looked up once per device.  A frame is ~25 launches and its queued rate had become the
// HOST's (tools/host_enqueue.py: 0.19 ms of calls per frame against 0.17 ms of kernels on the busiest stream); hipLaunchKernelGGL finds
// the kernel by its host address and marshals argument by argument on every call - 5.9 against 3.3 us per call under rocprofv3's HIP trace.
template <class T> struct FhKArgs;
template <class... K> struct FhKArgs<void (*)(K...)> {
    template <class... A> static size_t pack(char* buf, A&&... a) {
        size_t off = 0;
        auto put = [&](auto v) {
            constexpr size_t al = alignof(decltype(v));
            off = (off + al - 1) & ~(al - 1);
            memcpy(buf + off, &v, sizeof(v));
            off += sizeof(v);
        };
        (put(static_cast<K>(std::forward<A>(a))), ...);
        return off;
    }
    static constexpr size_t bytes = (sizeof(K) + ... + 0) + 16 * sizeof...(K);      // (an upper bound with padding)
};
template <auto Kernel, class... A>
static inline void fh_launch(fhip_ctx* ctx, dim3 g, dim3 b, size_t lds, hipStream_t st, A&&... a) {
    // (contexts live on their own threads: the handle of a (kernel, device) is looked up by whoever comes first, and by a second thread
    // that comes at the same time - to the same value.  Per device: a handle, and whether the look-up failed THERE; devices beyond
    // the table take the launch by symbol)
    constexpr int MAXDEV = 64;
    static std::atomic<hipFunction_t> fns[MAXDEV];
    static std::atomic<bool> no_handle[MAXDEV];
    const int device = ctx->device;
    hipFunction_t fn = nullptr;
    if (device >= 0 && device < MAXDEV) {
        fn = fns[device].load(std::memory_order_acquire);
        if (!fn && !no_handle[device].load(std::memory_order_relaxed)) {
            if (hipGetFuncBySymbol(&fn, (const void*)Kernel) != hipSuccess || !fn) { fn = nullptr; no_handle[device].store(true, std::memory_order_relaxed); (void)hipGetLastError(); }
            else fns[device].store(fn, std::memory_order_release);
        }
    }
    hipError_t e;
    if (!fn) {
        hipLaunchKernelGGL(Kernel, g, b, lds, st, std::forward<A>(a)...);
        e = hipGetLastError();
    } else {
        alignas(16) char buf[FhKArgs<decltype(Kernel)>::bytes];
        size_t bytes = FhKArgs<decltype(Kernel)>::pack(buf, std::forward<A>(a)...);
        void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, buf, HIP_LAUNCH_PARAM_BUFFER_SIZE, &bytes, HIP_LAUNCH_PARAM_END};
        e = hipModuleLaunchKernel(fn, g.x, g.y, g.z, b.x, b.y, b.z, (unsigned)lds, st, nullptr, extra);
    }
    if (e != hipSuccess) {       // (reported when the frame has been queued, as a failed assembly launch is: capi_tapes.hpp launch_asm)
        ctx->launch_failed = true;
        ctx->last_hip_error = (int)e;
        if (ctx->err.empty()) ctx->err = std::string("kernel launch: ") + hipGetErrorString(e);
        (void)hipGetLastError();
    }
}
#define FH_KLAUNCH(kernel, grid, block, lds, stream, ...) fh_launch<kernel>(ctx, grid, block, lds, stream, ##__VA_ARGS__)

