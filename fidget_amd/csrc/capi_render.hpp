// Fragment of capi.hip (the 2D / 3D tile renderers: frame set-up, the coarse levels, slabs, streams and frame pipelining); not a stand-alone header: included by capi.hip only.
// ---- renders ---------------------------------------------------------------------------
static const uint32_t VM_TILES_2D[] = {128, 32, 8};        // fidget-core/src/vm/mod.rs:255-257
static const uint32_t VM_TILES_3D[] = {128, 64, 32, 16, 8};  // fidget-core/src/vm/mod.rs:251-253
// RenderHints of the HIP shape (the reference lets every shape type pick its own, shape.rs RenderHints):
// a fan-out of 4^3 = 64 children fills a wavefront
// (128 -> 32 -> 8).  The root tile stays the one the reference's VmShape hints give for the image
// size, so that exactly the same voxels are covered (a root tile overhanging the image in z is
// evaluated there by the reference too).

static const uint32_t FH_LEAF_REGS = 40, FH_LEAF_REGS_T = 44;      // (gen_interp.py main(): fh_columns' 40 x 2 shape, fh_columns_t's 44 x 4)
static const uint32_t FH_NORMAL_REGS = 40;                         // (gen_normals.py NR)
struct RenderSetup {
    FhRenderState S;
    std::vector<FhGroup> roots;
    uint32_t n_slabs = 1, n_layers = 1;      // z-slabs (steps of the per-slab chains), root-tile layers
    uint32_t slab_lo = 0, slab_hi = 1;   // z-slabs this render covers (all of them unless the volume is split in z: octant shards)
    size_t lds_tiles_mid = 0, lds_tiles_big = 0, lds_tiles_small = 0, lds_points_big = 0, lds_normals_big = 0, lds_normals_small = 0;
    uint32_t table_words = 0, n_footprints = 0, groups_per_slab = 0;
    bool smooth_tape = false;      // the root tape has a choice in fewer than every tenth op (and more than 200 ops): a blend whose leaves stay long
    uint32_t hit_bucket_cap = 0;   // the normals kernel's work lists (k_hits3d): entries per bucket, words of the whole thing per slab context
    size_t hit_words = 0;
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
    uint32_t n_chain = 0;          // ops of the root chain (their table lies behind d_ctab's t.n_choices entries)
    uint32_t p2_cap_kept = 0;      // kept ops per child the linked prune's LDS areas are sized for (children beyond: the scalar sweep behind it)
    uint32_t exp_levels = 0;
    uint32_t col_slots = 0, col_depmask = 0, col_flags = 0;   // 3D: axis slots x | y << 8 | z << 16 (0xFF none), inputs varying along a pixel column, bit 16 projective
    bool zrep = false;        // ... column-invariant parents are evaluated for one z-layer only (k_tape_flags)
    bool xy_fixed = false, root_invariant = false;   // 3D, set before prepare(): x and y do not move along a pixel column; the ROOT tape reads nothing that does
    bool one_level_64 = false;   // 2D, a one-level list: root groups of 64 tiles through the split tile stage (render2d_frame's small-image passes)
    bool classify_only = false;  // ... and the pass that only classifies its tiles and writes their fills (no prune, no leaves)
    bool root_zrep = false;   // ... then the root level evaluates ONE layer of root tiles per z-slab and hands the result to the layers stacked on it
    bool front_only = false;  // ... and only the front slab is rendered (slab_stop = slab_hi - 1)
    bool alt_pre = false;     // ... and consecutive frames' root levels take the pre-pass and the tail stream in turn (render3d_frame)
    uint32_t slab_stop = 0;   // the slabs rendered: slab_hi - 1 down to slab_stop (= slab_lo unless front_only)
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

static std::vector<uint32_t> hip_tiles_3d(uint32_t max_size) {
    std::vector<uint32_t> v = trim_tiles(VM_TILES_3D, 5, max_size);
    std::vector<uint32_t> out{v[0]};
    for (uint32_t t = v[0]; t > 8;) { t = std::max<uint32_t>(t / 4, 8); out.push_back(t); }
    return out;
}

// 2D hint of the HIP shape: 128 -> 16 with 16 x 16 pixel leaves - what fidget-jit uses (fidget-jit/src/lib.rs:984-986); a fan-out
// of 64 children per parent fills a wavefront of the tile-stage kernels (the VM's 128 / 32 / 8 fans out by 16)
static const uint32_t HIP_TILES_2D[] = {128, 16};
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
// 3D: input slots of the axes, which inputs change along a pixel column (a z coefficient in the axis' matrix row, or a projective
// matrix), and whether the root tape reads any of them - from the camera matrix, the input binding and the tape alone (before prepare)
// (option stats, bit 1: where the host thread's time of a 3D frame goes - set-up, prepare, state upload, coarse levels, slabs, finish -
// printed every 200 frames; tools/host_enqueue.py)
struct HostSpans {
    double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t n = 0;
    std::chrono::steady_clock::time_point last;
    void start() { last = std::chrono::steady_clock::now(); }
    void mark(int k) { const auto now = std::chrono::steady_clock::now(); t[k] += std::chrono::duration<double, std::micro>(now - last).count(); last = now; }
    void frame() {
        if (++n % 200 == 0) {
            fprintf(stderr, "fidget-hip host us per frame: set-up %.1f prepare %.1f upload %.1f coarse %.1f slabs %.1f finish %.1f\n", t[0] / 200, t[1] / 200, t[2] / 200, t[3] / 200, t[4] / 200, t[5] / 200);
            for (double& x : t) x = 0;
        }
    }
};
static thread_local HostSpans g_spans;      // (one context per thread: each thread's frames, not a mix)
#define FH_SPAN(k) do { if (ctx->opt.stats & 2) g_spans.mark(k); } while (0)

static void column_setup(fhip_ctx* ctx, const fhip_tape* tape, const FhRender& P, RenderSetup& R) {
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
    R.xy_fixed = !proj && (slot[0] < 0 || !((R.col_depmask >> slot[0]) & 1)) && (slot[1] < 0 || !((R.col_depmask >> slot[1]) & 1));
    // (FHIP_NO_COLUMN_INV=1, diagnostics / bench: no column-invariance short cut anywhere - every input counts as varying
    // along z - which is what a model with z in every tape gets)
    const bool no_inv = ctx->opt.no_column_inv != 0;
    if (no_inv) R.col_depmask = 0xFFFFFFFFu;
    R.root_invariant = !no_inv && R.col_depmask != 0xFFFFFFFFu;
    if (R.root_invariant) {
        // (the input slots the tape reads, found once per tape: a frame's set-up walked the tape three times for this)
        uint32_t reads = tape->input_slots.load(std::memory_order_acquire);
        if (reads & 0x80000000u) {          // not known yet (bit 31: input slots are < 31... FH_MAX_INPUTS)
            reads = 0;
            for (uint64_t w : tape->t.ops)
                if (FH_W_OP((uint32_t)w) == FH_INPUT) reads |= 1u << ((uint32_t)(w >> 32) & 31u);
            reads &= 0x7FFFFFFFu;
            tape->input_slots.store(reads, std::memory_order_release);
        }
        R.root_invariant = (reads & R.col_depmask & 0x7FFFFFFFu) == 0;
    }
}

// A frame before this one ran out of tape arena (k_finish3d / k_latch_arena said so in the pinned host word): wait for what is in flight
// and let the sets come back twice as large, up to option arena_mb.  The frames that overflowed were right (their tiles kept their
// parents' tapes), only slower.
// Wait for the work of THIS context - its own streams and its lanes' - and nobody else's: other contexts on the device keep running
// (hipDeviceSynchronize here stalled every thread's context for an arena that belongs to one).
static hipError_t sync_own_streams(fhip_ctx* ctx) {
    hipError_t e = hipSuccess;
    for (hipStream_t s : {ctx->stream, ctx->stream2, ctx->stream3, ctx->stream_pre}) {
        const hipError_t r = hipStreamSynchronize(s);       // (a null ctx->stream is the device's default stream: the caller chose it)
        if (r != hipSuccess && e == hipSuccess) e = r;
    }
    for (fhip_ctx* L : ctx->lanes)
        if (L) { const hipError_t r = sync_own_streams(L); if (r != hipSuccess && e == hipSuccess) e = r; }
    return e;
}
// `volume_hint`: voxels of a 3D frame whose ROOT tape reads an input that changes along a pixel column (0: none such, or 2D): every
// tile of every slab then keeps a tape of its own, and the first 128 MB overflow in the first frame - which is then right but many
// times slower (children keep their parents' tapes), as are the frames until the growth below has caught up.  Sized at about a byte
// per voxel from the start instead (prospero.vm 1024^3 with z in every tape: peak 0.9 GB per set), before anything is in flight.
static fhip_status grow_arena_if_asked(fhip_ctx* ctx, size_t tape_ops, uint64_t volume_hint = 0) {
    size_t need = ((tape_ops + 64) * 8 + 4096) * 4;       // (root tape + its groups, with room to prune into)
    if (volume_hint) need = std::max<size_t>(need, std::min<uint64_t>(volume_hint, ctx->arena_cap_bytes));
    bool grow = ctx->host_flags && ctx->host_flags[0] != 0 && ctx->arena_bytes < ctx->arena_cap_bytes;
    size_t want = grow ? ctx->arena_bytes * (ctx->arena_bytes <= ((size_t)FH_ARENA_START_MB << 20) ? 4 : 2) : ctx->arena_bytes;      // (the first step is the big one: a frame that outgrows the first 128 MB is usually one with z in every tape, at 4 x the ops and more)
    if (want < need) { want = need; grow = ctx->arena_bytes < std::min(need, ctx->arena_cap_bytes); }
    if (!grow) return FHIP_OK;
    HIP_TRY(ctx, sync_own_streams(ctx));       // (every stream of the context, its lanes included: a set's arena is about to be replaced)
    ctx->host_flags[0] = 0;
    ctx->arena_bytes = std::min(ctx->arena_cap_bytes, want);
    return FHIP_OK;
}

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
    // (a one-level 2D list - the small-image passes of render2d_frame: root groups of 64 tiles - takes the 64-lane tile stage too)
    const uint32_t TL = R.tl = (fanout > 16 || (!is3d && ts.size() == 1 && R.one_level_64)) ? 64 : 16;
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
    // (a two-level list - root tiles of 32^3 straight above the leaves, what small images and parts of a frame take - gets a pre-pass of
    // its ONE coarse level; option slab_layers counts layers of 128 voxels, whatever the root tile)
    const bool prepass_ok = is3d && ts.size() >= 2 && n_layers <= FH_MAX_SLABS;
    uint32_t SL = prepass_ok ? (uint32_t)std::max(1, std::min(8, ctx->opt.slab_layers)) * std::max<uint32_t>(1, 128 / ts[0]) : 1u;
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
    // (the most registers of a leaf the leaf kernel takes - gen_interp.py main(): the largest register-file shape of fh_columns / fh_columns_t)
    R.S.leaf_asm_regs = !R.asm_points ? 32u : (R.asm_points_t ? FH_LEAF_REGS_T : FH_LEAF_REGS);
    // (fh_normals_t has the transcendental, rng and atan2 handlers; a modulo's gradient - div_euclid - keeps the C++ kernel)
    R.asm_normals = R.asm_points && !ctx->opt.no_asm_normals && (!R.asm_points_t || !tape_has_mod(t));
    R.S.norm_asm_regs = R.asm_normals ? FH_NORMAL_REGS : 32u;

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
    S.pre_levels = prepass_ok ? std::min<uint32_t>(2, (uint32_t)ts.size() - 1) : 0;

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
    R.smooth_tape = t.ops.size() > 200 && (size_t)t.n_choices * 10 < t.ops.size();
    // Column invariance at the ROOT: a root tape that reads no input varying along a pixel column, under a camera
    // that keeps x and y fixed along it, has the same interval, the same choices and the same pruned tape in every root tile of a
    // column of root tiles.  One layer per z-slab is evaluated (the slab's back-most: FhGroup::x = how many layers of the slab it stands
    // for) and the push stage hands the result to the stack - a fill with the nearest copy's depth, ONE queue entry carrying the copies,
    // exactly what the levels below do for column-invariant parents.  prospero.vm at 1024^3: 64 root tiles instead of 512.
    R.root_zrep = is3d && S.pre_levels > 0 && ctx->use_split && TL == 64 && R.xy_fixed && R.root_invariant && (ctx->opt.no_zrep == 0 || ctx->opt.no_zrep == 3);
    // ... and of such a frame ONLY THE FRONT SLAB is rendered at all.  Nothing the frame evaluates depends on z: every tile, every leaf of
    // a slab further back repeats the front slab's result for its column with a smaller depth - a filled tile is filled in front of it, a
    // leaf's hits are the front leaf's hits, a pixel the front slab left empty is outside the model at every z - and the image takes the
    // largest depth.  The slabs behind the first (prospero.vm at 1024^3: half the root level's children - the linked prune then runs
    // its workgroups in one round instead of two -, one of two tile chains, one of two leaf launches) are not queued.  (no_zrep 3: every
    // slab, as before.)
    R.front_only = R.root_zrep && ctx->opt.no_zrep == 0;
    R.slab_stop = R.front_only && R.slab_hi > R.slab_lo ? R.slab_hi - 1 : R.slab_lo;
    uint32_t q0_layers = S.pre_levels ? layer_hi - layer_lo : 1;
    if (R.root_zrep) {
        q0_layers = 0;
        for (uint32_t sb = R.slab_hi; sb-- > R.slab_stop;) {       // front slabs first
            const uint32_t lo = std::max(layer_lo, sb * SL), hi = std::min(layer_hi, (sb + 1) * SL);
            if (lo >= hi) continue;
            q0_layers++;
            for (const Run& r : runs) {
                FhGroup g{};
                g.tape = root;
                g.first = r.first; g.n = r.n; g.stride = r.stride;
                g.z = lo * ts[0]; g.x = hi - lo;
                R.roots.push_back(g);
            }
        }
    } else
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

    { const fhip_status gs_ = grow_arena_if_asked(ctx, t.ops.size(), is3d && !R.root_invariant ? (uint64_t)P.width * P.height * P.depth : 0); if (gs_) return gs_; }
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
        // (three footprint lists and the normals kernel's list of leaves with a hit: at most every leaf of a slab)
        // (a footprint's pixels name at most one leaf per layer of the slab; footprint i of the class lists goes to bucket i % 64)
        R.hit_bucket_cap = (uint32_t)(((size_t)R.n_footprints + FH_HIT_BUCKETS - 1) / FH_HIT_BUCKETS * (P.slab / tl));
        R.hit_words = (size_t)FH_HIT_BUCKETS * (FH_HIT_STRIDE + R.hit_bucket_cap);
        HIP_TRY(ctx, ctx->fp_lists.ensure(((size_t)R.n_footprints * 3 + R.hit_words) * 4));
        HIP_TRY(ctx, ctx->fp_lists_b.ensure(extra * ((size_t)R.n_footprints * 3 + R.hit_words) * 4));
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
        S.hit_list = (uint32_t*)ctx->fp_lists.p + (size_t)3 * R.n_footprints;
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
    R.split = ctx->use_split && R.tl == 64 && true;
    // (tapes with sin cos tan asin acos atan exp ln: the *_t variants of the tile kernels, which carry those interval handlers;
    // and, since round 5, those for atan2, mod, mix, rand)
    R.asm_tiles_t = !tape_asm_ok(t) && tape_tiles_t_ok(t) && !ctx->opt.no_asm_tiles_t;
    // (not with a register file in HBM: the assembly tile kernels - fh_prune1, the groups path and the linked prune with them - keep
    // registers AND choices in LDS, and a tape of few registers can still outgrow it by its choices alone, ~5 600 of them)
    R.asm_tiles = R.split && ctx->use_asm && !ctx->opt.no_asm_tiles && (tape_asm_ok(t) || R.asm_tiles_t) && t.n_regs <= 128 && !R.big_hbm;
    R.asm_tiles_t = R.asm_tiles_t && R.asm_tiles;
    // (fhip_render_counters out[6]: frames whose tile stage took the HIP kernels although nobody switched the assembly ones off - left: tapes
    // of more than 128 registers and register files in HBM)
    if (!R.asm_tiles && R.split && ctx->use_asm && !ctx->opt.no_asm_tiles) ctx->hip_tile_frames++;
    // levels whose forward pass exports its choices to the one-wave-per-child prune (fh_prune1): long tapes, few parents.
    // 3D: of the pre-pass levels, level 0 (measured); 2D: level 0
    R.exp_levels = is3d ? std::min(S.pre_levels, 1u) : 1u;
    R.prune1 = R.asm_tiles && !R.asm_tiles_t && R.exp_levels > 0;      // (the *_t kernels have no export mode)
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
                    // the root chain's ops (plan.chain: acc = min / max(acc, term) all the way to the OUTPUT op) in evaluation order, behind the choice
                    // table: the linked prune's liveness pass starts from every kept op of the chain at once instead of walking it link by link
                    std::vector<uint32_t> chain = chain_table(tape, cops);
                    const size_t n_cops = std::max<size_t>(cops.size(), 1);
                    cops.resize(n_cops + (chain.size() + 1) / 2, 0);
                    if (!chain.empty()) memcpy(cops.data() + n_cops, chain.data(), chain.size() * 4);
                    hipError_t e = hipMalloc((void**)&dl, lk.size() * 8);
                    if (e == hipSuccess) e = hipMemcpy(dl, lk.data(), lk.size() * 8, hipMemcpyHostToDevice);
                    if (e == hipSuccess) e = hipMalloc((void**)&dc, std::max<size_t>(cops.size(), 1) * 8);
                    if (e == hipSuccess) e = hipMemcpy(dc, cops.data(), cops.size() * 8, hipMemcpyHostToDevice);
                    if (e != hipSuccess) {
                        if (dl) (void)hipFree(dl);
                        if (dc) (void)hipFree(dc);
                        HIP_TRY(ctx, e);
                    }
                    tape->d_links = dl; tape->d_ctab = dc; tape->n_chain = (uint32_t)chain.size();
                }
            }
            R.p2_cap_kept = (uint32_t)FH_P2_MAX_KEPT;
            R.lds_prune2 = (size_t)FH_P2_WPB * fh_p2_wave_lds(t.n_choices, R.p2_cap_kept) + (((size_t)tape->n_chain * 4 + 15) & ~(size_t)15);
            R.n_chain = tape->n_chain;
            // (one workgroup of FH_P2_WPB children per CU: beyond two rounds of them - 2048^3 has 4 096 root tiles - the scalar sweep,
            // whose waves all fit the machine at once, is the faster one again: 2.09 against 2.17 ms per frame)
            R.prune2 = tape->d_links && tape->d_ctab && ctx->opt.prune2 && t.ops.size() <= FH_P2_MAX_OPS && t.n_choices <= FH_P2_MAX_CHOICES &&
                       R.lds_prune2 <= FH_LDS_MAX && R.roots.size() * 64 <= (size_t)2 * ctx->n_cu * FH_P2_WPB;      // (a root group = up to 64 root tiles)
            R.d_ctab = tape->d_ctab;
            R.d_links = tape->d_links;
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
    S.want_stats = (ctx->profiling || ctx->probe || (ctx->opt.stats & 1)) ? 1 : 0;
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
    const unsigned blocks = (unsigned)std::max<size_t>(8, std::min<size_t>((size_t)ctx->n_cu * 32, (most + 64 * 64 - 1) / (64 * 64)));
    FH_KLAUNCH(k_frame_begin, dim3(blocks), dim3(WAVE), 0, ctx->stream, fb);
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
        if (R.tl == 64) FH_KLAUNCH((k_tiles<IS3D, FULL, BIG, 64>), dim3(grid), dim3(WAVE), lds, ctx->stream, dS, level); \
        else FH_KLAUNCH((k_tiles<IS3D, FULL, BIG, 16>), dim3(grid), dim3(WAVE), lds, ctx->stream, dS, level);            \
    } while (0)
// 3D tile stage of one level as three kernels (see kernels.hip "Split 3D tile stage")
// Rare mode: blocks per folded launch, and where a slab context's blocks keep their register files (a slab context = one FhRenderState of the set)
static const uint32_t FH_RARE_BLOCKS = 8;
static char* rare_file(fhip_ctx* ctx, FhRenderState* dS) {
    if (!ctx->rare_now) return nullptr;
    return (char*)ctx->rare_scratch.p + (size_t)(dS - (FhRenderState*)ctx->state.p) * FH_RARE_BLOCKS * ctx->rare_stride;
}
static void launch_tiles_split(fhip_ctx* ctx, const RenderSetup& R, FhRenderState* dS, int level, bool is3d) {
    // Persistent waves with a static round robin over the parents (one short workgroup per parent was measured slower in the
    // pipelined frame: the tile stage then takes more of the machine from the leaf kernel it overlaps with).
    const int gs = blocks_for(ctx, R.lds_tiles_small, 8);
    const int gb = blocks_big(ctx, R, R.lds_tiles_big, 8);
    const int gp = ctx->n_cu * 8;
    // Rare mode (render3d): at a per-slab level the two launches for parents outside the small slot list are not made; the push kernel's last
    // blocks evaluate such parents in C++ (none, nearly always)
    const bool rare_level = ctx->rare_now && is3d && R.asm_tiles && !ctx->opt.no_tiles_v && level > 0 && R.S.pre_levels > 0 && (uint32_t)level >= R.S.pre_levels &&
                            !(R.prune1 && (uint32_t)level < R.exp_levels);
    launch(ctx, FHIP_K_TILES, [&] {
        if (is3d) FH_KLAUNCH(k_tsetup3d, dim3(gp), dim3(WAVE), 0, ctx->stream, dS, level);
        else FH_KLAUNCH(k_tsetup2d, dim3(gp), dim3(WAVE), 0, ctx->stream, dS, level);
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
            if (R.S.top_chain && R.S.n_top > 64u * FH_CHAIN_SEG) FH_KLAUNCH(k_tchain3d_chunks, dim3(WAVE, blocks), dim3(WAVE), 0, ctx->stream, dS);
            else if (R.S.top_chain) FH_KLAUNCH(k_tchain3d, dim3(WAVE, blocks), dim3(WAVE), 0, ctx->stream, dS);
            else FH_KLAUNCH(k_ttop3d, dim3(blocks), dim3(WAVE), 0, ctx->stream, dS);
            // (the marks and the gather of the root tape's choice words do not depend on each other: one launch, the marks in the blocks
            // behind the gather's)
            if (R.classify_only || !root_words) FH_KLAUNCH(k_tmark3d, dim3(blocks), dim3(WAVE), 0, ctx->stream, dS);
            if (R.classify_only) return;       // (the fills of the decided tiles follow below; nothing is pruned or queued)
            if (root_words) FH_KLAUNCH(k_tscatter3d, dim3(root_words + 1, blocks), dim3(WAVE), 0, ctx->stream, dS, group_words, root_words);
            if (R.prune2) {
                hipEvent_t ea = nullptr, eb = nullptr;      // (timed under the fh_prune1 slot of the per-kernel profile: it replaces that launch)
                if (ctx->profiling) { (void)hipEventCreate(&ea); (void)hipEventCreate(&eb); (void)hipEventRecord(ea, ctx->stream); }
                FH_KLAUNCH(k_prune2, dim3(blocks * FH_P2_PER_SLOT), dim3(FH_P2_WPB * FH_P2_WPC * 64), R.lds_prune2, ctx->stream, dS, 0u, 1u, root_words,
                                   (const uint2*)R.d_links, (const uint2*)R.d_ctab, 2u, R.S.troot_len, R.S.troot_choices, R.p2_cap_kept,
                                   (const uint32_t*)(R.d_ctab + std::max<uint32_t>(R.S.troot_choices, 1)), R.n_chain);
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
                              ctx->stream != rest_stream && ctx->stream != ctx->stream_pre && is3d;
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
                both_lists = vk && (uint32_t)level < R.S.pre_levels && !side;
                if (vk && !both_lists) {
                    // (32 waves per compute unit: twice what fits at once - the waves take the slots round robin, and with as many waves as
                    // fit a launch of 1.5 slots per wave lasted two rounds of its longest parents: 0.213 -> 0.192 ms on the general path)
                    const int v32_waves = 32;
                    ka.n_waves = (uint32_t)(ctx->n_cu * v32_waves);
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
            if (rare_level) return;      // (the parents outside the small list: the blocks behind k_tpush3d's)
            // (a ROOT tape that fits fh_tiles_v64 - 64 registers, 512 choices - and is not pruned through exported choices takes it too:
            // bear.vm's 23 registers, 512^3: the root level 255 -> 160 us with the interval file in VGPRs instead of LDS)
            const bool root_v64 = vk && level == 0 && is3d && !R.groups && R.S.P.max_regs <= V64_REGS && R.S.P.max_choices <= V64_CHOICES;
            if (vk && (level > 0 || root_v64)) {
                const int v64_waves = 8;
                // (per-slab levels: the parents' tapes fit fh_tiles_v32 but for a rare one - an empty launch of 2048 waves of 176
                // VGPRs each, queued behind the leaf kernel of the slab in front, was measured to hold the tile chain up for
                // 130 us: a small persistent grid there)
                const int v64_slab_waves = 128;
                const bool per_slab = (uint32_t)level >= R.S.pre_levels && R.S.pre_levels > 0;
                ka.max_regs = V64_REGS; ka.max_choices = V64_CHOICES;
                ka.n_waves = per_slab ? (uint32_t)v64_slab_waves : (uint32_t)(ctx->n_cu * v64_waves);
                if (both_lists) ka.flags |= 16u;
                const uint32_t plain_flags = ka.flags;
                (void)launch_asm(ctx, R.asm_tiles_t ? FH_ASM_TILES_V64_T : FH_ASM_TILES_V64, ka.n_waves, &ka, sizeof(ka), 0, 1, big_stream);
                if (ctx->post_v64_stream && !per_slab && !big_stream) {      // (side_only_l1: the level's remaining launches - the LDS layouts' rest, the push - leave the side stream)
                    (void)hipEventRecord(ctx->ev_l1, ctx->stream);
                    (void)hipStreamWaitEvent(ctx->post_v64_stream, ctx->ev_l1, 0);
                    ctx->stream = ctx->post_v64_stream;
                }
                ka.flags = plain_flags & ~16u;
                ka.skip_regs = V64_REGS; ka.skip_choices = V64_CHOICES;
                rest = R.S.P.max_regs > V64_REGS || R.S.P.max_choices > V64_CHOICES;
            }
            // Pre-pass levels below the root: a few hundred parents whose tapes are far smaller than the
            // root's.  With the root-sized LDS layout only one wave fits a CU (256 at a time); a medium
            // layout takes those that fit it three to a CU, the root-sized launch takes the rest.
            const bool mid = !(vk && level > 0) && level > 0 && (uint32_t)level < R.S.pre_levels && R.lds_tiles_mid * 2 <= R.lds_tiles_big;
            if (mid) {
                const int gm = blocks_for(ctx, R.lds_tiles_mid, 8);
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
            if (R.full) FH_KLAUNCH((k_teval3d<true, false>), dim3(gs), dim3(WAVE), R.lds_tiles_small, ctx->stream, dS, level);
            else FH_KLAUNCH((k_teval3d<false, false>), dim3(gs), dim3(WAVE), R.lds_tiles_small, ctx->stream, dS, level);
        }
        if (R.full) FH_KLAUNCH((k_teval3d<true, true>), dim3(gb), dim3(WAVE), R.lds_tiles_big, ctx->stream, dS, level);
        else FH_KLAUNCH((k_teval3d<false, true>), dim3(gb), dim3(WAVE), R.lds_tiles_big, ctx->stream, dS, level);
    });
    // (last level: fewer waves, several parents each - one leaf reservation per wave)
#ifndef FH_PUSH_MUL
#define FH_PUSH_MUL 2
#endif
    const int push_mul = FH_PUSH_MUL;
    const int gpush = (level + 1 == (int)R.S.P.n_levels) ? ctx->n_cu * push_mul : gp;
    launch(ctx, FHIP_K_TILES, [&] {
        // (above the leaf level: 16 more waves per parent for the fills of its interval-full children - kernels.hip tfill3d_body)
        const uint32_t rb = rare_level ? FH_RARE_BLOCKS : 0u;
        if (is3d) FH_KLAUNCH(k_tpush3d, dim3(gpush + rb, (level + 1 == (int)R.S.P.n_levels) ? 1 : 17), dim3(WAVE), 0, ctx->stream, dS, level, rb, rare_file(ctx, dS), ctx->rare_stride);
        else {
            if (!R.classify_only) FH_KLAUNCH(k_tpush2d, dim3(gpush), dim3(WAVE), 0, ctx->stream, dS, level);
            const uint32_t slots_max = R.S.qcap[level] * ((level == 0 && R.groups) ? R.S.n_tgroups : 1u);
            FH_KLAUNCH(k_tfill2d, dim3(64, slots_max), dim3(256), 0, ctx->stream, dS, level);
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

static fhip_status render2d_frame(fhip_ctx* ctx, const fhip_tape* tape, const fhip_render2d_config* cfg, float* out,
                                  int out_is_device) {
    if (ctx->is_cancelled()) return fail(ctx, FHIP_ERR_CANCELLED, "cancelled");
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
    std::vector<uint32_t> ts = cfg->tile_sizes ? trim_tiles(cfg->tile_sizes, cfg->n_tile_sizes, std::max(cfg->width, cfg->height))
                                               : trim_tiles(HIP_TILES_2D, 2, std::max(cfg->width, cfg->height));
    // A 2D fill says at which level of the caller's list it was decided (pixel.rs:225-229), so the list is part of the result and is kept -
    // but a step of the list whose fan-out exceeds the 64 lanes of a wavefront (128 -> 8: 256 children) is taken in two: a level in between
    // that the caller does not see.  A tile decided there is what its children of the caller's next level would ALL have been decided as
    // (interval arithmetic is inclusion monotone), so its fill carries THEIR level; an undecided one hands them a tape pruned once more,
    // which changes no value.  The image is the one the caller's list gives, fill tags included.
    std::vector<uint32_t> tags;
    {
        std::vector<uint32_t> out;
        for (size_t i = 0; i < ts.size(); i++) {
            if (i) {
                uint32_t a = ts[i - 1];
                const uint32_t b = ts[i];
                if (a <= b || a % b) return fail(ctx, FHIP_ERR_UNSUPPORTED, "bad tile size list");
                while (a / b > 8) {       // the largest tile below `a` that takes at most 8 x 8 of `a` and is made of whole `b`s
                    uint32_t c = 0;
                    for (uint32_t k = 8; k >= 2 && !c; k--) if (a % k == 0 && (a / k) % b == 0 && a / k > b) c = a / k;
                    if (!c) break;        // (no such divisor: left to prepare(), which refuses the fan-out)
                    out.push_back(c); tags.push_back((uint32_t)i);
                    a = c;
                }
            }
            out.push_back(ts[i]); tags.push_back((uint32_t)i);
        }
        if (out.size() > FH_MAX_LEVELS) return fail(ctx, FHIP_ERR_UNSUPPORTED, "1..8 tile levels supported");
        ts = out;
    }
    const size_t npix = (size_t)cfg->width * cfg->height;
    float* d_out = out;
    if (!out_is_device) { HIP_TRY(ctx, ctx->tmp_out.ensure(npix * 4)); d_out = (float*)ctx->tmp_out.p; }
    // One pass over a tile list: the tile levels, then the leaf pixels (`classify_only`: the root level's fills, nothing else)
    auto pass = [&](RenderSetup& Q, const std::vector<uint32_t>& tiles, const std::vector<uint32_t>& tg) -> fhip_status {
        for (size_t i = 0; i < FH_MAX_LEVELS; i++) Q.S.P.tag[i] = i < tg.size() ? tg[i] : (uint32_t)i;
        fhip_status ps = prepare(ctx, tape, false, tiles, PartSpec{}, Q);
        if (ps) return ps;
        Q.S.image2d = d_out;
        FhRenderState* dS = (FhRenderState*)ctx->state.p;
        const FrameClear no_clear[3] = {};
        ps = upload_frame(ctx, tape, Q, no_clear);
        if (ps) return ps;
        for (uint32_t l = 0; l < Q.S.P.n_levels; l++) {
            if (ctx->is_cancelled()) return fail(ctx, FHIP_ERR_CANCELLED, "cancelled");
            launch_tiles(ctx, Q, dS, (int)l, false);
        }
        if (Q.classify_only) return FHIP_OK;
        launch(ctx, FHIP_K_POINTS, [&] {
            if (Q.full) FH_KLAUNCH((k_pixels2d<32, true>), dim3(ctx->n_cu * 8), dim3(WAVE), 0, ctx->stream, dS);
            else FH_KLAUNCH((k_pixels2d<32, false>), dim3(ctx->n_cu * 16), dim3(WAVE), 0, ctx->stream, dS);
        });
        if (Q.S.P.max_regs > 32)
            launch(ctx, FHIP_K_POINTS, [&] {
                const int g = blocks_big(ctx, Q, Q.lds_points_big, 16);
                if (Q.full) FH_KLAUNCH((k_pixels2d<0, true>), dim3(g), dim3(WAVE), Q.lds_points_big, ctx->stream, dS);
                else FH_KLAUNCH((k_pixels2d<0, false>), dim3(g), dim3(WAVE), Q.lds_points_big, ctx->stream, dS);
            });
        if (ctx->host_flags) FH_KLAUNCH(k_latch_arena, dim3(1), dim3(1), 0, ctx->stream, dS, 1u, ctx->host_flags);    // (an arena that ran out: grown before the next frame)
        return FHIP_OK;
    };
    // Small images of a large tape (round 5).  With 128 x 128 root tiles a 256 x 256 image is FOUR one-wave chains over a tape that a quarter
    // of the image barely prunes: 4.1 ms for prospero.vm, where the 4096 x 4096 image takes 0.5.  The root level's forward pass is parallel
    // over the tape whatever the number of tiles (term groups) and the linked prune takes a thousand children in a round, so the ROOT tape
    // is pruned per leaf tile directly - a one-level pass over the list's last entry - and the pixels are evaluated from those tapes;
    // values and decisions are the two-level recursion's (inclusion monotone intervals).  What the caller's list
    // still decides is the level a fill SAYS it was decided at (pixel.rs:225-229): a leaf tile's fill is tagged with the last level, and a
    // second, classify-only pass over the root tiles writes the fills of the decided ones - level 0 - over whatever their leaf tiles
    // wrote (a decided root tile's leaf tiles are all decided the same way, so nothing else is overwritten).
    const fh::HostTape& tt = tape->t;
    const bool small_ok = ts.size() == 2 && tags.size() == 2 && ts[1] >= 8 && ts[0] / ts[1] <= 8 && ctx->opt.root32_max > 0 && ctx->use_split && ctx->use_asm &&
                          !tape->tgroups.empty() && !ctx->opt.no_tape_groups && ctx->opt.prune2 && tape_asm_ok(tt) && tt.ops.size() <= FH_P2_MAX_OPS &&
                          tt.n_choices <= FH_P2_MAX_CHOICES &&
                          (uint64_t)((P.width + ts[1] - 1) / ts[1]) * ((P.height + ts[1] - 1) / ts[1]) <= 2048u &&
                          (uint64_t)((P.width + ts[0] - 1) / ts[0]) * ((P.height + ts[0] - 1) / ts[0]) <= 64u;
    if (small_ok) {
        RenderSetup A = R;
        A.one_level_64 = true;
        st = pass(A, std::vector<uint32_t>{ts[1]}, std::vector<uint32_t>{1u});
        if (st) return st;
        if (!P.pixel_perfect) {       // (pixel-perfect renders have no fills: pixel.rs:345-368)
            RenderSetup B = R;
            B.one_level_64 = true; B.classify_only = true;
            st = pass(B, std::vector<uint32_t>{ts[0]}, std::vector<uint32_t>{0u});
            if (st) return st;
        }
    } else {
        st = pass(R, ts, tags);
        if (st) return st;
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_done, ctx->stream));     // (a later pipelined 3D frame that takes this buffer set waits for it)
    ctx->ev_done_valid = true;
    if (!out_is_device) {
        HIP_TRY(ctx, hipMemcpyAsync(out, d_out, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
        return finish_render(ctx);
    }
    return FHIP_OK;
}

static fhip_status render3d_frame(fhip_ctx* ctx, const fhip_tape* tape, const fhip_render3d_config* cfg, void* out,
                                  int out_is_device, const PartSpec& part) {
    if (ctx->is_cancelled()) return fail(ctx, FHIP_ERR_CANCELLED, "cancelled");
    if (ctx->opt.stats & 2) g_spans.start();
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
    column_setup(ctx, tape, P, R);
    std::vector<uint32_t> ts = cfg->tile_sizes ? trim_tiles(cfg->tile_sizes, cfg->n_tile_sizes, std::max(cfg->width, cfg->height))
                                               : hip_tiles_3d(std::max(cfg->width, cfg->height));
    bool own_tiles = !cfg->tile_sizes;
    if (cfg->tile_sizes) {
        // Any list the reference accepts (TileSizes::new, fidget-core/src/render/mod.rs:181-251: descending, each a multiple of the next;
        // fidget-jit's own hint is [64, 16, 8], a caller's [64, 16, 4] is valid there) is accepted here: what the device's kernels cannot
        // take as given - leaves other than 8^3 (one 8 x 8 footprint per wavefront), a fan-out above 64 children (one per lane) - is
        // rendered with the library's list instead.  A 3D image does not depend on the tile sizes, so the caller
        // cannot tell, except by the time; fhip_render_counters out[7] counts such frames.
        bool valid = cfg->n_tile_sizes >= 1 && cfg->tile_sizes[cfg->n_tile_sizes - 1] >= 1, native = valid;
        for (uint32_t i = 1; i < cfg->n_tile_sizes && valid; i++)
            valid = cfg->tile_sizes[i - 1] > cfg->tile_sizes[i] && cfg->tile_sizes[i] > 0 && cfg->tile_sizes[i - 1] % cfg->tile_sizes[i] == 0;
        if (!valid) return fail(ctx, FHIP_ERR_UNSUPPORTED, "bad tile size list");
        native = ts.back() == 8 && ts.size() <= FH_MAX_LEVELS;
        for (size_t i = 1; i < ts.size() && native; i++) { const uint32_t n = ts[i - 1] / ts[i]; native = n * n * n <= 64; }
        if (!native) { ts = hip_tiles_3d(std::max(cfg->width, cfg->height)); own_tiles = true; ctx->substituted_tiles++; }
    }
    // Few tiles, long tape (a small image, a part of a frame on one rank of several, a model without z): root tiles of 32^3 straight
    // above the leaves.  With 128^3 root tiles such a frame is a handful of one-wave chains over tapes that a 128^3 tile barely prunes
    // (prospero.vm at 512^3: a root tile keeps up to 1 795 of 6 363 ops - beyond the linked prune's and fh_tiles_v64's limits, so the
    // LDS-file kernel and the scalar sweep walk them: 3.3 ms for one frame).  The root level's forward pass is parallel over the tape
    // (term groups) however many tiles there are, and the linked prune handles a thousand children in one round, each a wave: pruning
    // the ROOT tape per 32^3 tile costs what pruning it per 128^3 tile costs, its tapes are what level 1 would have arrived at, and
    // level 1 - the longest kernel of the frame - is not run at all: 512^3 3.25 -> 1.45 ms alone.  A 3D image does not depend on the
    // tile sizes, so this is the library's choice whenever the caller gave none: taken while the root level has at
    // most `root32_max` children - counting one layer per z-slab when the root tape reads nothing that changes along a pixel column
    // (root_zrep, prepare) - and the tape is one the groups + linked prune path takes.
    if (own_tiles && ctx->opt.root32_max > 0 && ts.size() == 3 && ts[0] == 128 && ctx->use_split && ctx->use_asm &&
        !tape->tgroups.empty() && !ctx->opt.no_tape_groups && ctx->opt.prune2 && tape_asm_ok(tape->t) &&
        tape->t.ops.size() <= FH_P2_MAX_OPS && tape->t.n_choices <= FH_P2_MAX_CHOICES) {
        const uint64_t cols = (uint64_t)((P.width + 31) / 32) * ((P.height + 31) / 32) / std::max<uint32_t>(1, part.n_shards * part.nx * part.ny);
        const bool dedupe = R.xy_fixed && R.root_invariant && (ctx->opt.no_zrep == 0 || ctx->opt.no_zrep == 3);
        const uint64_t layers = dedupe ? (ctx->opt.no_zrep == 0 ? 1u : (uint64_t)std::max<uint32_t>(2, (P.depth + 511) / 512))      // (one layer per slab; the front slab only)
                                       : (uint64_t)((P.depth + 31) / 32) / std::max<uint32_t>(1, part.nz);
        // (measured, profiles/r05c: up to two rounds of the linked prune's workgroups - 2 048 children - always; up to root32_max when a
        // 128^3 root tile is a quarter of the image or more - there the 128^3 tiles' tapes stay long whatever is done: 512^3 with z in
        // every tape, 4 096 children, 3.65 -> 1.72 ms; an octant of a 1024^3 frame, as many children of a model twice the size: 1.10 -> 1.33)
        const uint64_t children = cols * std::max<uint64_t>(layers, 1);
        if ((children <= 2048 || (children <= (uint64_t)ctx->opt.root32_max && std::max(P.width, P.height) <= 512)) && (P.depth + 31) / 32 <= FH_MAX_SLABS)
            ts = {32, 8};
    }
    // Frame pipelining (asynchronous renders): this frame takes the buffer set the previous frame did not use, and everything up
    // to and including its coarse levels is queued on a stream of its own - it depends on nothing the previous frame does, so it
    // runs beside that frame's slabs.  The slabs' tile chains follow on the side stream (after the previous frame's), the leaf
    // chains and the final image on the caller's stream as before.
    hipStream_t const main_stream = ctx->stream;
    // (a tape whose register files live in HBM takes the slow path: one region per workgroup, shared by the launches of a frame, so
    // nothing of the frame runs beside anything else)
    const bool huge = (size_t)std::max<uint32_t>(tape->t.n_regs, 1) * WAVE * 16 > FH_LDS_MAX || tiles_lds(std::max<uint32_t>(tape->t.n_regs, 1), tape->t.n_choices, 64) > FH_LDS_MAX;
    const bool fpipe = ctx->frame_pipeline && ctx->use_pipeline && !ctx->profiling && out_is_device && !huge;
    struct StreamGuard { fhip_ctx* c; hipStream_t s; ~StreamGuard() { c->stream = s; } } stream_guard{ctx, main_stream};
    bool frames_queued = false;     // the frame before this one is still under way (the caller queues frames back to back)
    bool lone = false;              // ... a pipelined context's frame with nothing under way before it
    if (fpipe) {
        // (rotate: the current set goes to the back of the ring, the set used longest ago comes forward)
        for (uint32_t i = 0; i < ctx->extra_sets; i++) std::swap(static_cast<FrameBufs&>(*ctx), ctx->others[i]);
        frames_queued = ctx->extra_sets > 0 && ctx->others[0].ev_done_valid && hipEventQuery(ctx->others[0].ev_done) == hipErrorNotReady;
        (void)hipGetLastError();
        // Two root levels side by side.  A frame of one coarse level whose tapes read no z (front slab only: one light tile chain, a leaf
        // stage of 5 k leaves) is its root level and little else: 165 us of kernels in one dependent chain on the pre-pass stream against
        // 100 on the side stream and 100 for lists + leaves + normals together - and that chain set the rate of queued frames.  Such frames
        // take the pre-pass stream and the tail stream IN TURN for their root level, and keep what the tail stream carried (lists, normals)
        // on the caller's stream around the leaf kernel: still four streams (a fifth shares a hardware queue with one of them and
        // serialises against it, measured in round 2), two frames' root levels in flight.
        R.alt_pre = ts.size() == 2 && R.xy_fixed && R.root_invariant && ctx->opt.no_zrep == 0 && ctx->stream3 &&
                    part.n_shards == 1 && part.nx * part.ny * part.nz == 1;
        // A frame ALONE - nothing of the frame before it is under way - keeps its coarse levels on the caller's stream: there is nothing to run
        // beside, and every change of stream is an event's latency (prospero.vm 1024^3, one frame alone: 0.388 -> 0.33 ms).  The frame queued
        // behind it takes the pre-pass stream as before and overlaps with it.
        lone = !frames_queued;
#ifdef FH_EXP_NO_LONE      // experiment (tools/build_lib_variant.py): every frame's coarse levels on the pre-pass stream, as until round 6
        lone = false;
#endif
        hipStream_t const pre_stream = lone ? main_stream : (R.alt_pre && (ctx->pre_turn++ & 1u) ? ctx->stream3 : ctx->stream_pre);
        ctx->stream = pre_stream;
        if (ctx->ev_done_valid) HIP_TRY(ctx, hipStreamWaitEvent(pre_stream, ctx->ev_done, 0));   // the set's previous frame has left it
    }
    FH_SPAN(0);
    st = prepare(ctx, tape, true, ts, part, R);
    if (st) return st;
    FH_SPAN(1);
    R.zrep = R.split && R.S.pre_levels > 0 && R.xy_fixed && !ctx->opt.no_column_inv && ctx->opt.no_zrep != 1;
    const size_t npix = (size_t)cfg->width * cfg->height;
    FhGeometryPixel* d_out = (FhGeometryPixel*)out;
    if (!out_is_device) { HIP_TRY(ctx, ctx->tmp_out.ensure(npix * sizeof(FhGeometryPixel))); d_out = (FhGeometryPixel*)ctx->tmp_out.p; }
    FhRenderState* dS = (FhRenderState*)ctx->state.p;
    // (FHIP_DEBUG_ZFILL, diagnostics: every pixel already at the far depth - the front slab's leaf kernel then finds all its
    // leaves but nothing pending, which times its per-workgroup and per-leaf set-up without the interpretation)
    const FrameClear clear3[3] = {{ctx->zbuf.p, npix * 8, 0u}, {ctx->normals.p, npix * 12, 0u},
                                  {ctx->mind.p, R.mind_words * 4, 0u}};
    st = upload_frame(ctx, tape, R, clear3);
    if (st) return st;
    FH_SPAN(2);
    const uint32_t n_groups = R.groups_per_slab;
    const uint32_t pre = R.S.pre_levels;
    const int reset_blocks = (int)std::max<uint32_t>(1, std::min<uint32_t>(1024, (std::max(R.table_words, n_groups) + 255) / 256));
    const int class_blocks = (int)((R.n_footprints + FH_CLASSIFY_FP - 1) / FH_CLASSIFY_FP);
    // Pipelined frames: the root level stays on the pre-pass stream, the level below it moves to the head of this frame's tile
    // chains on the side stream.  The two coarse levels of a frame are one dependent chain of ~0.9 ms that, on one stream, set
    // the frame rate; split, the root level of frame n + 1 runs beside level 1 and the slabs of frame n, and the side stream
    // carries level 1 + the (now few) slab steps of its own frame.  (A frame alone sees no difference: the same chain.)
    const bool l1_side = fpipe && !lone && pre > 1 && ctx->stream2 &&
                         ctx->use_pipeline && R.slab_hi - R.slab_lo > 1 && n_groups > 0;
    // (option side_only_l1, on: the side stream - the busiest one of a pipelined frame, 0.43 ms of the 0.526 - carries level 1's evaluate + prune
    // launches and nothing else: the flags of level 1's tapes are set at the end of the root level on the pre-pass stream, and what follows
    // level 1 - the flags of its children, the frame mark, the fork of the slab contexts - goes to the stream the tile chains run on)
    // (only for frames whose tile chains will run on the tail stream - `tiles_first` below, the same conditions: a frame with heavy leaf
    // kernels keeps its tile chains on the side stream, and its fork must not queue behind the previous frame's tail work)
    // Rare mode.  Four launches of a slab exist for tapes too large for the assembly kernels' register files - a leaf of more than 32
    // registers (its points, then its normals, in the C++ kernels with an LDS file), a parent of a per-slab tile level outside the small
    // slot list (fh_tiles_v64, then fh_tiles) - and find nothing to do in nearly every frame: prospero.vm 1024^3 0.138 -> 0.126 ms per
    // frame without them.  While the last finished frame of this context met no such tape (k_finish3d: host_flags[2]) the slab does not
    // make them: the last FH_RARE_BLOCKS blocks of k_classify3d, k_hits3d and k_tpush3d do their work - correct for any number of such
    // tapes, slow for many (a wave per block, the register files in HBM), and the first frame that meets one puts the launches back.
    {
        const size_t need = std::max(std::max(R.lds_tiles_big, R.lds_points_big), R.lds_normals_big);
        const size_t stride = (need + 255) / 256 * 256;
        // (a root tape of <= 32 registers has no large leaves, but its per-slab levels still launch fh_tiles_v64 for the other slot list)
        ctx->rare_now = R.split && R.asm_tiles && R.asm_points && R.asm_normals && !R.big_hbm && ctx->host_flags && ctx->host_flags[2] == 0 &&
                        stride * FH_RARE_BLOCKS * 4 <= ((size_t)256 << 20);
        if (ctx->rare_now) {
            HIP_TRY(ctx, ctx->rare_scratch.ensure(stride * FH_RARE_BLOCKS * 4));      // (a set's state buffer holds four slab contexts)
            ctx->rare_stride = (uint32_t)stride;
            ctx->rare_frames++;
        }
    }
    const bool rare = ctx->rare_now;
    bool l1_only = false;
    if (l1_side && pre == 2 && ctx->stream3 && R.asm_points) {
        const bool pipe_plan = ctx->use_pipeline && !ctx->profiling && R.slab_hi - R.slab_lo > 1 && n_groups > 0 && !R.big_hbm;
        const uint32_t nc_plan = pipe_plan ? std::min<uint32_t>(ctx->slab_contexts, R.slab_hi - R.slab_lo) : 1;
        l1_only = pipe_plan && R.root_invariant && R.slab_hi - R.slab_lo <= nc_plan;      // (column_setup: no tape of the frame reads an input that changes along a pixel column)
    }
    if (pre && n_groups) {  // coarse levels of every slab in one go
        for (uint32_t l = 0; l < pre; l++) {
            const bool flags_here = R.zrep && l > 0;
            if (l == 1 && l1_side) {
                if (flags_here && l1_only) launch(ctx, FHIP_K_OTHER, [&] { FH_KLAUNCH(k_tape_flags, dim3(ctx->n_cu * 4), dim3(WAVE), 0, ctx->stream, dS, (int)l, R.col_depmask, 0, FhFork{}); });
                HIP_TRY(ctx, hipEventRecord(ctx->ev_l0, ctx->stream_pre));
                HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_l0, 0));
                ctx->stream = ctx->stream2;
            }
            if (flags_here && !(l == 1 && l1_only)) launch(ctx, FHIP_K_OTHER, [&] { FH_KLAUNCH(k_tape_flags, dim3(ctx->n_cu * 4), dim3(WAVE), 0, ctx->stream, dS, (int)l, R.col_depmask, 0, FhFork{}); });
            ctx->post_v64_stream = (l == 1 && l1_only) ? ctx->stream3 : nullptr;
            launch_tiles(ctx, R, dS, (int)l, true);
            ctx->post_v64_stream = nullptr;
        }
        if (l1_only && ctx->stream != ctx->stream3) {      // (level 1 did not go through fh_tiles_v64: switch here)
            HIP_TRY(ctx, hipEventRecord(ctx->ev_l1, ctx->stream2));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream3, ctx->ev_l1, 0));
            ctx->stream = ctx->stream3;
        }
    }
    // Two-stream pipeline over the z-slabs: the tile stage of a slab runs on the side stream while
    // the leaves of the slab in front of it are evaluated on the caller's stream.  The occlusion
    // pyramid is then one slab stale, which is still exact (depths only grow).  Two slab contexts
    // (dS, dS + 1) alternate; each owns its leaves, leaf table, footprint lists and arena half.
    FhRenderState* const dS0 = dS;
    const bool pipe = ctx->use_pipeline && !ctx->profiling && R.slab_hi - R.slab_lo > 1 && n_groups > 0 && !R.big_hbm;
    hipStream_t const side_stream = ctx->stream2;
    const uint32_t NC = pipe ? std::min<uint32_t>(ctx->slab_contexts, R.slab_hi - R.slab_lo) : 1;     // (no more contexts than slabs: each takes its share of the arena)
    ctx->forked = pipe ? NC : 0;
    FhFork fork{};
    fork.n = NC; fork.mark = (pre && n_groups) ? 1u : 0u;
    fork.leaves = (FhLeaf*)ctx->leaves_b.p; fork.leaf_table = (FhLeafRef*)ctx->leaf_table_b.p; fork.fp_lists = (uint32_t*)ctx->fp_lists_b.p;
    fork.leaf_cap = (size_t)R.S.leaf_cap; fork.n_footprints = (size_t)R.n_footprints; fork.hit_words = R.hit_words;
    bool forked_here = false;
    if (pre && n_groups) {
        // (one coarse level - root tiles of 32^3 - in a pipelined frame: that level IS the frame's longest chain and the pre-pass stream the
        // pacemaker of the pipeline, so what follows its push - the flags of the parked parents, the frame mark, the fork of the slab
        // contexts - goes to the head of the tile chains on the side stream, which has no level 1 to carry in such a frame)
        if (fpipe && !lone && pipe && pre == 1 && side_stream && side_stream != ctx->stream) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_l0, ctx->stream));
            HIP_TRY(ctx, hipStreamWaitEvent(side_stream, ctx->ev_l0, 0));
            ctx->stream = side_stream;
        }
        if (R.zrep) launch(ctx, FHIP_K_OTHER, [&] {
            // (the fork of a pipelined frame's slab contexts - or, with ONE context, the frame mark alone - in the same launch)
            FH_KLAUNCH(k_tape_flags, dim3(ctx->n_cu * 8 + 1), dim3(WAVE), 0, ctx->stream, dS, (int)pre, R.col_depmask, 1, fork);
            forked_here = true;
        });
        if (!pipe && !forked_here) launch(ctx, FHIP_K_OTHER, [&] { FH_KLAUNCH(k_mark_frame, dim3(1), dim3(1), 0, ctx->stream, dS); });
    }
    if (pipe) {
        if (!forked_here) FH_KLAUNCH(k_fork_state, dim3(1), dim3(1), 0, ctx->stream, dS0, fork);
        HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
        if (ctx->stream != side_stream) HIP_TRY(ctx, hipStreamWaitEvent(side_stream, ctx->ev_fork, 0));
    }
    if (fpipe && ctx->stream != main_stream) {      // the rest of the frame is the caller's stream's (and the side stream's, which waits for the fork above)
        // (one coarse level, its tail on the side stream, and tile chains to follow there: the caller's stream waits for the first tile
        // chain's event, which lies behind everything queued so far - no event of its own for that)
        const bool implied = pipe && pre == 1 && n_groups > 0 && side_stream && ctx->stream == side_stream;
        if (!implied) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_pre, ctx->stream));     // (the stream the last coarse-level kernel went to)
            HIP_TRY(ctx, hipStreamWaitEvent(main_stream, ctx->ev_pre, 0));
        }
        ctx->stream = main_stream;
    }
    FH_SPAN(3);
    int last_tail_idx = -1;
    // Where a slab's tile chain goes: the side stream, or (option tiles_stream = 1, pipelined frames of at most as many slabs
    // as there are slab contexts) the tail stream, every slab's chain queued there BEFORE the tail work of the first slab - the
    // side stream then carries level 1 of the coarse levels alone, the pre-pass stream the root level, and the three chains
    // of consecutive frames run beside each other.
    // (2, the default: there when the ROOT tape reads no input that changes along a pixel column - then no tape of the frame does,
    // the leaf stage is light and the tail stream has room; a frame whose leaf kernels fill the machine wants its tile chains on
    // the high-priority side stream: prospero.vm 1024^3 0.77 -> 0.64 ms per frame there, the same frames with the column-invariance
    // short cuts off 1.86 -> 2.01)
    const bool root_invariant = R.root_invariant;
    const bool tiles_first = pipe && l1_side && root_invariant && ctx->stream3 && R.asm_points && R.slab_hi - R.slab_lo <= NC;
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
                              ((P.width + 31) / 32) * ((P.height + 31) / 32) <= 1024;
            const bool rebuild = k != (int)R.slab_hi - 1;  // the first slab sees an empty image (pyramid pre-zeroed)
            // (32 / 8 with its one pre-pass level: both pyramid levels rebuilt and the slab reset in one launch as well)
            const bool pyr2 = P.n_levels == 2 && P.tiles[1] == 8 && P.tiles[0] == 32 && pre == 1;
            if (rebuild && pyr2) {
                const uint32_t n0 = ((P.width + 31) / 32) * ((P.height + 31) / 32);
                FH_KLAUNCH(k_slab_begin2, dim3(n0 + reset_blocks), dim3(256), 0, ctx->stream, dS, n0, R.table_words, (uint32_t)k, n_groups);
                return;
            }
            if (rebuild && pyr3 && pre == 2) {
                const uint32_t n1 = ((P.width + 31) / 32) * ((P.height + 31) / 32);
                FH_KLAUNCH(k_slab_begin3, dim3(n1 + reset_blocks), dim3(256), 0, ctx->stream, dS, n1, R.table_words, (uint32_t)k, n_groups);
                return;
            }
            // (workgroups of one wave: they find room beside a leaf kernel that fills the machine - 73 us per launch on the general path with four)
            FH_KLAUNCH(k_reset_slab, dim3(reset_blocks * 4), dim3(WAVE), 0, ctx->stream, dS, R.table_words, (uint32_t)k, n_groups,
                               (pyr3 && rebuild) ? 1u : 0u);
            if (rebuild && pyr3) {
                const uint32_t n1 = ((P.width + 31) / 32) * ((P.height + 31) / 32);
                FH_KLAUNCH(k_minpyramid3, dim3(n1), dim3(256), 0, ctx->stream, dS);
            } else if (rebuild)
                FH_KLAUNCH(k_minpyramid, dim3(P.roots_x * P.roots_y), dim3(256), 0, ctx->stream, dS);
        });
        for (uint32_t l = pre; l < P.n_levels; l++) launch_tiles(ctx, R, dS, (int)l, true);
        if (pipe) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_tiles[idx], tile_stream));
            ctx->stream = main_stream;
        }
        return FHIP_OK;
    };
    if (tiles_first)
        for (int k = (int)R.slab_hi - 1; k >= (int)R.slab_stop && n_groups; k--) {
            const fhip_status ts_ = tile_step(k, (int)R.slab_hi - 1 - k);
            if (ts_) { ctx->stream = main_stream; return ts_; }
        }
    for (int k = (int)R.slab_hi - 1; k >= (int)R.slab_stop && n_groups; k--) {  // front to back (voxel.rs:252-261)
        if (ctx->is_cancelled()) { ctx->stream = main_stream; return fail(ctx, FHIP_ERR_CANCELLED, "cancelled"); }
        const int idx = (int)R.slab_hi - 1 - k;
        if (!tiles_first) {
            const fhip_status ts_ = tile_step(k, idx);
            if (ts_) { ctx->stream = main_stream; return ts_; }
        }
        dS = dS0 + (pipe ? (uint32_t)idx % NC : 0u);
        hipStream_t const leaf_stream = main_stream;
        if (pipe) HIP_TRY(ctx, hipStreamWaitEvent(leaf_stream, ctx->ev_tiles[idx], 0));
        // The leaf kernel is the slab's critical chain.  What surrounds it - the footprint lists (needed by the normals and the
        // LDS-class leaves only), those leaves (any order with the others: atomic-max z-buffer) and the normals of the slab's
        // hits - are small launches that leave the machine mostly idle, so in the pipelined frame they run on a third stream
        // beside the leaf kernel of the NEXT slab: the normals kernel only takes hits of its own slab's depth range, and a hit
        // behind them can never replace them.  (Measured with three slab contexts, ms per frame: everything on the caller's stream 2.44, the normals only on the third stream 2.30, lists + normals 2.16 - once the min-depth pyramid kernel of the tile chain ran in blocks of four waves: its 16-wave blocks found no room beside a leaf kernel that is never interrupted, 166 us instead of 10.  FHIP_TAIL_STREAM=0 / 2 / 1.)
        const int tail_mode = 1;   // (lists + normals on the tail stream; normals only - 2 - and off - 0 - were measured slower: DESIGN_HISTORY.md)
        // (option tail_on_main - 0 never, 1 always, 2 when frames are queued back to back: when the tile chains run on the tail stream, the
        // slab's small kernels stay on the caller's stream around its leaf kernel - otherwise the tail stream, serial, waits for every leaf
        // kernel with the NEXT frame's tile chains queued behind: 0.45 ms of it per frame for 0.40 of work.  A frame alone is 70 us
        // quicker with them beside its leaf kernels, hence the test)
        const bool on_main = tiles_first && frames_queued;
        const bool tail = pipe && ctx->stream3 && tail_mode > 0 && R.asm_points && !on_main && !R.alt_pre;   // (the HIP leaf kernels walk the footprint lists)
        const uint32_t z_lo = (uint32_t)k * P.slab, z_hi = z_lo + P.slab;
        auto classify_work = [&] {
            launch(ctx, FHIP_K_OTHER, [&] {
                FH_KLAUNCH(k_classify3d, dim3(class_blocks + (rare ? FH_RARE_BLOCKS : 0u)), dim3(P.slab / 8 > 16 ? 256 : WAVE), 0, ctx->stream, dS, R.asm_points ? 1 : 0, (uint32_t)class_blocks,
                           rare_file(ctx, dS), ctx->rare_stride);
            });
            if (P.max_regs > R.S.leaf_asm_regs && !rare)      // (rare mode: in the blocks behind k_classify3d's)
                launch(ctx, FHIP_K_POINTS, [&] {
                    const int g = blocks_big(ctx, R, R.lds_points_big, 16);
                    if (R.full) FH_KLAUNCH((k_leaves3d<2, 0, 1, true>), dim3(g), dim3(WAVE), R.lds_points_big, ctx->stream, dS);
                    else FH_KLAUNCH((k_leaves3d<2, 0, 1, false>), dim3(g), dim3(WAVE), R.lds_points_big, ctx->stream, dS);
                });
        };
        auto normals_work = [&] {
#ifdef FH_EXP_SKIP_NORMALS      // experiment (tools/build_lib_variant.py): a frame without its normals kernels - what they cost beside the leaf kernels
            return;
#endif
            launch(ctx, FHIP_K_NORMALS, [&] {
                const int gs = blocks_for(ctx, R.lds_normals_small, 8), gb = blocks_big(ctx, R, R.lds_normals_big, 8);
                if (R.asm_normals) {
                    // (lists 0 and 1 of k_classify3d hold every footprint whose leaves need <= 32 registers - the assembly interpreter's file:
                    // k_hits3d turns them into the list of leaves that own a hit, the normals kernel takes one leaf per wave pass)
                    const uint32_t hb = std::min<uint32_t>(R.n_footprints, (uint32_t)ctx->n_cu * 64);
                    FH_KLAUNCH(k_hits3d, dim3(hb + (rare ? FH_RARE_BLOCKS : 0u)), dim3(WAVE), 0, ctx->stream, dS, z_lo, z_hi, R.hit_bucket_cap, hb, rare_file(ctx, dS), ctx->rare_stride);
                    // (wave w walks bucket w % 64 with a stride of n_waves / 64)
                    struct { FhRenderState* S; uint32_t n_waves, slots, z_lo, z_hi, bucket_cap, pad; } kn = {dS, std::max<uint32_t>((uint32_t)(ctx->n_cu * 8) / FH_HIT_BUCKETS, 1u) * FH_HIT_BUCKETS, R.col_slots, z_lo, z_hi, R.hit_bucket_cap, 0};
                    (void)launch_asm(ctx, R.asm_points_t ? FH_ASM_NORMALS_T : FH_ASM_NORMALS, kn.n_waves, &kn, sizeof(kn));
                }
                else if (R.full) FH_KLAUNCH((k_normals3d<true, false>), dim3(gs), dim3(WAVE), R.lds_normals_small, ctx->stream, dS, z_lo, z_hi);
                else FH_KLAUNCH((k_normals3d<false, false>), dim3(gs), dim3(WAVE), R.lds_normals_small, ctx->stream, dS, z_lo, z_hi);
                if (P.max_regs > R.S.norm_asm_regs && !(rare && R.asm_normals)) {      // (rare mode: in the blocks behind k_hits3d's)
                    if (R.full) FH_KLAUNCH((k_normals3d<true, true>), dim3(gb), dim3(WAVE), R.lds_normals_big, ctx->stream, dS, z_lo, z_hi);
                    else FH_KLAUNCH((k_normals3d<false, true>), dim3(gb), dim3(WAVE), R.lds_normals_big, ctx->stream, dS, z_lo, z_hi);
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
                // per-frame constants of the leaf kernel (gen_interp.py gen_columns): input slots of the axes, the inputs that change
                // along a pixel column (a z coefficient in the axis' matrix row, or a projective matrix), projective flag
                const uint32_t blk = 4;   // footprints per workgroup: gen_interp.py FH_BLKL = 2
                const uint32_t n_blocks = (R.n_footprints + blk - 1) / blk;
                // Column walk (flags bit 20): one footprint COLUMN of the slab's leaf table per wave instead of one block of four footprints
                // of one layer.  A frame whose tapes read nothing that changes along a pixel column queues at most one leaf per column
                // and slab (the nearest of a stack), so its table is nearly empty - prospero 1024^3: 5.5 k leaves in 1 M entries - and
                // the (blocks, layers) grid is 262 144 workgroups of which 98 % load four empty entries and leave: 66 of the launch's
                // 71 us.  By columns it is 16 384 waves, each with its column's 64 entries in one load.  Leaves of a column are then
                // taken one after the other by one wave, front to back, which is wrong for frames with a leaf in most layers (the
                // launch would last as long as its fullest column: measured in round 3, bear.vm 2.40 -> 4.07 ms): option column_walk
                // 1 = only where the tapes guarantee sparse columns, 0 never, 2 always (tests).
                const bool by_columns = ctx->opt.column_walk == 2 || (ctx->opt.column_walk == 1 && R.xy_fixed && R.root_invariant && (ctx->opt.no_zrep == 0 || ctx->opt.no_zrep == 3));
                const uint32_t layers = P.slab / 8;
                // (the slab context's leaf table, as k_fork_state lays the contexts out: the kernel takes it - with the table's shape - from its
                // kernarg, so that a wave whose part of the table is empty leaves after one dependent load)
                const uint32_t sk = (uint32_t)(dS - dS0);
                const void* const slab_table = sk == 0 ? (const void*)R.S.leaf_table : (const void*)((const FhLeafRef*)ctx->leaf_table_b.p + (size_t)(sk - 1) * R.S.leaf_cap);
                // ... and for every other frame, round 6: the column walk by GROUPS of 2^g layers (flags bits 24 .. 27, grid y = the group,
                // front group first; option column_group = g, 0: the block walk below).  A wave keeps its footprint: the pixels' set-up,
                // their matrix products and their z-buffer words are loaded once per wave instead of once per leaf (the z-buffer words
                // were two thirds of the launch's HBM traffic), hits stay in registers from leaf to leaf and leave in one atomic.
                // (a blend - few min / max, nothing to prune: bear.vm's leaves keep 350 of the root's 650 ops - wants half the group: a wave's
                // leaves are taken one after the other, and the launch lasts as long as its fullest waves - 512^3, ms per frame by g = 0 / 1 /
                // 2 / 3: 1.24 / 0.97 / 1.09 / 1.33; prospero.vm's 22-op leaves on the general path: 0.433 / 0.424 ms per launch by g = 1 / 2)
                const uint32_t g_opt = (uint32_t)std::min(std::max(ctx->opt.column_group, 0), 6);
                const uint32_t g = by_columns ? 6u : (R.smooth_tape && g_opt > 1 ? g_opt - 1 : g_opt);
                if ((by_columns && layers <= 64) || (!by_columns && g > 0)) {
                    struct { FhRenderState* S; uint32_t n_waves, slots, depmask, flags, pad[2]; const void* table; uint32_t nfpl, layers; } ka =
                        {dS, 0u, R.col_slots, R.col_depmask, R.col_flags | (1u << 20) | (g << 24), {0, 0}, slab_table, R.n_footprints, layers};
                    const int which = R.asm_points_t ? FH_ASM_COLUMNS_T : FH_ASM_COLUMNS;
                    (void)launch_asm(ctx, which, (R.n_footprints + 63) / 64 * 64, &ka, sizeof(ka), 0, (layers + (1u << g) - 1) >> g, leaf_stream);
                    return;
                }
                // (pad[0]: floor(2^32 / blocks per layer) - the kernel rotates a layer's blocks by a per-layer offset, which is what balances
                // the launch, and takes the remainder by this reciprocal instead of a subtraction loop)
                struct { FhRenderState* S; uint32_t n_waves, slots, depmask, flags, pad[2]; const void* table; uint32_t nfpl, layers; } ka =
                    {dS, 0u, R.col_slots, R.col_depmask, R.col_flags, {n_blocks > 1 ? (uint32_t)(((uint64_t)1 << 32) / n_blocks) : 0u, 0}, slab_table, R.n_footprints, layers};
                const int which = R.asm_points_t ? FH_ASM_COLUMNS_T : FH_ASM_COLUMNS;
                (void)launch_asm(ctx, which, n_blocks, &ka, sizeof(ka), 0, P.slab / 8, leaf_stream);
            } else if (R.full) {
                FH_KLAUNCH((k_leaves3d<0, 16, 4, true>), dim3(ctx->n_cu * 8), dim3(WAVE), 0, ctx->stream, dS);
                FH_KLAUNCH((k_leaves3d<1, 32, 2, true>), dim3(ctx->n_cu * 8), dim3(WAVE), 0, ctx->stream, dS);
            } else {
                FH_KLAUNCH((k_leaves3d<0, 16, 4, false>), dim3(ctx->n_cu * 16), dim3(WAVE), 0, ctx->stream, dS);
                FH_KLAUNCH((k_leaves3d<1, 32, 2, false>), dim3(ctx->n_cu * 16), dim3(WAVE), 0, ctx->stream, dS);
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
    FH_SPAN(4);
    if (last_tail_idx >= 0) HIP_TRY(ctx, hipStreamWaitEvent(main_stream, ctx->ev_leaves[last_tail_idx], 0));   // the third stream is serial: the last slab's normals
    launch(ctx, FHIP_K_OTHER, [&] { FH_KLAUNCH(k_finish3d, dim3(ctx->n_cu * 16), dim3(WAVE), 0, ctx->stream, dS0, d_out, std::max<uint32_t>(ctx->forked, 1u), (uint32_t*)ctx->sticky.p, ctx->host_flags); });
    HIP_TRY(ctx, hipGetLastError());
    if (ctx->launch_failed) { ctx->launch_failed = false; return FHIP_ERR_HIP; }   // (message in fhip_last_error)
    ctx->async_pending = out_is_device != 0;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_done, main_stream));     // (a later pipelined frame that takes this set waits for it)
    ctx->ev_done_valid = true;
    if (ctx->opt.stats & 2) { g_spans.mark(5); g_spans.frame(); }
    if (!out_is_device) {
        HIP_TRY(ctx, hipMemcpyAsync(out, d_out, npix * sizeof(FhGeometryPixel), hipMemcpyDeviceToHost, ctx->stream));
        return finish_render(ctx);
    }
    return FHIP_OK;
}
// Frame lanes (option frame_lanes, default 4; 0 / 1: off).  The stage pipeline above runs the stages of consecutive frames beside
// each other, each stage on its stream; how far that goes is set by the busiest stream.  The other arrangement for frames that the
// caller queues back to back (the frame before is still under way): WHOLE frames beside each other - lane i % K is a child context
// that keeps to one stream (no_pipeline) and renders into an image of its own; the caller's stream waits for it and copies the image
// out, so `out` is only ever touched on the caller's stream, in order.  First measured with separate contexts driven in turn
// (tools/two_contexts.py under FHIP_NO_PIPELINE=1, profiles/r04r): bear.vm 512^3 - fh_columns_t with its 256 VGPRs, two waves per SIMD,
// 3.2 ms of kernels in a frame of 1.86 ms - 1.38 ms with three contexts; which arrangement a 3D frame takes is decided by lane_mode
// below.  A frame alone, a host output buffer or a profiled frame take the stage pipeline as before.  Parts of a frame (shards, blocks: what a rank of a
// multi-GPU job renders) are frames like any other here (option lanes_parts): one octant of prospero.vm 1024^3, queued, 0.72 -> 0.31 ms; a
// column shard of eight stays where it is, 0.43 (profiles/r04r/lanes_parts.txt).
static bool lanes_possible(fhip_ctx* ctx, int out_is_device) {
    if (ctx->opt.frame_lanes < 2 || ctx->is_lane || !out_is_device || ctx->profiling || ctx->probe || (ctx->opt.stats & 1)) return false;
    if (!ctx->ev_last_valid) return false;
    const hipError_t q = hipEventQuery(ctx->ev_last);      // the frame before this one: still under way?
    (void)hipGetLastError();
    return q == hipErrorNotReady;
}
static void lane_tune_release(fhip_ctx* ctx);
static void lanes_release(fhip_ctx* ctx, bool keep_measurements) {
    if (!keep_measurements) lane_tune_release(ctx);       // (what was measured was measured under the options of the moment)
    for (fhip_ctx* L : ctx->lanes) {
        hipStream_t const s = L->lane_stream_owned ? L->stream : nullptr;
        if (L->lane_done) (void)hipEventDestroy(L->lane_done);
        if (L->lane_copied) (void)hipEventDestroy(L->lane_copied);
        L->lane_img.release();
        fhip_ctx_destroy(L);
        if (s) (void)hipStreamDestroy(s);
    }
    ctx->lanes.clear();
}
// One frame on the next lane: render(lane, image) queues it on the lane's stream into the lane's own image of `bytes` bytes
static fhip_status run_on_lane_(fhip_ctx* ctx, uint32_t max_lanes, size_t bytes, void* out, const std::function<fhip_status(fhip_ctx*, void*)>& render) {
    const uint32_t K = std::min((uint32_t)std::min(ctx->opt.frame_lanes, 8), max_lanes);
    if (ctx->opt.lanes_fail) return fail(ctx, FHIP_ERR_HIP, "frame lanes: failure provoked (option lanes_fail)");
    while (ctx->lanes.size() < K) {
        // (the first three lanes ride on the streams of the stage pipeline, which is idle while the lanes run: the runtime shares a few
        // hardware queues - four by default - among all streams that have work, and lanes on streams of their own ended up two to a
        // queue behind the idle ones: bear.vm 512^3 1.55 ms per frame instead of 1.4)
        hipStream_t const mine[3] = {ctx->stream_pre, ctx->stream3, ctx->stream2};
        const size_t k = ctx->lanes.size();
        hipStream_t s = k < 3 ? mine[k] : nullptr;
        const bool owned = !s;
        if (owned) HIP_TRY(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        fhip_ctx* L = nullptr;
        const fhip_status st = fhip_ctx_create(ctx->device, (void*)s, &L);
        if (st) { if (owned) (void)hipStreamDestroy(s); return fail(ctx, st, "frame lanes: no child context"); }
        L->lane_stream_owned = owned;
        L->opt = ctx->opt;
        L->opt.no_pipeline = 1;
        L->opt.frame_lanes = 0;
        apply_options(L);
        L->is_lane = true;
        (void)hipEventCreateWithFlags(&L->lane_done, hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&L->lane_copied, hipEventDisableTiming);
        ctx->lanes.push_back(L);
    }
    fhip_ctx* const L = ctx->lanes[ctx->lane_next++ % K];
    L->cancel_src = &ctx->cancelled;
    L->watch.store(ctx->watch.load(std::memory_order_relaxed), std::memory_order_relaxed);
    HIP_TRY(ctx, L->lane_img.ensure(bytes));
    if (L->lane_copied_valid) HIP_TRY(ctx, hipStreamWaitEvent(L->stream, L->lane_copied, 0));   // (its previous image has been copied out)
    const fhip_status st = render(L, L->lane_img.p);
    if (st) { ctx->err = L->err; return st; }
    HIP_TRY(ctx, hipEventRecord(L->lane_done, L->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, L->lane_done, 0));
    HIP_TRY(ctx, hipMemcpyAsync(out, L->lane_img.p, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(L->lane_copied, ctx->stream));
    L->lane_copied_valid = true;
    ctx->lane_frames++;
    if (ctx->tune_cur < 0) ctx->lane_frames_wanted++;      // (not a frame of a measuring window: somebody's arrangement of choice)
    return FHIP_OK;
}
// (FHIP_LANE_FALLBACK: the lanes' own resources could not be had - device memory for a child context's buffers, say: the caller renders the
// frame under the stage pipeline instead, and the context gives its lanes up for good; option lanes_fail provokes it, for the test)
static const fhip_status FHIP_LANE_FALLBACK = (fhip_status)1000;
static fhip_status run_on_lane(fhip_ctx* ctx, uint32_t max_lanes, size_t bytes, void* out, const std::function<fhip_status(fhip_ctx*, void*)>& render) {
    ctx->last_hip_error = 0;
    for (fhip_ctx* L : ctx->lanes) if (L) L->last_hip_error = 0;
    const fhip_status st = run_on_lane_(ctx, max_lanes, bytes, out, render);
    if (st != FHIP_ERR_HIP) return st;
    // Only what says "the lanes' resources cannot be had" sends the frame back to the stage pipeline: out of memory (a child context's
    // buffers, its image), a child context or stream that could not be made, the provoked failure of the test.  Any other HIP error - a
    // fault inside the lane's render - is the caller's to see, with the lane's message.
    bool resources = ctx->opt.lanes_fail != 0 || ctx->last_hip_error == (int)hipErrorOutOfMemory || ctx->err.rfind("frame lanes:", 0) == 0;
    for (fhip_ctx* L : ctx->lanes) if (L && L->last_hip_error == (int)hipErrorOutOfMemory) resources = true;
    if (!resources) return st;
    if (ctx->mesh_leaves.cap) {       // (a mesh build's leaf records - gigabytes kept for the next build - may be what the lanes had no room beside)
        (void)hipDeviceSynchronize();
        ctx->mesh_leaves.release();
    }
    fprintf(stderr, "fidget-hip: frame lanes given up for this context (%s); frames keep to the stage pipeline\n", ctx->err.c_str());
    (void)hipGetLastError();
    (void)hipStreamSynchronize(ctx->stream);      // (nothing of the lanes may be pending on the caller's stream when they go)
    ctx->opt.frame_lanes = 0;
    lanes_release(ctx);
    ctx->err.clear();
    return FHIP_LANE_FALLBACK;
}
// Stage pipeline or lanes?  Measured with three lanes on the stage pipeline's streams (ms per queued frame, profiles/r04r/lanes_all.txt):
// prospero.vm 1024^3 0.505 / 0.570 (0.540 with four), with the column short cuts off 1.625 / 1.73 - but 512^3 1.66 / 1.15, 2048^3 2.18 / 1.99,
// colonnade.vm 1024^3 0.605 / 0.454, 512^3 0.334 / 0.248, bear.vm 512^3 1.84 / 1.37.  The stage pipeline wins where a frame's stages happen to be
// of equal length, which is a property of the model AND the size; nothing the host knows before the frame predicts it.  So it is measured:
// consecutive queued frames of one kind (tape, image size) run TUNE_SKIP + TUNE_WIN frames under the stage pipeline, as many on the lanes and as many
// under the stage pipeline again (a burst of frames starts on a machine whose clocks are still coming up, which counted against whatever
// was measured first: the general path's stage pipeline read 2.0 ms in the first window and runs at 1.63), each window timed between two
// events on the caller's stream after TUNE_SKIP frames of settling; the lanes are kept if they beat the better of the two stage windows
// by 3 %, for the life of the context (or until an option changes).  Frames that break the sequence - another kind, a frame alone - restart the
// window, so a queue of mixed frames never decides and keeps the prior: lanes for tapes with transcendental opcodes, the stage pipeline
// otherwise.  Both arrangements give the same image, bit for bit (tests/test_gpu_parity.py).
static constexpr uint32_t TUNE_SKIP = 4, TUNE_WIN = 8;       // (round 6: 40 frames to a verdict instead of 48)
// The lanes' window is followed by more frames on the lanes: K lanes finish their frames K at a time (heavy frames: every 5.5 ms four
// images of the general path of prospero.vm 1024^3), and a window whose last frame is the last frame ON the lanes ends with a group of one -
// a lone frame that takes half the time: such a window read 1.07 ms per frame where the lanes run at 1.38 and the stage pipeline at 1.34
// (profiles/r06b/experiments.txt 8).  With frames queued behind it the window's last group is a whole one.
// (with K lanes - option frame_lanes, 4 - the lanes' window is K frames of settling, 2 K frames timed, K frames behind them)
static uint32_t tune_lanes(const fhip_ctx* ctx) { return (uint32_t)std::min(std::max(ctx->opt.frame_lanes, 2), 8); }
static void lane_tune_release(fhip_ctx* ctx) {
    for (auto& t : ctx->lane_tune)
        for (hipEvent_t& e : t.ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    ctx->lane_tune.clear();
    ctx->tune_last_key = 0;
    ctx->tune_cur = -1;
}
static bool lane_mode(fhip_ctx* ctx, const fhip_tape* tape, const fhip_render3d_config* cfg, int out_is_device, const PartSpec& part) {
    ctx->tune_cur = -1;
    const bool whole = part.n_shards == 1 && part.nx * part.ny * part.nz == 1;
    const bool possible = lanes_possible(ctx, out_is_device) && ctx->use_pipeline && ctx->frame_pipeline;
    const bool prior = ctx->use_asm && !ctx->opt.no_columns_t && !tape_asm_ok(tape->t);
    if (!possible) { ctx->tune_last_key = 0; return false; }       // (a frame alone, a profiled frame, ...: the stage pipeline; the sequence is broken)
    if (!ctx->opt.lanes_tune) return prior;
    uint64_t key = tape->serial * 0x9E3779B97F4A7C15ull;
    key ^= ((uint64_t)cfg->width << 42) ^ ((uint64_t)cfg->height << 21) ^ (uint64_t)cfg->depth;
    key ^= (((uint64_t)part.shard * 64 + part.n_shards) * 0xD6E8FEB86659FD93ull) ^ ((((uint64_t)part.ix * 16 + part.iy) * 16 + part.iz) * 4096 + (part.nx * 16 + part.ny) * 16 + part.nz) * 0xA24BAED4963EE407ull;
    key |= 1;
    int at = -1;
    for (size_t i = 0; i < ctx->lane_tune.size(); i++) if (ctx->lane_tune[i].key == key) at = (int)i;
    if (at < 0) {
        if (ctx->lane_tune.size() < 32) { ctx->lane_tune.emplace_back(); at = (int)ctx->lane_tune.size() - 1; }
        else {      // the entry used longest ago makes room
            at = 0;
            for (size_t i = 1; i < ctx->lane_tune.size(); i++) if (ctx->lane_tune[i].used < ctx->lane_tune[at].used) at = (int)i;
            for (hipEvent_t& e : ctx->lane_tune[at].ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
            ctx->lane_tune[at] = fhip_ctx::LaneTune();
        }
        ctx->lane_tune[at].key = key;
    }
    fhip_ctx::LaneTune& T = ctx->lane_tune[at];
    T.used = ++ctx->tune_clock;
    if (T.phase == 4) { ctx->tune_last_key = key; return T.lanes; }
    if (ctx->tune_last_key != key) {       // the first frame of a run of this kind: the window starts again behind it, and it goes where the prior says
        ctx->tune_last_key = key;
        if (T.phase < 3) { T.n = 0; return prior; }
    }
    if (T.phase == 3) {
        const hipError_t q = hipEventQuery(T.ev[5]);
        (void)hipGetLastError();
        if (q != hipSuccess) return false;        // (under the stage pipeline until the last window's last event has passed)
        bool ok = true;
        for (int w = 0; w < 3; w++) {
            float t = 0.0f;
            ok = ok && hipEventElapsedTime(&t, T.ev[2 * w], T.ev[2 * w + 1]) == hipSuccess && t > 0.0f;
            T.ms[w] = t / (w == 1 ? 2 * tune_lanes(ctx) : TUNE_WIN);
        }
        (void)hipGetLastError();
        T.lanes = ok ? T.ms[1] < 0.97f * std::min(T.ms[0], T.ms[2]) : prior;
        T.phase = 4;
        // Memory by need: the child contexts exist since the measuring windows (a buffer set each, 0.5 GB).  When the verdict is the stage
        // pipeline, nothing else has ever used them and no other kind of frame is being measured, they go back - all windows' frames are
        // through (the last window's last event has passed, above), their images copied out before it.  A later frame that wants lanes
        // (a 2D queue, another kind's verdict) makes them again.
        if (!T.lanes && ctx->lane_frames_wanted == 0) {
            bool others = false;
            for (const fhip_ctx::LaneTune& o : ctx->lane_tune) others = others || (&o != &T && (o.phase != 4 || o.lanes));
            if (!others) lanes_release(ctx, true);
        }
        return T.lanes;
    }
    ctx->tune_cur = at;
    return T.phase == 1;
}
static fhip_status frame_queued(fhip_ctx* ctx, int out_is_device) {      // (the end of this frame, for the next one's lanes_possible)
    if (!out_is_device || ctx->is_lane || ctx->opt.frame_lanes < 2) return FHIP_OK;
    if (!ctx->ev_last) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_last, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_last, ctx->stream));
    ctx->ev_last_valid = true;
    if (ctx->tune_cur >= 0) {       // a frame of a measuring window: its marks
        fhip_ctx::LaneTune& T = ctx->lane_tune[(size_t)ctx->tune_cur];
        ctx->tune_cur = -1;
        T.n++;
        const uint32_t K = tune_lanes(ctx);
        const uint32_t skip = T.phase == 1 ? K : TUNE_SKIP, win = T.phase == 1 ? 2 * K : TUNE_WIN, tail = T.phase == 1 ? K : 0u;
        if (T.n == skip || T.n == skip + win) {
            hipEvent_t& e = T.ev[2 * T.phase + (T.n == skip ? 0 : 1)];
            if (!e) HIP_TRY(ctx, hipEventCreate(&e));
            HIP_TRY(ctx, hipEventRecord(e, ctx->stream));
        }
        if (T.n == skip + win + tail) { T.phase++; T.n = 0; }
    }
    return FHIP_OK;
}
static fhip_status render3d_part(fhip_ctx* ctx, const fhip_tape* tape, const fhip_render3d_config* cfg, void* out,
                                 int out_is_device, const PartSpec& part) {
    (void)hipSetDevice(ctx->device);
    fhip_status st = lane_mode(ctx, tape, cfg, out_is_device, part)
        ? run_on_lane(ctx, 8, (size_t)cfg->width * cfg->height * sizeof(FhGeometryPixel), out,
                      [&](fhip_ctx* L, void* img) { return render3d_frame(L, tape, cfg, img, 1, part); })
        : render3d_frame(ctx, tape, cfg, out, out_is_device, part);
    if (st == FHIP_LANE_FALLBACK) st = render3d_frame(ctx, tape, cfg, out, out_is_device, part);
    if (st) ctx->tune_cur = -1;
    return st ? st : frame_queued(ctx, out_is_device);
}
// (Four lanes by default, the fourth on a stream of its own: against three, prospero.vm 512^3 1.14 -> 0.91 ms per frame, colonnade.vm 1024^3 /
// 512^3 0.441 / 0.252 -> 0.413 / 0.212, bear.vm the same, gyroid-sphere 1024^3 1.14 -> 1.18; 2D frames keep to the three on the stage
// pipeline's streams: 4096^2 0.266 with three, 0.306 with four - profiles/r04r/lanes_3_or_4.txt.)
// 2D frames have no stage pipeline at all - a frame is one chain of tile levels and a pixel kernel on the caller's stream - so every
// queued 2D frame with a device output takes a lane: prospero.vm 4096^2 0.475 -> 0.333 ms per frame, 1024^2 0.83 -> 0.50 with three
// one-stream contexts in turn (profiles/r04r/frame_major3.txt)
fhip_status fhip_render2d(fhip_ctx* ctx, const fhip_tape* tape, const fhip_render2d_config* cfg, float* out, int out_is_device) {
    (void)hipSetDevice(ctx->device);
    ctx->tune_cur = -1;
    ctx->tune_last_key = 0;       // (a 2D frame between two 3D frames of one kind: their window starts again)
    fhip_status st = lanes_possible(ctx, out_is_device)
        ? run_on_lane(ctx, 3, (size_t)cfg->width * cfg->height * 4, out, [&](fhip_ctx* L, void* img) { return render2d_frame(L, tape, cfg, (float*)img, 1); })
        : render2d_frame(ctx, tape, cfg, out, out_is_device);
    if (st == FHIP_LANE_FALLBACK) st = render2d_frame(ctx, tape, cfg, out, out_is_device);
    return st ? st : frame_queued(ctx, out_is_device);
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
    FH_KLAUNCH(k_merge_depth, dim3((unsigned)std::min<uint64_t>((n_pixels + 255) / 256, 65535)), dim3(256), 0, ctx->stream,
                       (FhGeometryPixel*)front, (const FhGeometryPixel*)back, (size_t)n_pixels, image_depth);
    HIP_TRY(ctx, hipGetLastError());
    return FHIP_OK;
}
