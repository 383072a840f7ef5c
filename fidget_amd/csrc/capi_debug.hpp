// Fragment of capi.hip (profiling, diagnostics, host graph entry points); not a stand-alone header: included by capi.hip only.
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
    for (int i = 0; i < 2; i++) out[4 + i] = S.count[i + 1];
    out[6] = ctx->hip_tile_frames;        // (frames whose tile stage took the HIP kernels implicitly: tapes beyond the assembly kernels' register files)
    out[7] = ctx->substituted_tiles;      // (3D frames rendered with the library's tile list in place of a caller's the kernels cannot take)
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
// ... and its root chain as the linked prune's liveness pass gets it (capi_tapes.hpp chain_table): choice ordinal | op index << 16 per chain
// op, evaluation order; 0: the root is no chain (or the tape does not qualify)
uint32_t fhip_debug_tape_chain(const fhip_tape* tape, uint32_t* out, uint32_t cap) {
    std::vector<uint64_t> lk;
    std::vector<uint64_t> cops;
    if (!fh::compute_links(tape->t, lk, cops)) return 0;
    const std::vector<uint32_t> chain = chain_table(tape, cops);
    for (size_t i = 0; i < chain.size() && i < cap; i++) out[i] = chain[i];
    return (uint32_t)chain.size();
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

// Diagnostics: the embedded copy `copy` (order of gen_trans.COPIES) of compiled routine `fn` (index into gen_trans.FUNCS + FUNCS4)
// over the n floats with bit patterns first .. first + n - 1 (n a multiple of 256), against the routine as the HIP kernels inline
// it: out = {results whose bits differ, an input where they do}; a (copy, fn) that does not exist reports every result
fhip_status fhip_debug_trans_probe(fhip_ctx* ctx, uint32_t copy, uint32_t fn, uint32_t first, uint64_t n, uint64_t out[2]) {
    (void)hipSetDevice(ctx->device);
    if (!n || n % 256 || n > (1ull << 28)) return fail(ctx, FHIP_ERR_UNSUPPORTED, "trans probe: n must be a multiple of 256, at most 2^28");
    if (!ctx->asm_fn[FH_ASM_TRANS_PROBE]) return fail(ctx, FHIP_ERR_UNSUPPORTED, "trans probe: the assembly kernels are not loaded");
    HIP_TRY(ctx, ctx->io_a.ensure(n * 4));
    HIP_TRY(ctx, ctx->io_b.ensure(64));
    HIP_TRY(ctx, hipMemsetAsync(ctx->io_a.p, 0x5A, n * 4, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->io_b.p, 0, 64, ctx->stream));
    struct { void* out; uint32_t first, copy, fn, pad; } ka = {ctx->io_a.p, first, copy, fn, 0};
    if (launch_asm(ctx, FH_ASM_TRANS_PROBE, (uint32_t)(n / 256), &ka, sizeof(ka)) != hipSuccess) return FHIP_ERR_HIP;
    hipLaunchKernelGGL(k_trans_compare, dim3(ctx->n_cu * 16), dim3(256), 0, ctx->stream, (int)fn, first, (size_t)n, (const uint32_t*)ctx->io_a.p,
                       (unsigned long long*)ctx->io_b.p);
    HIP_TRY(ctx, hipGetLastError());
    unsigned long long r[2];
    HIP_TRY(ctx, hipMemcpyAsync(r, ctx->io_b.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    out[0] = r[1]; out[1] = r[0] & 0xFFFFFFFFull;
    return FHIP_OK;
}

// Diagnostics: how many 3D frames of this context (its lanes included) were rendered in rare mode (capi_render.hpp render3d) so far
uint64_t fhip_debug_rare_frames(const fhip_ctx* ctx) {
    if (!ctx) return 0;
    uint64_t n = ctx->rare_frames;
    for (const fhip_ctx* l : ctx->lanes) n += l ? l->rare_frames : 0;
    return n;
}
// Diagnostics: how many frames of this context went to a frame lane (capi_render.hpp run_on_lane) so far
uint64_t fhip_debug_lane_frames(const fhip_ctx* ctx) { return ctx ? ctx->lane_frames : 0; }
// ... and what the arrangement tuner (capi_render.hpp lane_mode) knows about the kind of 3D frame queued last: its phase (0 / 1 / 2 measuring the
// stage pipeline, the lanes, the stage pipeline again; 3 waiting; 4 decided; -1: no such frame yet, or the last frame broke the sequence),
// ms[0 .. 2] = ms per frame of the three windows, *lanes = the decision
int fhip_debug_lane_tune(const fhip_ctx* ctx, float ms[3], int* lanes) {
    if (!ctx) return -1;
    for (const auto& t : ctx->lane_tune)
        if (t.key == ctx->tune_last_key) {
            if (ms) { ms[0] = t.ms[0]; ms[1] = t.ms[1]; ms[2] = t.ms[2]; }
            if (lanes) *lanes = t.lanes ? 1 : 0;
            return t.phase;
        }
    return -1;
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
