// Fragment of capi.hip (contexts and their options); not a stand-alone header: included by capi.hip only.
// ---- the host's libm against the routines the device restates (include/fidget_hip.h) -------------------------------------------------
int fhip_libm_probe(char* msg, size_t cap) {
    using namespace fhlm;
    struct R1 { const char* name; float (*mine)(float); float (*host)(float); };
    const R1 r1[] = {{"sinf", [](float x) { return sincosf_<MemTables, false>(x); }, [](float x) { return ::sinf(x); }},
                     {"cosf", [](float x) { return sincosf_<MemTables, true>(x); }, [](float x) { return ::cosf(x); }},
                     {"tanf", [](float x) { return tanf_<MemTables>(x); }, [](float x) { return ::tanf(x); }},
                     {"asinf", [](float x) { return asinf_(x); }, [](float x) { return ::asinf(x); }},
                     {"acosf", [](float x) { return acosf_(x); }, [](float x) { return ::acosf(x); }},
                     {"atanf", [](float x) { return atanf_(x); }, [](float x) { return ::atanf(x); }},
                     {"expf", [](float x) { return expf_<MemTables>(x); }, [](float x) { return ::expf(x); }},
                     {"logf", [](float x) { return logf_<MemTables>(x); }, [](float x) { return ::logf(x); }}};
    // 32 arguments: ordinary values of both signs, points around the reductions' boundaries (pi/4 multiples, 120, 2^-12, 1), tiny / huge
    static const float args[32] = {0.1f, -0.3f, 0.5f, 0.7853982f, 0.7853981f, 1.0f, -1.0f, 1.5707964f, 2.0f, 3.0f, 3.1415927f, -4.5f, 6.2831855f, 10.0f,
                                   -25.132742f, 100.0f, 119.99999f, 120.0f, 1000.5f, -31415.926f, 1.0e6f, 3.0e8f, 1.0e-3f, 2.4414062e-4f, -1.0e-5f,
                                   1.0e-20f, 0.9999999f, 0.99f, -0.6f, 0.25f, 87.0f, -80.0f};
    int bad = 0;
    if (msg && cap) msg[0] = 0;
    auto differ = [](float a, float b) { return !(a != a && b != b) && f2u(a) != f2u(b); };
    for (const R1& r : r1)
        for (float x : args) {
            volatile float xv = x;          // (no constant folding of the host's call: the RUNNING libm is what is asked)
            const float a = r.mine(xv), b = r.host(xv);
            if (differ(a, b) && bad++ == 0 && msg && cap)
                snprintf(msg, cap, "%s(%a): device family 0x%08x, host libm 0x%08x", r.name, (double)x, f2u(a), f2u(b));
        }
    for (int i = 0; i < 32; i++) {
        volatile float y = args[i], x = args[(i * 7 + 3) & 31];
        const float a = atan2f_(y, x), b = ::atan2f(y, x);
        if (differ(a, b) && bad++ == 0 && msg && cap)
            snprintf(msg, cap, "atan2f(%a, %a): device family 0x%08x, host libm 0x%08x", (double)y, (double)x, f2u(a), f2u(b));
    }
    return bad;
}
static void libm_probe_once() {
    static std::once_flag once;
    std::call_once(once, [] {
        char msg[160];
        const int bad = fhip_libm_probe(msg, sizeof msg);
        const char* q = getenv("FHIP_QUIET");
        if (bad && !(q && q[0] == '1'))
            fprintf(stderr, "fidget-hip: this host's libm is not the one the device restates (glibc 2.35 x86-64, FMA variants; trans_libm.hpp): %d of 288 "
                            "probe arguments differ, first %s.  Transcendental opcodes on the device will differ from this host's CPU evaluators "
                            "in the last bits for some arguments.\n", bad, msg);
    });
}

// ---- context ---------------------------------------------------------------------------
fhip_status fhip_ctx_create(int device, void* stream, fhip_ctx** out) {
    if (!out) return FHIP_ERR_BAD_TAPE;
    *out = nullptr;
    libm_probe_once();
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
    {   // diagnostics (A/B runs of assembly variants, tools/variants.py): FHIP_INTERP_CO=<file> replaces the embedded code object
        std::vector<char> alt;
        if (const char* p = getenv("FHIP_INTERP_CO")) {
            if (FILE* f = fopen(p, "rb")) {
                fseek(f, 0, SEEK_END);
                long n = ftell(f);
                fseek(f, 0, SEEK_SET);
                alt.resize(n > 0 ? (size_t)n : 0);
                if (n <= 0 || fread(alt.data(), 1, (size_t)n, f) != (size_t)n) alt.clear();
                fclose(f);
            }
            if (alt.empty()) { fprintf(stderr, "fidget-hip: FHIP_INTERP_CO=%s cannot be read\n", p); delete c; return FHIP_ERR_HIP; }
            fprintf(stderr, "fidget-hip: assembly kernels from %s\n", p);
        }
        if (hipModuleLoadData(&c->asm_mod, alt.empty() ? (const void*)fh_interp_co : (const void*)alt.data()) != hipSuccess) { delete c; return FHIP_ERR_HIP; }
    }
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
    (void)hipEventCreateWithFlags(&c->ev_l1, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&c->ev_rest_fork, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&c->ev_rest_join, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming);
    for (auto& o : c->others) (void)hipEventCreateWithFlags(&o.ev_done, hipEventDisableTiming);
    if (!c->stream3 && hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking) != hipSuccess) { delete c; return FHIP_ERR_HIP; }
    if (hipStreamCreateWithFlags(&c->stream_pre, hipStreamNonBlocking) != hipSuccess) { delete c; return FHIP_ERR_HIP; }
    if (c->sticky.ensure(256) != hipSuccess || hipMemset(c->sticky.p, 0, 256) != hipSuccess) { delete c; return FHIP_ERR_HIP; }
    {
        void* hf = nullptr;
        if (hipHostMalloc(&hf, 64, hipHostMallocDefault) != hipSuccess) { delete c; return FHIP_ERR_HIP; }
        memset(hf, 0, 64);
        ((uint32_t*)hf)[2] = 1;      // (no frame has finished yet: the kernels for the large tapes are launched on their own - capi_render.hpp rare mode)
        c->host_flags = (volatile uint32_t*)hf;
    }
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
static void lanes_release(fhip_ctx* ctx, bool keep_measurements = false);      // (capi_render.hpp)
void fhip_ctx_destroy(fhip_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    lanes_release(c);
    if (c->ev_last) (void)hipEventDestroy(c->ev_last);
    if (c->stream_pre) (void)hipStreamSynchronize(c->stream_pre);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    (void)hipStreamSynchronize(c->stream);
    DevBuf* bufs[] = {&c->tmp_out, &c->io_a, &c->io_b, &c->io_c, &c->io_d, &c->io_e, &c->sticky};
    for (DevBuf* b : bufs) b->release();
    c->release_all();
    for (auto& o : c->others) o.release_all();
    if (c->stream_pre) (void)hipStreamDestroy(c->stream_pre);
    if (c->ev_rest_fork) (void)hipEventDestroy(c->ev_rest_fork);
    if (c->ev_rest_join) (void)hipEventDestroy(c->ev_rest_join);
    if (c->ev_pre) (void)hipEventDestroy(c->ev_pre);
    if (c->ev_l0) (void)hipEventDestroy(c->ev_l0);
    if (c->ev_l1) (void)hipEventDestroy(c->ev_l1);
    for (auto& sg : c->staging) { if (sg.p) (void)hipHostFree(sg.p); if (sg.ev) (void)hipEventDestroy(sg.ev); }
    c->mesh_leaves.release();
    if (c->host_flags) (void)hipHostFree((void*)c->host_flags);
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
    for (fhip_ctx* L : c->lanes) {      // frames that went to the lanes: each lane's own checks
        const fhip_status s3 = fhip_ctx_sync(L);
        if (s3 && st == FHIP_OK) { st = s3; c->err = L->err; }
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
// (the frame lanes are contexts of their own whose per-level checks read THIS flag through fhip_ctx::cancel_src: a cancel from another
// thread reaches a frame a lane is queueing without touching the lanes' vector, which belongs to the render thread)
void fhip_cancel(fhip_ctx* c) { c->cancelled.store(1); }
void fhip_cancel_reset(fhip_ctx* c) { c->cancelled.store(0); }
void fhip_cancel_watch(fhip_ctx* c, const void* flag) { c->watch.store((const volatile unsigned char*)flag); }
// Device and pinned memory the context keeps between calls for speed alone - the mesher's leaf records (17 GB after one depth-10 build),
// its landing area and host-side caches, the frame lanes (child contexts with buffers of their own) - given back.  The next call that
// wants them makes them again.  Waits for the context's work first.
fhip_status fhip_ctx_trim(fhip_ctx* c) {
    if (!c) return FHIP_ERR_UNSUPPORTED;
    (void)hipSetDevice(c->device);
    HIP_TRY(c, hipDeviceSynchronize());
    lanes_release(c);
    c->mesh_leaves.release();
    if (c->mesh_pinned) { (void)hipHostFree(c->mesh_pinned); c->mesh_pinned = nullptr; c->mesh_pinned_cap = 0; }
    return FHIP_OK;
}
static hipError_t sync_own_streams(fhip_ctx* ctx);      // (capi_render.hpp)
fhip_status fhip_ctx_reserve_arena(fhip_ctx* c, size_t megabytes) {
    if (!c) return FHIP_ERR_UNSUPPORTED;
    (void)hipSetDevice(c->device);
    const size_t want = std::min(c->arena_cap_bytes, megabytes << 20);
    if (want <= c->arena_bytes) return FHIP_OK;
    HIP_TRY(c, sync_own_streams(c));        // (the sets' arenas are replaced as the frames take them in turn)
    c->arena_bytes = want;
    return FHIP_OK;
}
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
            lanes_release(c);      // (the lanes carry a copy of the options: made again, with the new ones, by the next frame that wants them)
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
