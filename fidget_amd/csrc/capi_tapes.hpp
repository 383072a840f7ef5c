// Fragment of capi.hip (tapes: import, graph front end, simplify, register allocation as the reference's RegTape); not a stand-alone header: included by capi.hip only.
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
// ... the *_t tile kernels have an interval handler for every opcode (round 5 added atan2, modulo, rand and mix: gen_tiles.py b_atan2 ..
// b_mix), so every tape the assembly leaf kernels take, the assembly tile kernels take
static bool tape_tiles_t_ok(const fh::HostTape&) { return true; }
// The root chain of a tape for the linked prune's liveness pass (prune2.hip B1): when the root tree is a chain acc = min / max(acc, term)
// all the way to the OUTPUT op (plan.chain), its ops in evaluation order as choice ordinal | op index << 16 - `cops` is compute_links'
// per-choice table (word 1 = the op's index | class << 16).  Empty when the root is no chain.
static std::vector<uint32_t> chain_table(const fhip_tape* tape, const std::vector<uint64_t>& cops) {
    std::vector<uint32_t> chain;
    if (tape->plan.chain && tape->plan.top.size() < 65536) {
        chain.assign(tape->plan.top.size(), 0xFFFFFFFFu);
        for (size_t q = 0; q < tape->plan.choice_src.size() && q < cops.size(); q++)
            if ((tape->plan.choice_src[q] >> 24) == 255 && (tape->plan.choice_src[q] & 0xFFFFFFu) < chain.size())
                chain[tape->plan.choice_src[q] & 0xFFFFFFu] = (uint32_t)q | ((uint32_t)((cops[q] >> 32) & 0xFFFFu) << 16);
        for (uint32_t c : chain) if (c == 0xFFFFFFFFu) { chain.clear(); break; }
    }
    return chain;
}

// What kinds of opcodes a tape holds: one walk per tape, remembered in the tape (a frame's set-up asked five times, 4 us each for
// prospero's 6 363 ops, when the host thread had become the pacemaker of queued frames).  bit 0: looked up; 1: transcendental / rng /
// atan2 / mix / modulo opcodes (the *_t kernels' handlers); 2: a modulo; 3: transcendental / atan2 / modulo (the C++ kernels' FULL variants)
static uint32_t tape_class(const fh::HostTape& t) {
    uint32_t c = __atomic_load_n(&t.op_class, __ATOMIC_ACQUIRE);
    if (c) return c;
    c = 1;
    for (uint64_t w : t.ops) {
        const uint32_t op = FH_W_OP((uint32_t)w);
        if ((op >= FH_SIN && op <= FH_LN) || op == FH_RAND) c |= 2;
        if (op >= FH_SIN && op <= FH_LN) c |= 8;
        if (op >= FH_ADD_RR) {
            const int base = op >= FH_SUB_IR ? (int[]){1, 3, 4, 5, 6, 7}[op - FH_SUB_IR] : (int)((op - FH_ADD_RR) % 12);
            if (base == 4 || base == 6 || base == 7) c |= 2;  // atan2, mix, mod
            if (base == 4 || base == 7) c |= 8;
            if (base == 7) c |= 4;
        }
    }
    __atomic_store_n(&t.op_class, c, __ATOMIC_RELEASE);
    return c;
}
static bool tape_has_mod(const fh::HostTape& t) { return (tape_class(t) & 4) != 0; }
// The assembly interpreters implement every opcode except the transcendental, modulo and rng ones
static bool tape_asm_ok(const fh::HostTape& t) { return (tape_class(t) & 2) == 0; }
static bool tape_is_full(const fh::HostTape& t) { return (tape_class(t) & 8) != 0; }

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
    fhip_tape* t = new fhip_tape();
    if (!simplify_host(p, choices, t->t)) { delete t; return fail(ctx, FHIP_ERR_BAD_CHOICE_SLICE, "Choice::Unknown in trace"); }
    *child = t;
    return FHIP_OK;
}
