// Fragment of capi.hip (trait-level evaluators (point, interval, bulk, gradient) and geometry helpers); not a stand-alone header: included by capi.hip only.
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
