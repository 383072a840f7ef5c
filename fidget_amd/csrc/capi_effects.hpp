// Fragment of capi.hip (image effects); not a stand-alone header: included by capi.hip only.
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
