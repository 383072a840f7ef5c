// fidget-hip: post-processing of rendered images on the device (fidget-raster/src/effects.rs), the step right
// after the render: the GeometryPixel / RawDistancePixel image is already resident in HBM.  One thread per pixel;
// every kernel is a pure gather over a few neighbours: HBM-bound, a handful of microseconds at 1024^2.
// f32 op order follows the reference (and nalgebra's small-vector code: dot = (a0*b0 + a1*b1) + a2*b2,
// norm = sqrt(0 + dot), normalize = v / norm, M * v by columns); built with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

#include "dev_ops.hpp"
#include "render_state.h"

namespace fhfx {
using fhd::pcg;

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 mul(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 divs(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 normalize(V3 a) { return divs(a, sqrtf(0.0f + dot(a, a))); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// ordered_float::OrderedFloat: NaN above everything, equal to itself
__device__ __forceinline__ int of_cmp(float a, float b) {
    const bool na = a != a, nb = b != b;
    if (na || nb) return na == nb ? 0 : (na ? 1 : -1);
    return a < b ? -1 : (a > b ? 1 : 0);
}
__device__ __forceinline__ uint8_t to_u8(float v) { return !(v > 0.0f) ? 0 : (v >= 255.0f ? 255 : (uint8_t)v); }   // `as u8`

// effects.rs:17-36 denoise_normals + 252-326 denoise_pixel (radius 2)
__global__ void k_fx_denoise(const FhGeometryPixel* __restrict__ img, int W, int H, FhGeometryPixel* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    const FhGeometryPixel p = img[(size_t)y * W + x];
    FhGeometryPixel o;
    o.depth = p.depth;
    o.normal[0] = o.normal[1] = o.normal[2] = 0.0f;
    if (p.depth > 0) {
        V3 best = v3(p.normal[0], p.normal[1], p.normal[2]);
        if (!(p.normal[2] > 0.0f)) {
            const int r = 2;
            bool have = false;
            float best_score = 0.0f;
            for (int w = 0; w < 4; w++) {
                const int xm = (w & 1) ? -r : 0, ym = (w & 2) ? -r : 0;
                V3 sum = v3(0.0f, 0.0f, 0.0f);
                int count = 0;
                for (int i = 0; i <= r; i++)
                    for (int j = 0; j <= r; j++) {
                        const int tx = x + xm + i, ty = y + ym + j;
                        if (tx < 0 || ty < 0 || tx >= W || ty >= H) continue;
                        const FhGeometryPixel q = img[(size_t)ty * W + tx];
                        if (q.depth != 0 && q.normal[2] > 0.0f) { sum = add(sum, v3(q.normal[0], q.normal[1], q.normal[2])); count++; }
                    }
                if (count == 0) continue;
                const V3 mean = divs(sum, (float)count);
                float score = 0.0f;
                for (int i = 0; i <= r; i++)
                    for (int j = 0; j <= r; j++) {
                        const int tx = x + xm + i, ty = y + ym + j;
                        if (tx < 0 || ty < 0 || tx >= W || ty >= H) continue;
                        const FhGeometryPixel q = img[(size_t)ty * W + tx];
                        if (q.depth != 0) score += dot(v3(q.normal[0], q.normal[1], q.normal[2]), mean);
                    }
                if (!have || of_cmp(score, best_score) >= 0) { have = true; best_score = score; best = mean; }   // max_by_key: last maximum
            }
        }
        o.normal[0] = best.x; o.normal[1] = best.y; o.normal[2] = best.z;
    }
    out[(size_t)y * W + x] = o;
}

// effects.rs:73-95 compute_ssao + 156-250 compute_pixel_ssao; kernel [nk][3], noise [nn][2]
__global__ void k_fx_ssao(const FhGeometryPixel* __restrict__ img, int W, int H, int D, const float* __restrict__ kernel, int nk,
                          const float* __restrict__ noise, int nn, float* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    const FhGeometryPixel px = img[(size_t)y * W + x];
    const uint32_t d = px.depth;
    float res = __uint_as_float(0x7fc00000u);
    if (d != 0) {
        const float scale_min = (float)min(min(W, H), D);
        const float sx = scale_min / (float)W, sy = scale_min / (float)H, sz = scale_min / (float)D;
        const V3 p = v3((((float)x + 0.5f) / (float)W - 0.5f) * 2.0f, (((float)y + 0.5f) / (float)H - 0.5f) * 2.0f, (((float)d / (float)D) - 0.5f) * 2.0f);
        const V3 n = normalize(v3(px.normal[0], px.normal[1], px.normal[2]));
        const uint32_t ri = pcg((uint32_t)y + pcg((uint32_t)x)) % (uint32_t)nn;   // rng::mix(pos.0 = y, pos.1 = x)
        const V3 rvec = v3(noise[2 * ri], noise[2 * ri + 1], 0.0f);
        const V3 tangent = normalize(sub(rvec, mul(n, dot(rvec, n))));
        const V3 bitangent = cross(n, tangent);
        const float RADIUS = 0.1f;
        float occlusion = 0.0f;
        for (int i = 0; i < nk; i++) {
            const float k0 = kernel[3 * i], k1 = kernel[3 * i + 1], k2 = kernel[3 * i + 2];
            V3 off = v3(tangent.x * k0, tangent.y * k0, tangent.z * k0);
            off = v3(bitangent.x * k1 + off.x, bitangent.y * k1 + off.y, bitangent.z * k1 + off.z);
            off = v3(n.x * k2 + off.x, n.y * k2 + off.y, n.z * k2 + off.z);
            off = mul(off, RADIUS);
            off.x *= sx; off.y *= sy; off.z *= sz;
            const V3 sp = add(off, p);
            const float fx = ((sp.x / 2.0f) + 0.5f) * (float)W, fy = ((sp.y / 2.0f) + 0.5f) * (float)H;
            uint32_t actual_h = 0;
            if (fx < (float)W && fy < (float)H && fx > 0.0f && fy > 0.0f) actual_h = img[(size_t)(uint32_t)fy * W + (uint32_t)fx].depth;
            const float actual_z = (((float)actual_h / (float)D) - 0.5f) * 2.0f;
            const float dz = sp.z - actual_z;
            if (dz < RADIUS) occlusion += (sp.z <= actual_z) ? 1.0f : 0.0f;
            else if (dz < RADIUS * 2.0f && sp.z <= actual_z) { const float t = (RADIUS - (dz - RADIUS)) / RADIUS; occlusion += t * t; }
        }
        res = 1.0f - (occlusion / (float)nk);
    }
    out[(size_t)y * W + x] = res;
}

// effects.rs:98-115 blur_ssao + 329-392 compute_pixel_blur (radius 2)
__global__ void k_fx_blur(const float* __restrict__ ssao, int W, int H, float* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    const float s0 = ssao[(size_t)y * W + x];
    float best = s0;
    if (s0 == s0) {
        const int r = 2;
        bool have = false;
        float best_dev = 0.0f;
        for (int w = 0; w < 4; w++) {
            const int xm = (w & 1) ? -r : 0, ym = (w & 2) ? -r : 0;
            float sum = 0.0f;
            int count = 0;
            for (int i = 0; i <= r; i++)
                for (int j = 0; j <= r; j++) {
                    const int tx = x + xm + i, ty = y + ym + j;
                    if (tx < 0 || ty < 0 || tx >= W || ty >= H) continue;
                    const float s = ssao[(size_t)ty * W + tx];
                    if (s == s) { sum += s; count++; }
                }
            if (count == 0) continue;
            const float mean = sum / (float)count;
            float stdev = 0.0f;
            for (int i = 0; i <= r; i++)
                for (int j = 0; j <= r; j++) {
                    const int tx = x + xm + i, ty = y + ym + j;
                    if (tx < 0 || ty < 0 || tx >= W || ty >= H) continue;
                    const float s = ssao[(size_t)ty * W + tx];
                    if (s == s) { const float e = mean - s; stdev += e * e; }
                }
            const float dev = stdev / (float)count;
            if (!have || of_cmp(dev, best_dev) < 0) { have = true; best_dev = dev; best = mean; }   // min_by_key: first minimum
        }
    }
    out[(size_t)y * W + x] = best;
}

// effects.rs:42-67 apply_shading + 118-153 shade_pixel; ssao may be null; out: 3 bytes per pixel
__global__ void k_fx_shade(const FhGeometryPixel* __restrict__ img, int W, int H, int D, const float* __restrict__ ssao, uint8_t* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    const FhGeometryPixel px = img[(size_t)y * W + x];
    uint8_t c = 0;
    if (px.depth > 0) {
        const V3 n = normalize(v3(px.normal[0], px.normal[1], px.normal[2]));
        const V3 p = v3(2.0f * ((float)x / (float)W - 0.5f), 2.0f * ((float)y / (float)H - 0.5f), 2.0f * ((float)px.depth / (float)D - 0.5f));
        const float lights[3][4] = {{5.0f, -5.0f, 10.0f, 0.5f}, {-5.0f, 0.0f, 10.0f, 0.15f}, {0.0f, -5.0f, 10.0f, 0.15f}};
        float accum = 0.2f;
        for (int l = 0; l < 3; l++) {
            const V3 dir = normalize(sub(v3(lights[l][0], lights[l][1], lights[l][2]), p));
            accum += fmaxf(dot(dir, n), 0.0f) * lights[l][3];
        }
        if (ssao) accum *= ssao[(size_t)y * W + x] * 0.6f + 0.4f;
        if (accum < 0.0f) accum = 0.0f;
        if (accum > 1.0f) accum = 1.0f;
        c = to_u8(accum * 255.0f);
    }
    uint8_t* o = out + ((size_t)y * W + x) * 3;
    o[0] = o[1] = o[2] = c;
}

// ---- 2D: RawDistancePixel (pixel.rs:159-241) -> RGBA; mode 0 to_rgba_bitmap (443-464), 1 same with transparent, 2 to_debug_bitmap
// (467-496), 3 to_rgba_distance (506-547)
__device__ __forceinline__ bool px_is_distance(float v) { return !(v != v) || (__float_as_uint(v) & (0xFFu << 9)) != (0xF6u << 9); }
__global__ void k_fx_rgba(const float* __restrict__ img, size_t n, int mode, uchar4* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float f = img[i];
    const bool dist = px_is_distance(f);
    const uint32_t bits = __float_as_uint(f);
    const bool fill_inside = (bits & 1u) == 1u;
    uchar4 o = make_uchar4(0, 0, 0, 255);
    if (mode <= 1) {
        const bool inside = dist ? f < 0.0f : fill_inside;
        if (inside) o = make_uchar4(255, 255, 255, 255);
        else if (mode == 1) o = make_uchar4(0, 0, 0, 0);
    } else if (mode == 2) {
        if (dist) { const uint8_t c = f < 0.0f ? 255 : 0; o = make_uchar4(c, c, c, 255); }
        else {
            const uint8_t depth = (uint8_t)(bits >> 1), hi = fill_inside ? 255 : 50;
            if (depth == 0) o.x = hi; else if (depth == 1) o.y = hi; else if (depth == 2) o.z = hi; else { o.x = hi; o.y = hi; }
        }
    } else {
        if (!dist) o = fill_inside ? make_uchar4(184, 235, 255, 255) : make_uchar4(217, 144, 72, 255);
        else if (f != f) o = make_uchar4(255, 0, 0, 255);
        else {
            const float rgb[3] = {1.0f - copysignf(0.1f, f), 1.0f - copysignf(0.4f, f), 1.0f - copysignf(0.7f, f)};
            const float af = fabsf(f);
            const float dim = 1.0f - fhd::t_exp(-4.0f * af);
            const float bands = 0.8f + 0.2f * fhd::t_cos(140.0f * f);
            uint8_t c[3];
            for (int k = 0; k < 3; k++) {
                float v = rgb[k] * dim * bands;
                for (int pass = 0; pass < 2; pass++) {
                    const float e1 = pass == 0 ? 0.015f : 0.005f;
                    float t = (af - 0.0f) / (e1 - 0.0f);
                    if (t < 0.0f) t = 0.0f;
                    if (t > 1.0f) t = 1.0f;
                    const float a = 1.0f - t * t * (3.0f - 2.0f * t);
                    v = v * (1.0f - a) + 1.0f * a;
                }
                if (v < 0.0f) v = 0.0f;
                if (v > 1.0f) v = 1.0f;
                c[k] = to_u8(v * 255.0f);
            }
            o = make_uchar4(c[0], c[1], c[2], 255);
        }
    }
    out[i] = o;
}

}  // namespace fhfx
