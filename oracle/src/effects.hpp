// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of fidget-raster/src/effects.rs (post-processing of rendered images): each
// function cites the lines it follows.  f32 arithmetic op by op (-ffp-contract=off); nalgebra's
// small-vector code is restated from its structure: Vector3 dot = (a0*b0 + a1*b1) + a2*b2
// (blas.rs dotx, the 3-row special case), norm = sqrt(0 + dot(v, v)) (norm.rs norm_squared),
// normalize = each component / norm (unscale), Matrix3 * Vector3 by columns: col0*x0, then
// col_j*x_j + acc (blas.rs gemv).  PARITY: the reference has no test or golden image for this file;
// these functions are pinned only by hand-checked small cases in tests/test_effects.py.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "types.hpp"

namespace orc {

struct GeomPx {
    float n[3];
    uint32_t depth;
};

struct V3 {
    float x, y, z;
};
static inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
static inline float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float norm3(V3 a) { return std::sqrt(0.0f + dot3(a, a)); }
static inline V3 normalize3(V3 a) { return a / norm3(a); }
static inline V3 cross3(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

// OrderedFloat total order used by max_by_key / min_by_key: NaN is greater than everything and equal to itself
static inline int of_cmp(float a, float b) {
    const bool na = a != a, nb = b != b;
    if (na || nb) return na == nb ? 0 : (na ? 1 : -1);
    return a < b ? -1 : (a > b ? 1 : 0);
}

// effects.rs:252-326 denoise_pixel
static inline void fx_denoise_pixel(const GeomPx* img, int W, int H, int x, int y, int r, float out[3]) {
    const GeomPx& p = img[(size_t)y * W + x];
    if (p.n[2] > 0.0f) { out[0] = p.n[0]; out[1] = p.n[1]; out[2] = p.n[2]; return; }
    const int win[4][2] = {{0, 0}, {-r, 0}, {0, -r}, {-r, -r}};
    bool have = false;
    float best_score = 0.0f;
    V3 best = v3(p.n[0], p.n[1], p.n[2]);
    for (int w = 0; w < 4; w++) {
        V3 sum = v3(0, 0, 0);
        int count = 0;
        for (int i = 0; i <= r; i++)
            for (int j = 0; j <= r; j++) {
                const int tx = x + win[w][0] + i, ty = y + win[w][1] + j;
                if (tx < 0 || ty < 0 || tx >= W || ty >= H) continue;
                const GeomPx& q = img[(size_t)ty * W + tx];
                if (q.depth != 0 && q.n[2] > 0.0f) { sum = sum + v3(q.n[0], q.n[1], q.n[2]); count++; }
            }
        if (count == 0) continue;
        const V3 mean = sum / (float)count;
        float score = 0.0f;
        for (int i = 0; i <= r; i++)
            for (int j = 0; j <= r; j++) {
                const int tx = x + win[w][0] + i, ty = y + win[w][1] + j;
                if (tx < 0 || ty < 0 || tx >= W || ty >= H) continue;
                const GeomPx& q = img[(size_t)ty * W + tx];
                if (q.depth != 0) score += dot3(v3(q.n[0], q.n[1], q.n[2]), mean);
            }
        // Iterator::max_by_key keeps the LAST of equal maxima
        if (!have || of_cmp(score, best_score) >= 0) { have = true; best_score = score; best = mean; }
    }
    out[0] = best.x; out[1] = best.y; out[2] = best.z;
}

// effects.rs:17-36 denoise_normals
static inline void fx_denoise_normals(const GeomPx* img, int W, int H, GeomPx* out) {
#pragma omp parallel for schedule(dynamic, 8)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            GeomPx o;
            o.depth = img[(size_t)y * W + x].depth;
            o.n[0] = o.n[1] = o.n[2] = 0.0f;
            if (o.depth > 0) fx_denoise_pixel(img, W, H, x, y, 2, o.n);
            out[(size_t)y * W + x] = o;
        }
}

// effects.rs:156-250 compute_pixel_ssao; kernel: 3 x nk column-major (x,y,z per sample), noise: 2 x nn
static inline float fx_pixel_ssao(const GeomPx* img, int W, int H, int D, int x, int y, const float* kernel, int nk, const float* noise, int nn) {
    const GeomPx& px = img[(size_t)y * W + x];
    const uint32_t d = px.depth;
    if (d == 0) return NAN;
    const float scale_min = (float)std::min(std::min(W, H), D);
    const float sx = scale_min / (float)W, sy = scale_min / (float)H, sz = scale_min / (float)D;
    const V3 p = v3((((float)x + 0.5f) / (float)W - 0.5f) * 2.0f, (((float)y + 0.5f) / (float)H - 0.5f) * 2.0f, (((float)d / (float)D) - 0.5f) * 2.0f);
    const V3 n = normalize3(v3(px.n[0], px.n[1], px.n[2]));
    const uint32_t ri = rng_mix((uint32_t)y, (uint32_t)x) % (uint32_t)nn;   // mix(pos.0 = y, pos.1 = x)
    const V3 rvec = v3(noise[2 * ri], noise[2 * ri + 1], 0.0f);
    const V3 tangent = normalize3(rvec - n * dot3(rvec, n));
    const V3 bitangent = cross3(n, tangent);
    const float RADIUS = 0.1f;
    float occlusion = 0.0f;
    for (int i = 0; i < nk; i++) {
        const float k0 = kernel[3 * i], k1 = kernel[3 * i + 1], k2 = kernel[3 * i + 2];
        // tbn * k: columns tangent, bitangent, n
        V3 off = v3(tangent.x * k0, tangent.y * k0, tangent.z * k0);
        off = v3(bitangent.x * k1 + off.x, bitangent.y * k1 + off.y, bitangent.z * k1 + off.z);
        off = v3(n.x * k2 + off.x, n.y * k2 + off.y, n.z * k2 + off.z);
        off = off * RADIUS;
        off.x *= sx; off.y *= sy; off.z *= sz;
        const V3 sp = off + p;
        const float fx = ((sp.x / 2.0f) + 0.5f) * (float)W, fy = ((sp.y / 2.0f) + 0.5f) * (float)H;
        uint32_t actual_h = 0;
        if (fx < (float)W && fy < (float)H && fx > 0.0f && fy > 0.0f) actual_h = img[(size_t)(uint32_t)fy * W + (uint32_t)fx].depth;
        const float actual_z = (((float)actual_h / (float)D) - 0.5f) * 2.0f;
        const float dz = sp.z - actual_z;
        if (dz < RADIUS) occlusion += (sp.z <= actual_z) ? 1.0f : 0.0f;
        else if (dz < RADIUS * 2.0f && sp.z <= actual_z) { const float t = (RADIUS - (dz - RADIUS)) / RADIUS; occlusion += t * t; }
    }
    return 1.0f - (occlusion / (float)nk);
}

// effects.rs:73-95 compute_ssao
static inline void fx_compute_ssao(const GeomPx* img, int W, int H, int D, const float* kernel, int nk, const float* noise, int nn, float* out) {
#pragma omp parallel for schedule(dynamic, 8)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            out[(size_t)y * W + x] = img[(size_t)y * W + x].depth > 0 ? fx_pixel_ssao(img, W, H, D, x, y, kernel, nk, noise, nn) : NAN;
}

// effects.rs:329-392 compute_pixel_blur
static inline float fx_pixel_blur(const float* ssao, int W, int H, int x, int y, int r) {
    const int win[4][2] = {{0, 0}, {-r, 0}, {0, -r}, {-r, -r}};
    bool have = false;
    float best_dev = 0.0f, best = ssao[(size_t)y * W + x];
    for (int w = 0; w < 4; w++) {
        float sum = 0.0f;
        int count = 0;
        for (int i = 0; i <= r; i++)
            for (int j = 0; j <= r; j++) {
                const int tx = x + win[w][0] + i, ty = y + win[w][1] + j;
                if (tx < 0 || ty < 0 || tx >= W || ty >= H) continue;
                const float s = ssao[(size_t)ty * W + tx];
                if (s == s) { sum += s; count++; }
            }
        if (count == 0) continue;
        const float mean = sum / (float)count;
        float stdev = 0.0f;
        for (int i = 0; i <= r; i++)
            for (int j = 0; j <= r; j++) {
                const int tx = x + win[w][0] + i, ty = y + win[w][1] + j;
                if (tx < 0 || ty < 0 || tx >= W || ty >= H) continue;
                const float s = ssao[(size_t)ty * W + tx];
                if (s == s) { const float e = mean - s; stdev += e * e; }
            }
        const float dev = stdev / (float)count;
        // Iterator::min_by_key keeps the FIRST of equal minima
        if (!have || of_cmp(dev, best_dev) < 0) { have = true; best_dev = dev; best = mean; }
    }
    return best;
}

// effects.rs:98-115 blur_ssao
static inline void fx_blur_ssao(const float* ssao, int W, int H, float* out) {
#pragma omp parallel for schedule(dynamic, 8)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const float s = ssao[(size_t)y * W + x];
            out[(size_t)y * W + x] = (s != s) ? NAN : fx_pixel_blur(ssao, W, H, x, y, 2);
        }
}

// `as u8`: saturating, NaN -> 0
static inline uint8_t fx_u8(float v) {
    if (!(v > 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}

// effects.rs:118-153 shade_pixel
static inline uint8_t fx_shade_pixel(const GeomPx* img, int W, int H, int D, const float* ssao, int x, int y) {
    const GeomPx& px = img[(size_t)y * W + x];
    const V3 n = normalize3(v3(px.n[0], px.n[1], px.n[2]));
    const V3 p = v3(2.0f * ((float)x / (float)W - 0.5f), 2.0f * ((float)y / (float)H - 0.5f), 2.0f * ((float)px.depth / (float)D - 0.5f));
    const float lights[3][4] = {{5.0f, -5.0f, 10.0f, 0.5f}, {-5.0f, 0.0f, 10.0f, 0.15f}, {0.0f, -5.0f, 10.0f, 0.15f}};
    float accum = 0.2f;
    for (int l = 0; l < 3; l++) {
        const V3 dir = normalize3(v3(lights[l][0], lights[l][1], lights[l][2]) - p);
        const float dn = dot3(dir, n);
        accum += rmax(dn, 0.0f) * lights[l][3];
    }
    if (ssao) accum *= ssao[(size_t)y * W + x] * 0.6f + 0.4f;
    // f32::clamp(0, 1): NaN stays NaN (-> 0 by the cast)
    if (accum < 0.0f) accum = 0.0f;
    if (accum > 1.0f) accum = 1.0f;
    return fx_u8(accum * 255.0f);
}

// effects.rs:42-67 apply_shading (ssao = the blurred occlusion map or null); out: W*H*3 bytes
static inline void fx_apply_shading(const GeomPx* img, int W, int H, int D, const float* ssao, uint8_t* out) {
#pragma omp parallel for schedule(dynamic, 8)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const uint8_t c = img[(size_t)y * W + x].depth > 0 ? fx_shade_pixel(img, W, H, D, ssao, x, y) : 0;
            uint8_t* o = out + ((size_t)y * W + x) * 3;
            o[0] = o[1] = o[2] = c;
        }
}

// ---- 2D: RawDistancePixel (pixel.rs:159-241) -> RGBA ---------------------------------------------------
static inline bool px_is_distance(float v) { return !(v != v) || (f2u(v) & (0xFFu << 9)) != (0xF6u << 9); }

// effects.rs:443-464 to_rgba_bitmap
static inline void fx_to_rgba_bitmap(const float* img, size_t n, int transparent, uint8_t* out) {
    for (size_t i = 0; i < n; i++) {
        const float v = img[i];
        const bool inside = px_is_distance(v) ? v < 0.0f : (f2u(v) & 1u) == 1u;
        uint8_t* o = out + 4 * i;
        if (inside) { o[0] = o[1] = o[2] = o[3] = 255; }
        else if (transparent) { o[0] = o[1] = o[2] = o[3] = 0; }
        else { o[0] = o[1] = o[2] = 0; o[3] = 255; }
    }
}

// effects.rs:467-496 to_debug_bitmap
static inline void fx_to_debug_bitmap(const float* img, size_t n, uint8_t* out) {
    for (size_t i = 0; i < n; i++) {
        const float v = img[i];
        uint8_t* o = out + 4 * i;
        o[3] = 255;
        if (px_is_distance(v)) { const uint8_t c = v < 0.0f ? 255 : 0; o[0] = o[1] = o[2] = c; continue; }
        const uint32_t bits = f2u(v);
        const bool inside = (bits & 1u) == 1u;
        const uint8_t depth = (uint8_t)(bits >> 1);
        const uint8_t hi = inside ? 255 : 50;
        o[0] = o[1] = o[2] = 0;
        if (depth == 0) o[0] = hi;
        else if (depth == 1) o[1] = hi;
        else if (depth == 2) o[2] = hi;
        else { o[0] = hi; o[1] = hi; }
    }
}

// effects.rs:506-547 to_rgba_distance
static inline void fx_to_rgba_distance(const float* img, size_t n, uint8_t* out) {
    for (size_t i = 0; i < n; i++) {
        const float f = img[i];
        uint8_t* o = out + 4 * i;
        o[3] = 255;
        if (!px_is_distance(f)) {
            const bool inside = (f2u(f) & 1u) == 1u;
            if (inside) { o[0] = 184; o[1] = 235; o[2] = 255; } else { o[0] = 217; o[1] = 144; o[2] = 72; }
            continue;
        }
        if (f != f) { o[0] = 255; o[1] = 0; o[2] = 0; continue; }
        const float rgb[3] = {1.0f - std::copysign(0.1f, f), 1.0f - std::copysign(0.4f, f), 1.0f - std::copysign(0.7f, f)};
        const float af = std::fabs(f);
        const float dim = 1.0f - std::exp(-4.0f * af);
        const float bands = 0.8f + 0.2f * std::cos(140.0f * f);
        auto smoothstep = [](float e0, float e1, float x) {
            float t = (x - e0) / (e1 - e0);
            if (t < 0.0f) t = 0.0f;
            if (t > 1.0f) t = 1.0f;
            return t * t * (3.0f - 2.0f * t);
        };
        auto mixf = [](float x, float y, float a) { return x * (1.0f - a) + y * a; };
        for (int c = 0; c < 3; c++) {
            float v = rgb[c] * dim * bands;
            v = mixf(v, 1.0f, 1.0f - smoothstep(0.0f, 0.015f, af));
            v = mixf(v, 1.0f, 1.0f - smoothstep(0.0f, 0.005f, af));
            if (v < 0.0f) v = 0.0f;
            if (v > 1.0f) v = 1.0f;
            o[c] = fx_u8(v * 255.0f);
        }
    }
}

}  // namespace orc
