// ORACLE — TEST INFRASTRUCTURE ONLY (see types.hpp header).
//
// Restates the tile-recursive 2D / 3D renderers that drive the evaluators:
//   fidget-core/src/render/region.rs   (screen_to_world, 87-108)
//   fidget-core/src/render/mod.rs      (RenderHandle::simplify, 96-152)
//   fidget-core/src/shape/mod.rs       (eval_raw 494-540 / 719-802, Transformable 894-948)
//   fidget-raster/src/lib.rs           (TileSizesRef::new 59-66, pixel_offset 83-90,
//                                       render_tiles 99-167)
//   fidget-raster/src/pixel.rs         (2D: RawDistancePixel 159-241, Worker 246-441, render 452-492)
//   fidget-raster/src/voxel.rs         (3D: GeometryPixel 122-134, Worker 190-484, render 500-553)
//
// nalgebra 0.35.0 (Cargo.lock:2682) is not vendored in the reference tree; the
// f32 point transform restates its published gemv/axpy structure:
//   out_i = ((m_i0*x + m_i1*y) + m_i2*z) + m_i3,  n = ((m30*x + m31*y) + m32*z) + m33,
//   out_i / n when n != 0   (Matrix::transform_point).
#pragma once
#include <omp.h>

#include "vm.hpp"

namespace orc {

// Row-major 4x4
struct Mat4 {
    float m[16];
    float at(int r, int c) const { return m[r * 4 + c]; }
};

// region.rs:87-108 for N = 2 (3x3) and N = 3 (4x4); `dim` = N + 1.
// out is row-major dim x dim.
static inline void screen_to_world(const uint32_t* size, int N, float* out) {
    const int D = N + 1;
    float center[3];
    uint32_t mn = size[0];
    for (int i = 0; i < N; i++) {
        center[i] = (float)size[i] / 2.0f;
        mn = std::min(mn, size[i]);
    }
    center[1] -= 1.0f;
    float scale = 2.0f / (float)mn;
    for (int r = 0; r < D; r++)
        for (int c = 0; c < D; c++) out[r * D + c] = (r == c) ? 1.0f : 0.0f;
    // append_translation_mut(&(-center)): self[(j,i)] += shift[j] * self[(D-1,i)]
    for (int i = 0; i < D; i++)
        for (int j = 0; j < D - 1; j++) {
            float add = (-center[j]) * out[(D - 1) * D + i];
            out[j * D + i] += add;
        }
    // append_nonuniform_scaling_mut: row i *= scaling[i], scaling[1] negated
    for (int i = 0; i < N; i++) {
        float s = scale;
        if (i == 1) s *= -1.0f;
        for (int c = 0; c < D; c++) out[i * D + c] *= s;
    }
}

// nalgebra matrix product (gemm -> per-column gemv -> axcpy), dim x dim
static inline void mat_mul(const float* a, const float* b, int D, float* out) {
    for (int j = 0; j < D; j++)
        for (int i = 0; i < D; i++) {
            float acc = a[i * D + 0] * b[0 * D + j];
            for (int k = 1; k < D; k++) acc = a[i * D + k] * b[k * D + j] + acc;
            out[i * D + j] = acc;
        }
}

// pixel.rs:281-285: lift a 3x3 to 4x4 preserving Z
static inline Mat4 lift_2d(const float* m3) {
    Mat4 o;
    const float t[16] = {m3[0], m3[1], 0.0f, m3[2], m3[3], m3[4], 0.0f, m3[5],
                         0.0f,  0.0f,  1.0f, 0.0f,  m3[6], m3[7], 0.0f, m3[8]};
    std::memcpy(o.m, t, sizeof(t));
    return o;
}

// shape/mod.rs:906-916 + nalgebra transform_point
static inline void transform_f32(const Mat4& t, float x, float y, float z, float* ox, float* oy, float* oz) {
    float n = ((t.m[12] * x + t.m[13] * y) + t.m[14] * z) + t.m[15];
    float a = ((t.m[0] * x + t.m[1] * y) + t.m[2] * z) + t.m[3];
    float b = ((t.m[4] * x + t.m[5] * y) + t.m[6] * z) + t.m[7];
    float c = ((t.m[8] * x + t.m[9] * y) + t.m[10] * z) + t.m[11];
    if (n != 0.0f) {
        a = a / n;
        b = b / n;
        c = c / n;
    }
    *ox = a; *oy = b; *oz = c;
}
// shape/mod.rs:918-932
static inline void transform_interval(const Mat4& t, Interval x, Interval y, Interval z, Interval* o) {
    Interval r[4];
    for (int i = 0; i < 4; i++) {
        r[i] = i_add(i_add(i_add(i_mul_f(x, t.m[i * 4 + 0]), i_mul_f(y, t.m[i * 4 + 1])), i_mul_f(z, t.m[i * 4 + 2])),
                     Interval(t.m[i * 4 + 3]));
    }
    o[0] = i_div(r[0], r[3]);
    o[1] = i_div(r[1], r[3]);
    o[2] = i_div(r[2], r[3]);
}
// shape/mod.rs:934-948
static inline void transform_grad(const Mat4& t, Grad x, Grad y, Grad z, Grad* o) {
    Grad r[4];
    for (int i = 0; i < 4; i++) {
        r[i] = g_add(g_add(g_add(g_mul_f(x, t.m[i * 4 + 0]), g_mul_f(y, t.m[i * 4 + 1])), g_mul_f(z, t.m[i * 4 + 2])),
                     Grad(t.m[i * 4 + 3]));
    }
    o[0] = g_div(r[0], r[3]);
    o[1] = g_div(r[1], r[3]);
    o[2] = g_div(r[2], r[3]);
}

// Counters used for the "algorithmic bytes" figure and to size
// the GPU work; pruning is deterministic so these are exact.
struct RenderStats {
    uint64_t interval_evals = 0;     // tiles interval-evaluated
    uint64_t interval_ops = 0;       // sum of tape lengths over those
    uint64_t interval_choices = 0;   // sum of choice counts over those
    uint64_t simplify_calls = 0;     // VmData::simplify executions (cache misses)
    uint64_t simplify_in = 0;        // sum parent tape lengths
    uint64_t simplify_out = 0;       // sum child tape lengths (kept or not)
    uint64_t simplify_kept = 0;      // children kept (shorter than parent)
    uint64_t tiles_full = 0, tiles_empty = 0, tiles_ambiguous = 0, tiles_skipped = 0;
    uint64_t float_evals = 0;        // bulk f32 calls (leaf tiles)
    uint64_t float_points = 0;
    uint64_t float_lane_ops = 0;     // sum tape_len * points
    uint64_t float_wave_ops = 0;     // sum tape_len * ceil(points/64)
    uint64_t grad_evals = 0, grad_points = 0, grad_lane_ops = 0, grad_tape_ops = 0;
    uint64_t invalid_intervals = 0;
    void add(const RenderStats& o) {
        const uint64_t* s = (const uint64_t*)&o;
        uint64_t* d = (uint64_t*)this;
        for (size_t i = 0; i < sizeof(RenderStats) / 8; i++) d[i] += s[i];
    }
};

enum SimplifyMode { SIMPLIFY_REFERENCE = 0, SIMPLIFY_NEVER = 1, SIMPLIFY_ALWAYS = 2 };

// render/mod.rs:19-179 (only the parts that affect which tape evaluates a tile)
struct RenderHandle {
    VmDataP shape;
    std::unique_ptr<RenderHandle> next;
    std::vector<uint8_t> next_trace;
    bool has_next = false;
    explicit RenderHandle(VmDataP s) : shape(std::move(s)) {}

    RenderHandle* simplify(const std::vector<uint8_t>& trace, int mode, RenderStats& st) {
        if (mode == SIMPLIFY_NEVER) return this;
        if (has_next && next_trace != trace) {
            next.reset();
            has_next = false;
        }
        if (!has_next) {
            VmDataP n = vmdata_simplify(*shape, trace.data(), trace.size(), shape->N);
            st.simplify_calls++;
            st.simplify_in += shape->len();
            st.simplify_out += n->len();
            if (mode != SIMPLIFY_ALWAYS && n->len() >= shape->len()) return this;  // mod.rs:125-129
            st.simplify_kept++;
            next.reset(new RenderHandle(n));
            next_trace = trace;
            has_next = true;
        }
        return next.get();
    }
};

// fidget-raster/src/lib.rs:59-66
static inline std::vector<uint32_t> tile_sizes_ref(const std::vector<uint32_t>& tiles, uint32_t max_size) {
    size_t i = tiles.size();
    for (size_t k = 0; k < tiles.size(); k++)
        if (tiles[k] < max_size) { i = k; break; }
    i = (i == 0) ? 0 : i - 1;
    return std::vector<uint32_t>(tiles.begin() + i, tiles.end());
}

// Map X/Y/Z onto the tape's variable slots (shape/mod.rs:518-532, 764-774).
struct Axes {
    int ix, iy, iz, n;
    std::vector<std::pair<int, float>> bound;  // (slot, value) for Var::V entries (ShapeVars<f32>)
    explicit Axes(const VarMap& v) : ix(v.x), iy(v.y), iz(v.z), n(v.len()) {}
};
// BoundShape::new (shape/mod.rs:848-857): every Var::V must have a value
static inline bool bind_vars(const VarMap& vm, const uint64_t* keys, const float* vals, size_t n, Axes& axes) {
    for (auto& kv : vm.v) {
        bool found = false;
        for (size_t i = 0; i < n; i++)
            if (keys[i] == kv.first) { axes.bound.push_back({kv.second, vals[i]}); found = true; break; }
        if (!found) return false;
    }
    return true;
}

// ---------------------------------------------------------------------------
// 2D (pixel.rs)
static const uint32_t PIXEL_KEY = 0xF6u << 9;
static inline float pixel_fill(uint32_t depth, bool inside) {  // pixel.rs:219-233
    return u2f(0x7FC00000u | ((depth & 0xFF) << 1) | (inside ? 1u : 0u) | PIXEL_KEY);
}
static inline float pixel_value(float p) { return std::isnan(p) ? u2f(0x7FC00000u) : p; }  // 235-241

struct Worker2D {
    std::vector<uint32_t> ts;
    Mat4 transform;
    float z;
    bool pixel_perfect;
    int mode;
    Axes axes;
    TracingEval<Interval> eval_interval;
    BulkEval<float> eval_float;
    std::vector<float> sx, sy, sz;
    std::vector<float> image;  // T0 x T0
    RenderStats st;
    std::vector<Interval> ivars;

    std::vector<std::vector<float>> bound_arrays;
    Worker2D(const std::vector<uint32_t>& ts_, const Mat4& t, float z_, bool pp, int mode_, const Axes& ax)
        : ts(ts_), transform(t), z(z_), pixel_perfect(pp), mode(mode_), axes(ax) {
        for (auto& b : axes.bound) bound_arrays.push_back(std::vector<float>((size_t)ts.back() * ts.back(), b.second));
        size_t n = (size_t)ts.back() * ts.back();
        sx.resize(n); sy.resize(n); sz.resize(n);
        ivars.resize(std::max(axes.n, 1));
    }
    size_t pixel_offset(uint32_t x, uint32_t y) const { return (x % ts[0]) + (size_t)(y % ts[0]) * ts[0]; }

    void render_tile(RenderHandle* shape, uint32_t cx, uint32_t cy) {
        image.assign((size_t)ts[0] * ts[0], 0.0f);
        recurse(shape, 0, cx, cy);
    }
    void recurse(RenderHandle* shape, size_t depth, uint32_t cx, uint32_t cy) {
        const uint32_t tile_size = ts[depth];
        Interval x((float)cx, (float)cx + (float)tile_size);
        Interval y((float)cy, (float)cy + (float)tile_size);
        Interval zz(z, z);
        Interval tr[3];
        transform_interval(transform, x, y, zz, tr);
        for (auto& v : ivars) v = Interval(0.0f);
        if (axes.ix >= 0) ivars[axes.ix] = tr[0];
        if (axes.iy >= 0) ivars[axes.iy] = tr[1];
        if (axes.iz >= 0) ivars[axes.iz] = tr[2];
        for (auto& b : axes.bound) ivars[b.first] = Interval(b.second);
        int simplify = eval_interval.eval(*shape->shape, ivars.data(), ivars.size());
        Interval i = eval_interval.out[0];
        st.interval_evals++;
        st.interval_ops += shape->shape->len();
        st.interval_choices += shape->shape->choice_count();

        if (!pixel_perfect) {
            int fill = -1;
            if (i.hi < 0.0f) fill = 1;
            else if (i.lo > 0.0f) fill = 0;
            if (fill >= 0) {
                if (fill) st.tiles_full++; else st.tiles_empty++;
                float f = pixel_fill((uint32_t)depth, fill == 1);
                for (uint32_t yy = 0; yy < tile_size; yy++) {
                    size_t start = pixel_offset(cx, cy + yy);
                    for (uint32_t xx = 0; xx < tile_size; xx++) image[start + xx] = f;
                }
                return;
            }
        }
        st.tiles_ambiguous++;
        RenderHandle* sub = shape;
        if (simplify == 1) sub = shape->simplify(eval_interval.choices, mode, st);

        if (depth + 1 < ts.size()) {
            uint32_t next = ts[depth + 1];
            uint32_t n = tile_size / next;
            for (uint32_t j = 0; j < n; j++)
                for (uint32_t ii = 0; ii < n; ii++) recurse(sub, depth + 1, cx + ii * next, cy + j * next);
        } else {
            pixels(sub, tile_size, cx, cy);
        }
    }
    void pixels(RenderHandle* shape, uint32_t tile_size, uint32_t cx, uint32_t cy) {
        size_t index = 0;
        for (uint32_t j = 0; j < tile_size; j++)
            for (uint32_t i = 0; i < tile_size; i++) {
                transform_f32(transform, (float)(cx + i), (float)(cy + j), z, &sx[index], &sy[index], &sz[index]);
                index++;
            }
        std::vector<float> zeros;
        const float* vars[8];
        int nv = std::max(axes.n, 1);
        zeros.assign(index, 0.0f);
        for (int k = 0; k < nv && k < 8; k++) vars[k] = zeros.data();
        if (axes.ix >= 0) vars[axes.ix] = sx.data();
        if (axes.iy >= 0) vars[axes.iy] = sy.data();
        if (axes.iz >= 0) vars[axes.iz] = sz.data();
        for (size_t q = 0; q < axes.bound.size(); q++) vars[axes.bound[q].first] = bound_arrays[q].data();
        eval_float.eval(*shape->shape, vars, nv, index);
        st.float_evals++;
        st.float_points += index;
        st.float_lane_ops += (uint64_t)shape->shape->len() * index;
        st.float_wave_ops += (uint64_t)shape->shape->len() * ((index + 63) / 64);
        const float* out = eval_float.out[0].data();
        index = 0;
        for (uint32_t j = 0; j < tile_size; j++) {
            size_t o = pixel_offset(cx, cy + j);
            for (uint32_t i = 0; i < tile_size; i++) image[o + i] = pixel_value(out[index++]);
        }
    }
};

struct RenderResult {
    RenderStats stats;
    bool ok = true;
};

// pixel.rs:452-492 + lib.rs:99-167.  `mat3` = world_to_model * screen_to_world
// already combined by the caller (row-major 3x3).  out = width*height floats.
static inline RenderResult render_2d(const VmDataP& shape, const float* mat3, uint32_t width, uint32_t height, float z,
                                     bool pixel_perfect, const std::vector<uint32_t>& tile_sizes, int mode,
                                     int threads, float* out, const uint64_t* var_keys = nullptr,
                                     const float* var_vals = nullptr, size_t n_vars = 0) {
    RenderResult res;
    Axes axes(*shape->vars);
    if (!bind_vars(*shape->vars, var_keys, var_vals, n_vars, axes)) { res.ok = false; return res; }  // MissingVar
    std::vector<uint32_t> ts = tile_sizes_ref(tile_sizes, std::max(width, height));
    const uint32_t t0 = ts[0];
    std::vector<std::pair<uint32_t, uint32_t>> tiles;
    for (uint32_t i = 0; i < (width + t0 - 1) / t0; i++)
        for (uint32_t j = 0; j < (height + t0 - 1) / t0; j++) tiles.push_back({i * t0, j * t0});
    Mat4 transform = lift_2d(mat3);
    for (size_t i = 0; i < (size_t)width * height; i++) out[i] = 0.0f;
    if (threads <= 0) threads = omp_get_max_threads();
    RenderStats total;
#pragma omp parallel num_threads(threads)
    {
        Worker2D w(ts, transform, z, pixel_perfect, mode, axes);
        RenderHandle rh(shape);
        g_invalid_intervals = 0;
#pragma omp for schedule(dynamic, 1)
        for (size_t k = 0; k < tiles.size(); k++) {
            w.render_tile(&rh, tiles[k].first, tiles[k].second);
            size_t index = 0;
            for (uint32_t j = 0; j < t0; j++) {
                uint32_t y = j + tiles[k].second;
                for (uint32_t i = 0; i < t0; i++) {
                    uint32_t x = i + tiles[k].first;
                    if (y < height && x < width) out[(size_t)y * width + x] = w.image[index];
                    index++;
                }
            }
        }
        w.st.invalid_intervals = g_invalid_intervals;
#pragma omp critical
        total.add(w.st);
    }
    res.stats = total;
    return res;
}

// ---------------------------------------------------------------------------
// 3D (voxel.rs)
struct GeometryPixel {  // voxel.rs:122-134
    float normal[3];
    uint32_t depth;
};

struct Worker3D {
    std::vector<uint32_t> ts;
    Mat4 transform;
    uint32_t image_depth;
    int mode;
    Axes axes;
    TracingEval<Interval> eval_interval;
    BulkEval<float> eval_float;
    BulkEval<Grad> eval_grad;
    std::vector<float> sx, sy, sz, zeros;
    std::vector<Grad> gx, gy, gz, gzeros;
    std::vector<size_t> columns;
    std::vector<GeometryPixel> out;  // T0 x T0
    std::vector<Interval> ivars;
    RenderStats st;

    std::vector<std::vector<float>> bound_arrays;
    std::vector<std::vector<Grad>> bound_grads;
    Worker3D(const std::vector<uint32_t>& ts_, const Mat4& t, uint32_t depth, int mode_, const Axes& ax)
        : ts(ts_), transform(t), image_depth(depth), mode(mode_), axes(ax) {
        size_t b = ts.back();
        for (auto& bv : axes.bound) {
            bound_arrays.push_back(std::vector<float>(b * b * b, bv.second));
            bound_grads.push_back(std::vector<Grad>(b * b, Grad(bv.second)));
        }
        sx.resize(b * b * b); sy.resize(b * b * b); sz.resize(b * b * b); zeros.assign(b * b * b, 0.0f);
        gx.resize(b * b); gy.resize(b * b); gz.resize(b * b); gzeros.assign(b * b, Grad(0.0f));
        ivars.resize(std::max(axes.n, 1));
    }
    size_t pixel_offset(uint32_t x, uint32_t y) const { return (x % ts[0]) + (size_t)(y % ts[0]) * ts[0]; }

    // voxel.rs:244-263
    void render_tile(RenderHandle* shape, uint32_t cx, uint32_t cy) {
        out.assign((size_t)ts[0] * ts[0], GeometryPixel{{0, 0, 0}, 0});
        uint32_t nk = (image_depth + ts[0] - 1) / ts[0];
        for (uint32_t k = nk; k-- > 0;) {
            if (!recurse(shape, 0, cx, cy, k * ts[0])) break;
        }
    }
    // voxel.rs:275-357
    bool recurse(RenderHandle* shape, size_t depth, uint32_t cx, uint32_t cy, uint32_t cz) {
        const uint32_t tile_size = ts[depth];
        const uint32_t fill_z = cz + tile_size + 1;
        bool all = true;
        for (uint32_t y = 0; y < tile_size && all; y++) {
            size_t i = pixel_offset(cx, cy + y);
            for (uint32_t x = 0; x < tile_size; x++)
                if (!(out[i + x].depth >= fill_z)) { all = false; break; }
        }
        if (all) { st.tiles_skipped++; return false; }

        Interval x((float)cx, (float)cx + (float)tile_size);
        Interval y((float)cy, (float)cy + (float)tile_size);
        Interval z((float)cz, (float)cz + (float)tile_size);
        Interval tr[3];
        transform_interval(transform, x, y, z, tr);
        for (auto& v : ivars) v = Interval(0.0f);
        if (axes.ix >= 0) ivars[axes.ix] = tr[0];
        if (axes.iy >= 0) ivars[axes.iy] = tr[1];
        if (axes.iz >= 0) ivars[axes.iz] = tr[2];
        for (auto& b : axes.bound) ivars[b.first] = Interval(b.second);
        int simplify = eval_interval.eval(*shape->shape, ivars.data(), ivars.size());
        Interval i = eval_interval.out[0];
        st.interval_evals++;
        st.interval_ops += shape->shape->len();
        st.interval_choices += shape->shape->choice_count();

        if (i.hi < 0.0f) {
            st.tiles_full++;
            for (uint32_t yy = 0; yy < tile_size; yy++) {
                size_t o = pixel_offset(cx, cy + yy);
                for (uint32_t xx = 0; xx < tile_size; xx++) out[o + xx].depth = std::max(out[o + xx].depth, fill_z);
            }
            return false;
        } else if (i.lo > 0.0f) {
            st.tiles_empty++;
            return true;
        }
        st.tiles_ambiguous++;
        RenderHandle* sub = shape;
        if (simplify == 1) sub = shape->simplify(eval_interval.choices, mode, st);

        if (depth + 1 < ts.size()) {
            uint32_t next = ts[depth + 1];
            uint32_t n = tile_size / next;
            for (uint32_t j = 0; j < n; j++)
                for (uint32_t ii = 0; ii < n; ii++)
                    for (uint32_t k = n; k-- > 0;) recurse(sub, depth + 1, cx + ii * next, cy + j * next, cz + k * next);
        } else {
            pixels(sub, tile_size, cx, cy, cz);
        }
        return true;
    }
    // voxel.rs:359-483
    void pixels(RenderHandle* shape, uint32_t tile_size, uint32_t cx, uint32_t cy, uint32_t cz) {
        size_t index = 0;
        columns.clear();
        for (uint32_t xy = 0; xy < tile_size * tile_size; xy++) {
            uint32_t i = xy % tile_size, j = xy / tile_size;
            size_t o = pixel_offset(cx + i, cy + j);
            uint32_t zmax = cz + tile_size;
            if (out[o].depth >= zmax) continue;
            for (uint32_t k = tile_size; k-- > 0;) {
                transform_f32(transform, (float)(cx + i), (float)(cy + j), (float)(cz + k), &sx[index], &sy[index],
                              &sz[index]);
                index++;
            }
            columns.push_back(xy);
        }
        size_t size = index;
        assert(size > 0);
        const float* vars[8];
        int nv = std::max(axes.n, 1);
        for (int k = 0; k < nv && k < 8; k++) vars[k] = zeros.data();
        if (axes.ix >= 0) vars[axes.ix] = sx.data();
        if (axes.iy >= 0) vars[axes.iy] = sy.data();
        if (axes.iz >= 0) vars[axes.iz] = sz.data();
        for (size_t q = 0; q < axes.bound.size(); q++) vars[axes.bound[q].first] = bound_arrays[q].data();
        eval_float.eval(*shape->shape, vars, nv, size);
        st.float_evals++;
        st.float_points += size;
        st.float_lane_ops += (uint64_t)shape->shape->len() * size;
        st.float_wave_ops += (uint64_t)shape->shape->len() * ((size + 63) / 64);
        const float* outv = eval_float.out[0].data();

        size_t grad = 0;
        for (size_t col = 0; col < columns.size(); col++) {
            const float* d = outv + col * tile_size;
            int kf = -1;
            for (uint32_t q = 0; q < tile_size; q++)
                if (d[q] < 0.0f) { kf = (int)q; break; }
            if (kf < 0) continue;
            uint32_t xy = (uint32_t)columns[col];
            uint32_t i = xy % tile_size, j = xy / tile_size;
            uint32_t k = tile_size - 1 - (uint32_t)kf;
            size_t o = pixel_offset(cx + i, cy + j);
            uint32_t zd = cz + k + 1;
            assert(out[o].depth < zd);
            out[o].depth = zd;
            Grad px((float)(cx + i), 1, 0, 0), py((float)(cy + j), 0, 1, 0), pz((float)(cz + k), 0, 0, 1);
            Grad t3[3];
            transform_grad(transform, px, py, pz, t3);
            gx[grad] = t3[0]; gy[grad] = t3[1]; gz[grad] = t3[2];
            columns[grad] = o;
            grad++;
        }
        if (grad > 0) {
            const Grad* gv[8];
            for (int k = 0; k < nv && k < 8; k++) gv[k] = gzeros.data();
            if (axes.ix >= 0) gv[axes.ix] = gx.data();
            if (axes.iy >= 0) gv[axes.iy] = gy.data();
            if (axes.iz >= 0) gv[axes.iz] = gz.data();
            for (size_t q = 0; q < axes.bound.size(); q++) gv[axes.bound[q].first] = bound_grads[q].data();
            eval_grad.eval(*shape->shape, gv, nv, grad);
            st.grad_evals++;
            st.grad_points += grad;
            st.grad_lane_ops += (uint64_t)shape->shape->len() * grad;
            st.grad_tape_ops += shape->shape->len();
            const Grad* g = eval_grad.out[0].data();
            for (size_t q = 0; q < grad; q++) {
                GeometryPixel& p = out[columns[q]];
                p.normal[0] = g[q].dx; p.normal[1] = g[q].dy; p.normal[2] = g[q].dz;
            }
        }
    }
};

// voxel.rs:500-553.  `mat4` = world_to_model * screen_to_world (row-major).
static inline RenderResult render_3d(const VmDataP& shape, const float* mat4, uint32_t width, uint32_t height,
                                     uint32_t depth, const std::vector<uint32_t>& tile_sizes, int mode, int threads,
                                     GeometryPixel* image, const uint64_t* var_keys = nullptr,
                                     const float* var_vals = nullptr, size_t n_vars = 0) {
    RenderResult res;
    Axes axes(*shape->vars);
    if (!bind_vars(*shape->vars, var_keys, var_vals, n_vars, axes)) { res.ok = false; return res; }
    std::vector<uint32_t> ts = tile_sizes_ref(tile_sizes, std::max(width, height));
    const uint32_t t0 = ts[0];
    std::vector<std::pair<uint32_t, uint32_t>> tiles;
    for (uint32_t i = 0; i < (width + t0 - 1) / t0; i++)
        for (uint32_t j = 0; j < (height + t0 - 1) / t0; j++) tiles.push_back({i * t0, j * t0});
    Mat4 transform;
    std::memcpy(transform.m, mat4, sizeof(transform.m));
    for (size_t i = 0; i < (size_t)width * height; i++) image[i] = GeometryPixel{{0, 0, 0}, 0};
    if (threads <= 0) threads = omp_get_max_threads();
    RenderStats total;
#pragma omp parallel num_threads(threads)
    {
        Worker3D w(ts, transform, depth, mode, axes);
        RenderHandle rh(shape);
        g_invalid_intervals = 0;
#pragma omp for schedule(dynamic, 1)
        for (size_t k = 0; k < tiles.size(); k++) {
            w.render_tile(&rh, tiles[k].first, tiles[k].second);
            size_t index = 0;
            for (uint32_t j = 0; j < t0; j++) {
                uint32_t y = j + tiles[k].second;
                for (uint32_t i = 0; i < t0; i++) {
                    uint32_t x = i + tiles[k].first;
                    if (x < width && y < height) {
                        size_t o = (size_t)y * width + x;
                        if (w.out[index].depth >= image[o].depth) {
                            uint32_t d = depth - 1;
                            if (w.out[index].depth >= d) image[o] = GeometryPixel{{0.0f, 0.0f, 1.0f}, d + 1};
                            else image[o] = w.out[index];
                        }
                    }
                    index++;
                }
            }
        }
        w.st.invalid_intervals = g_invalid_intervals;
#pragma omp critical
        total.add(w.st);
    }
    res.stats = total;
    return res;
}

}  // namespace orc
