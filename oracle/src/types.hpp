// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the reference's scalar / interval / gradient op semantics.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
// anything under oracle/.  The product (fidget_amd/) never includes this file.
//
// Follows (paths relative to /root/reference):
//   fidget-core/src/types/float.rs     (f32 choice ops)
//   fidget-core/src/types/interval.rs  (Interval)
//   fidget-core/src/types/grad.rs      (Grad)
//   fidget-core/src/rng/mod.rs         (hash / rand / mix)
//   fidget-core/src/vm/choice.rs       (Choice)
//
// Transcendentals call the platform libm (glibc), as Rust's std does on Linux;
// the reference's own tests compare against the same host libm
// (eval/test/mod.rs:194-203), so ulp-level transcendental values are
// "parity unpinned" by the reference itself.
//
// Build with -ffp-contract=off (Rust never contracts a*b+c).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace orc {

// vm/choice.rs:15-29
enum Choice : uint8_t { UNKNOWN = 0, LEFT = 1, RIGHT = 2, BOTH = 3 };

static inline uint32_t f2u(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}
static inline float u2f(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
static const float NANF = std::numeric_limits<float>::quiet_NaN();
static const float PI_F = 3.14159265358979323846f;   // std::f32::consts::PI
static const float TAU_F = 6.28318530717958647692f;  // std::f32::consts::TAU

// Set when Interval::new would have panicked (interval.rs:39-45).  The
// reference aborts; the oracle records it so tests can assert it never fires.
extern thread_local uint64_t g_invalid_intervals;

// ---------------------------------------------------------------------------
// rng/mod.rs:8-33
static inline uint32_t rng_hash(uint32_t v) {
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28) + 4)) ^ state) * 277803737u;
    return (word >> 22) ^ word;
}
static inline float rng_rand(uint32_t seed) {
    uint32_t h = rng_hash(seed);
    uint32_t bits = (h >> 9) | 0x3f800000u;
    return u2f(bits) - 1.0f;
}
static inline uint32_t rng_mix(uint32_t a, uint32_t b) {
    return rng_hash(a + rng_hash(b));
}

// ---------------------------------------------------------------------------
// Rust std f32 helpers
// f32::min / f32::max: IEEE minNum / maxNum (NaN-ignoring)
static inline float rmin(float a, float b) { return fminf(a, b); }
static inline float rmax(float a, float b) { return fmaxf(a, b); }
// f32::rem_euclid (core::f32): r = self % rhs; if r < 0 { r + rhs.abs() }
static inline float rem_euclid(float a, float b) {
    float r = fmodf(a, b);
    return (r < 0.0f) ? r + fabsf(b) : r;
}
// f32::div_euclid
static inline float div_euclid(float a, float b) {
    float q = truncf(a / b);
    if (fmodf(a, b) < 0.0f) {
        return (b > 0.0f) ? q - 1.0f : q + 1.0f;
    }
    return q;
}

// ---------------------------------------------------------------------------
// types/float.rs:66-142 (FloatExt for f32)
struct FC {
    float v;
    Choice c;
};
static inline float f_compare(float a, float b) {
    if (a < b) return -1.0f;
    if (a == b) return 0.0f;
    if (a > b) return 1.0f;
    return NANF;
}
static inline FC f_max_choice(float a, float b) {
    if (a > b) return {a, LEFT};
    if (b > a) return {b, RIGHT};
    return {(std::isnan(a) || std::isnan(b)) ? NANF : b, BOTH};
}
static inline FC f_min_choice(float a, float b) {
    if (a < b) return {a, LEFT};
    if (b < a) return {b, RIGHT};
    return {(std::isnan(a) || std::isnan(b)) ? NANF : b, BOTH};
}
static inline FC f_and_choice(float a, float b) {
    if (a == 0.0f) return {a, LEFT};
    return {b, RIGHT};
}
static inline FC f_or_choice(float a, float b) {
    if (a != 0.0f) return {a, LEFT};
    return {b, RIGHT};
}
static inline float f_rand(float a) { return rng_rand(f2u(a)); }
static inline float f_mix(float a, float b) { return u2f(rng_mix(f2u(a), f2u(b))); }
static inline float f_not(float a) { return (a == 0.0f) ? 1.0f : 0.0f; }

// ---------------------------------------------------------------------------
// types/interval.rs
struct Interval {
    float lo, hi;

    Interval() : lo(0), hi(0) {}
    // interval.rs:38-45 (Interval::new with its validity assertion)
    Interval(float l, float u) : lo(l), hi(u) {
        if (!(u >= l || (std::isnan(l) && std::isnan(u)))) {
            g_invalid_intervals++;
        }
    }
    // interval.rs:643-648 (From<f32>)
    Interval(float f) : lo(f), hi(f) {}

    bool has_nan() const { return std::isnan(lo) || std::isnan(hi); }
    bool contains(float v) const { return v >= lo && v <= hi; }
    float width() const { return hi - lo; }
};
struct IC {
    Interval v;
    Choice c;
};
static inline Interval I_nan() { return Interval(NANF); }

// interval.rs:68-78
static inline Interval i_abs(Interval a) {
    if (a.lo < 0.0f) {
        if (a.hi > 0.0f) return Interval(0.0f, rmax(a.hi, -a.lo));
        return Interval(-a.hi, -a.lo);
    }
    return a;
}
// interval.rs:84-94 (powi(2) == x*x)
static inline Interval i_square(Interval a) {
    if (a.hi < 0.0f) return Interval(a.hi * a.hi, a.lo * a.lo);
    if (a.lo > 0.0f) return Interval(a.lo * a.lo, a.hi * a.hi);
    if (a.has_nan()) return I_nan();
    float m = rmax(fabsf(a.lo), fabsf(a.hi));
    return Interval(0.0f, m * m);
}
// interval.rs:97-105; `as u8` is a saturating cast (NaN -> 0)
static inline int i_quadrant(float angle) {
    float q = rem_euclid(floorf(angle * 2.0f / PI_F), 4.0f);
    int u;
    if (std::isnan(q)) u = 0;
    else if (q <= 0.0f) u = 0;
    else if (q >= 255.0f) u = 255;
    else u = (int)q;
    return u;  // 0..3 in practice
}
// interval.rs:115-132
static inline Interval i_compare(Interval l, Interval r) {
    if (l.has_nan() || r.has_nan()) return I_nan();
    if (l.hi < r.lo) return Interval(-1.0f);
    if (l.lo > r.hi) return Interval(1.0f);
    if (l.lo == l.hi && r.lo == r.hi && l.lo == r.lo) return Interval(0.0f, 0.0f);
    return Interval(-1.0f, 1.0f);
}
// interval.rs:136-185
static inline Interval i_sin(Interval a) {
    if (a.has_nan()) return I_nan();
    if (a.width() >= TAU_F) return Interval(-1.0f, 1.0f);
    if (a.lo == a.hi) return Interval(sinf(a.lo));
    int lq = i_quadrant(a.lo), uq = i_quadrant(a.hi);
    float d = a.width();
    if (lq == uq && d >= PI_F) return Interval(-1.0f, 1.0f);
    if ((lq == 1 && uq == 1) || (lq == 2 && uq == 2)) return Interval(sinf(a.hi), sinf(a.lo));
    if ((lq == 0 && uq == 0) || (lq == 3 && uq == 3)) return Interval(sinf(a.lo), sinf(a.hi));
    if (lq == 3 && uq == 0) {
        if (d >= PI_F) return Interval(-1.0f, 1.0f);
        return Interval(sinf(a.lo), sinf(a.hi));
    }
    if (lq == 1 && uq == 2) {
        if (d >= PI_F) return Interval(-1.0f, 1.0f);
        return Interval(sinf(a.hi), sinf(a.lo));
    }
    if ((lq == 0 || lq == 3) && (uq == 1 || uq == 2)) return Interval(rmin(sinf(a.lo), sinf(a.hi)), 1.0f);
    if ((lq == 1 || lq == 2) && (uq == 3 || uq == 0)) return Interval(-1.0f, rmax(sinf(a.lo), sinf(a.hi)));
    return Interval(-1.0f, 1.0f);  // (Q0,Q3) | (Q2,Q1)
}
// interval.rs:188-235
static inline Interval i_cos(Interval a) {
    if (a.has_nan()) return I_nan();
    if (a.width() >= TAU_F) return Interval(-1.0f, 1.0f);
    if (a.lo == a.hi) return Interval(cosf(a.lo));
    int lq = i_quadrant(a.lo), uq = i_quadrant(a.hi);
    float d = a.width();
    if (lq == uq && d >= PI_F) return Interval(-1.0f, 1.0f);
    if ((lq == 2 && uq == 2) || (lq == 3 && uq == 3)) return Interval(cosf(a.lo), cosf(a.hi));
    if ((lq == 0 && uq == 0) || (lq == 1 && uq == 1)) return Interval(cosf(a.hi), cosf(a.lo));
    if (lq == 2 && uq == 3) {
        if (d >= PI_F) return Interval(-1.0f, 1.0f);
        return Interval(cosf(a.lo), cosf(a.hi));
    }
    if (lq == 0 && uq == 1) {
        if (d >= PI_F) return Interval(-1.0f, 1.0f);
        return Interval(cosf(a.hi), cosf(a.lo));
    }
    if ((lq == 2 || lq == 3) && (uq == 0 || uq == 1)) return Interval(rmin(cosf(a.lo), cosf(a.hi)), 1.0f);
    if ((lq == 0 || lq == 1) && (uq == 2 || uq == 3)) return Interval(-1.0f, rmax(cosf(a.lo), cosf(a.hi)));
    return Interval(-1.0f, 1.0f);  // (Q3,Q2) | (Q1,Q0)
}
// interval.rs:240-255
static inline Interval i_tan(Interval a) {
    float size = a.hi - a.lo;
    if (size >= PI_F) return I_nan();
    if (a.lo == a.hi) return Interval(tanf(a.lo));
    float l = tanf(a.lo), u = tanf(a.hi);
    if (u >= l) return Interval(l, u);
    return I_nan();
}
// interval.rs:260-281
static inline Interval i_asin(Interval a) {
    if (a.lo < -1.0f || a.hi > 1.0f) return I_nan();
    if (a.lo == a.hi) return Interval(asinf(a.lo));
    return Interval(asinf(a.lo), asinf(a.hi));
}
static inline Interval i_acos(Interval a) {
    if (a.lo < -1.0f || a.hi > 1.0f) return I_nan();
    if (a.lo == a.hi) return Interval(acosf(a.lo));
    return Interval(acosf(a.hi), acosf(a.lo));
}
// interval.rs:284-302
static inline Interval i_atan(Interval a) { return Interval(atanf(a.lo), atanf(a.hi)); }
static inline Interval i_exp(Interval a) { return Interval(expf(a.lo), expf(a.hi)); }
static inline Interval i_ln(Interval a) {
    if (a.lo <= 0.0f) return I_nan();
    return Interval(logf(a.lo), logf(a.hi));
}
// interval.rs:307-324
static inline Interval i_sqrt(Interval a) {
    if (a.lo < 0.0f) return I_nan();
    return Interval(sqrtf(a.lo), sqrtf(a.hi));
}
static inline Interval i_recip(Interval a) {
    if (a.lo > 0.0f || a.hi < 0.0f) return Interval(1.0f / a.hi, 1.0f / a.lo);
    return I_nan();
}
// interval.rs:332-370
static inline IC i_min_choice(Interval a, Interval b) {
    if (a.has_nan() || b.has_nan()) return {I_nan(), BOTH};
    Choice c = (a.hi < b.lo) ? LEFT : (b.hi < a.lo) ? RIGHT : BOTH;
    return {Interval(rmin(a.lo, b.lo), rmin(a.hi, b.hi)), c};
}
static inline IC i_max_choice(Interval a, Interval b) {
    if (a.has_nan() || b.has_nan()) return {I_nan(), BOTH};
    Choice c = (a.lo > b.hi) ? LEFT : (b.lo > a.hi) ? RIGHT : BOTH;
    return {Interval(rmax(a.lo, b.lo), rmax(a.hi, b.hi)), c};
}
// interval.rs:378-418
static inline IC i_and_choice(Interval a, Interval b) {
    if (a.has_nan() || b.has_nan()) return {I_nan(), BOTH};
    if (a.lo == 0.0f && a.hi == 0.0f) return {Interval(0.0f), LEFT};
    if (!a.contains(0.0f)) return {b, RIGHT};
    return {Interval(rmin(b.lo, 0.0f), rmax(b.hi, 0.0f)), BOTH};
}
static inline IC i_or_choice(Interval a, Interval b) {
    if (a.has_nan() || b.has_nan()) return {I_nan(), BOTH};
    if (!a.contains(0.0f)) return {a, LEFT};
    if (a.lo == 0.0f && a.hi == 0.0f) return {b, RIGHT};
    return {Interval(rmin(a.lo, b.lo), rmax(a.hi, b.hi)), BOTH};
}
// interval.rs:485-503
static inline Interval i_rem_euclid(Interval a, Interval o) {
    if (a.has_nan() || o.has_nan() || o.contains(0.0f)) return I_nan();
    if (o.lo == o.hi && o.lo > 0.0f) {
        float x = a.lo / o.lo, y = a.hi / o.lo;
        if (x != floorf(x) && floorf(x) == floorf(y)) {
            return Interval(rem_euclid(a.lo, o.lo), rem_euclid(a.hi, o.lo));
        }
        return Interval(0.0f, i_abs(o).hi);
    }
    return Interval(0.0f, i_abs(o).hi);
}
// interval.rs:507-537
static inline Interval i_floor(Interval a) { return Interval(floorf(a.lo), floorf(a.hi)); }
static inline Interval i_ceil(Interval a) { return Interval(ceilf(a.lo), ceilf(a.hi)); }
static inline Interval i_round(Interval a) { return Interval(roundf(a.lo), roundf(a.hi)); }
static inline Interval i_not(Interval a) {
    if (!a.contains(0.0f) && !a.has_nan()) return Interval(0.0f, 0.0f);
    if (a.lo == 0.0f && a.hi == 0.0f) return Interval(1.0f, 1.0f);
    return Interval(0.0f, 1.0f);
}
// interval.rs:541-597 (self = y)
static inline Interval i_atan2(Interval y, Interval x) {
    if (y.has_nan() || x.has_nan()) return I_nan();
    if (y.lo <= 0.0f && y.hi >= 0.0f && x.lo < 0.0f) return Interval(-PI_F, PI_F);
    float lower = INFINITY, upper = -INFINITY;
    auto update = [&](float yy, float xx) {
        float v = atan2f(yy, xx);
        lower = rmin(lower, v);
        upper = rmax(upper, v);
    };
    if (y.lo >= 0.0f) {
        if (x.lo >= 0.0f) {
            update(y.hi, x.lo);
            update(y.lo, x.hi);
        } else if (x.hi <= 0.0f) {
            update(y.lo, x.lo);
            update(y.hi, x.hi);
        } else {
            update(y.lo, x.lo);
            update(y.lo, x.hi);
        }
    } else if (y.hi <= 0.0f) {
        if (x.lo >= 0.0f) {
            update(y.lo, x.lo);
            update(y.hi, x.hi);
        } else if (x.hi <= 0.0f) {
            update(y.hi, x.lo);
            update(y.lo, x.hi);
        } else {
            update(y.hi, x.lo);
            update(y.hi, x.hi);
        }
    } else {
        update(y.lo, x.lo);
        update(y.hi, x.lo);
    }
    return Interval(lower, upper);
}
// interval.rs:600-627
static inline Interval i_mix(Interval a, Interval b) {
    if (a.has_nan() || b.has_nan() || f2u(a.lo) != f2u(a.hi) || f2u(b.lo) != f2u(b.hi)) return I_nan();
    return Interval(u2f(rng_mix(f2u(a.lo), f2u(b.lo))));
}
static inline Interval i_rand(Interval a) {
    if (a.has_nan() || f2u(a.lo) != f2u(a.hi)) return Interval(0.0f, 1.0f);
    return Interval(rng_rand(f2u(a.lo)));
}
// interval.rs:650-744
static inline Interval i_add(Interval a, Interval b) { return Interval(a.lo + b.lo, a.hi + b.hi); }
static inline Interval i_sub(Interval a, Interval b) { return Interval(a.lo - b.hi, a.hi - b.lo); }
static inline Interval i_neg(Interval a) { return Interval(-a.hi, -a.lo); }
static inline Interval i_mul(Interval a, Interval b) {
    if (a.has_nan() || b.has_nan()) return I_nan();
    float out[4] = {a.lo * b.lo, a.lo * b.hi, a.hi * b.lo, a.hi * b.hi};
    float lo = out[0], hi = out[0];
    for (int k = 1; k < 4; k++) {
        lo = rmin(lo, out[k]);
        hi = rmax(hi, out[k]);
    }
    return Interval(lo, hi);
}
static inline Interval i_mul_f(Interval a, float r) {
    if (a.has_nan() || std::isnan(r)) return I_nan();
    if (r < 0.0f) return Interval(a.hi * r, a.lo * r);
    return Interval(a.lo * r, a.hi * r);
}
static inline Interval i_div(Interval a, Interval b) {
    if (a.has_nan()) return I_nan();
    if (b.lo > 0.0f || b.hi < 0.0f) {
        float out[4] = {a.lo / b.lo, a.lo / b.hi, a.hi / b.lo, a.hi / b.hi};
        float lo = out[0], hi = out[0];
        for (int k = 1; k < 4; k++) {
            lo = rmin(lo, out[k]);
            hi = rmax(hi, out[k]);
        }
        return Interval(lo, hi);
    }
    return I_nan();
}

// ---------------------------------------------------------------------------
// types/grad.rs
struct Grad {
    float v, dx, dy, dz;
    Grad() : v(0), dx(0), dy(0), dz(0) {}
    Grad(float v_, float a, float b, float c) : v(v_), dx(a), dy(b), dz(c) {}
    Grad(float f) : v(f), dx(0), dy(0), dz(0) {}  // grad.rs:314-324
};
static inline Grad g_neg(Grad a) { return Grad(-a.v, -a.dx, -a.dy, -a.dz); }
static inline Grad g_abs(Grad a) { return (a.v < 0.0f) ? g_neg(a) : a; }
static inline Grad g_sqrt(Grad a) {
    float v = sqrtf(a.v);
    return Grad(v, a.dx / (2.0f * v), a.dy / (2.0f * v), a.dz / (2.0f * v));
}
static inline Grad g_sin(Grad a) {
    float c = cosf(a.v);
    return Grad(sinf(a.v), a.dx * c, a.dy * c, a.dz * c);
}
static inline Grad g_cos(Grad a) {
    float s = -sinf(a.v);
    return Grad(cosf(a.v), a.dx * s, a.dy * s, a.dz * s);
}
static inline Grad g_tan(Grad a) {
    float c0 = cosf(a.v);
    float c = c0 * c0;
    return Grad(tanf(a.v), a.dx / c, a.dy / c, a.dz / c);
}
static inline Grad g_asin(Grad a) {
    float r = sqrtf(1.0f - a.v * a.v);
    return Grad(asinf(a.v), a.dx / r, a.dy / r, a.dz / r);
}
static inline Grad g_acos(Grad a) {
    float r = sqrtf(1.0f - a.v * a.v);
    return Grad(acosf(a.v), -a.dx / r, -a.dy / r, -a.dz / r);
}
static inline Grad g_atan(Grad a) {
    float r = a.v * a.v + 1.0f;
    return Grad(atanf(a.v), a.dx / r, a.dy / r, a.dz / r);
}
static inline Grad g_exp(Grad a) {
    float v = expf(a.v);
    return Grad(v, v * a.dx, v * a.dy, v * a.dz);
}
static inline Grad g_ln(Grad a) { return Grad(logf(a.v), a.dx / a.v, a.dy / a.v, a.dz / a.v); }
static inline Grad g_min(Grad a, Grad b) {
    if (std::isnan(a.v) || std::isnan(b.v)) return Grad(NANF);
    return (a.v < b.v) ? a : b;
}
static inline Grad g_max(Grad a, Grad b) {
    if (std::isnan(a.v) || std::isnan(b.v)) return Grad(NANF);
    return (a.v > b.v) ? a : b;
}
static inline Grad g_rem_euclid(Grad a, Grad r) {
    float e = div_euclid(a.v, r.v);
    return Grad(rem_euclid(a.v, r.v), a.dx - r.dx * e, a.dy - r.dy * e, a.dz - r.dz * e);
}
static inline Grad g_and(Grad a, Grad b) { return (a.v == 0.0f) ? a : b; }
static inline Grad g_or(Grad a, Grad b) { return (a.v != 0.0f) ? a : b; }
static inline Grad g_floor(Grad a) { return Grad(floorf(a.v), 0, 0, 0); }
static inline Grad g_ceil(Grad a) { return Grad(ceilf(a.v), 0, 0, 0); }
static inline Grad g_round(Grad a) { return Grad(roundf(a.v), 0, 0, 0); }
static inline Grad g_atan2(Grad y, Grad x) {
    float d = x.v * x.v + y.v * y.v;
    return Grad(atan2f(y.v, x.v), (x.v * y.dx - y.v * x.dx) / d, (x.v * y.dy - y.v * x.dy) / d,
                (x.v * y.dz - y.v * x.dz) / d);
}
static inline Grad g_compare(Grad a, Grad b) { return Grad(f_compare(a.v, b.v)); }
static inline Grad g_not(Grad a) { return Grad((a.v == 0.0f) ? 1.0f : 0.0f); }
static inline Grad g_rand(Grad a) { return Grad(rng_rand(f2u(a.v))); }
static inline Grad g_mix(Grad a, Grad b) { return Grad(u2f(rng_mix(f2u(a.v), f2u(b.v)))); }
static inline Grad g_add(Grad a, Grad b) { return Grad(a.v + b.v, a.dx + b.dx, a.dy + b.dy, a.dz + b.dz); }
static inline Grad g_sub(Grad a, Grad b) { return Grad(a.v - b.v, a.dx - b.dx, a.dy - b.dy, a.dz - b.dz); }
static inline Grad g_mul(Grad a, Grad b) {
    return Grad(a.v * b.v, a.v * b.dx + b.v * a.dx, a.v * b.dy + b.v * a.dy, a.v * b.dz + b.v * a.dz);
}
static inline Grad g_mul_f(Grad a, float r) { return Grad(a.v * r, a.dx * r, a.dy * r, a.dz * r); }
static inline Grad g_div(Grad a, Grad b) {
    float d = b.v * b.v;
    return Grad(a.v / b.v, (b.v * a.dx - a.v * b.dx) / d, (b.v * a.dy - a.v * b.dy) / d,
                (b.v * a.dz - a.v * b.dz) / d);
}

}  // namespace orc
