// Device-side op semantics for the three value domains (f32, Interval, Grad).
//
// Every function states the reference definition it reproduces
// (fidget-core/src/types/{float,interval,grad}.rs, rng/mod.rs).  Compiled with
// -ffp-contract=off: the reference (Rust) never fuses a*b+c, and pruning
// decisions are f32 comparisons on these results, so they must be IEEE
// round-to-nearest op by op.  `/` and sqrtf are the correctly rounded forms
// (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt); denormals are kept.
//
// Transcendentals: the reference calls the platform libm through Rust's std and
// its tests compare with those calls exactly; trans_libm.hpp restates that libm's
// (glibc 2.35, x86-64 FMA variants) f32 routines operation by operation, so the
// device returns the host's bits (tools/libm_sweep.cpp, tests/test_gpu_math.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tape_format.h"
#include "trans_libm.hpp"

#define FH_DEV __device__ __forceinline__

namespace fhd {

FH_DEV uint32_t f2u(float f) { return __float_as_uint(f); }
FH_DEV float u2f(uint32_t u) { return __uint_as_float(u); }
FH_DEV float qnan() { return u2f(0x7fc00000u); }
FH_DEV bool isnan_(float f) { return f != f; }

// Rust f32::min / max == IEEE minNum / maxNum
FH_DEV float rmin(float a, float b) { return fminf(a, b); }
FH_DEV float rmax(float a, float b) { return fmaxf(a, b); }

// ---- transcendental f32: the host libm's routines (trans_libm.hpp) ---------------
FH_DEV float t_sin(float a) { return fhlm::sincosf_<fhlm::MemTables, false>(a); }
FH_DEV float t_cos(float a) { return fhlm::sincosf_<fhlm::MemTables, true>(a); }
FH_DEV float t_tan(float a) { return fhlm::tanf_<fhlm::MemTables>(a); }
FH_DEV float t_asin(float a) { return fhlm::asinf_(a); }
FH_DEV float t_acos(float a) { return fhlm::acosf_(a); }
FH_DEV float t_atan(float a) { return fhlm::atanf_(a); }
FH_DEV float t_exp(float a) { return fhlm::expf_<fhlm::MemTables>(a); }
FH_DEV float t_ln(float a) { return fhlm::logf_<fhlm::MemTables>(a); }
FH_DEV float t_atan2(float y, float x) { return fhlm::atan2f_(y, x); }

// ---- rng (rng/mod.rs:8-33) ---------------------------------------------------------
FH_DEV uint32_t pcg(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}
FH_DEV float f_rand(float a) { return u2f((pcg(f2u(a)) >> 9) | 0x3f800000u) - 1.0f; }
FH_DEV float f_mix(float a, float b) { return u2f(pcg(f2u(a) + pcg(f2u(b)))); }

// f32::rem_euclid / div_euclid (core): r = a % b; r < 0 ? r + |b| : r
FH_DEV float rem_euclid(float a, float b) {
    float r = fmodf(a, b);
    return r < 0.0f ? r + fabsf(b) : r;
}
FH_DEV float div_euclid(float a, float b) {
    float q = truncf(a / b);
    if (fmodf(a, b) < 0.0f) return b > 0.0f ? q - 1.0f : q + 1.0f;
    return q;
}

// =====================================================================================
// f32 (types/float.rs:66-142)
// =====================================================================================
FH_DEV float f_compare(float a, float b) { return a < b ? -1.0f : (a == b ? 0.0f : (a > b ? 1.0f : qnan())); }
FH_DEV float f_min(float a, float b, int& c) {
    if (a < b) { c = FH_CHOICE_LEFT; return a; }
    if (b < a) { c = FH_CHOICE_RIGHT; return b; }
    c = FH_CHOICE_BOTH;
    return (isnan_(a) || isnan_(b)) ? qnan() : b;
}
FH_DEV float f_max(float a, float b, int& c) {
    if (a > b) { c = FH_CHOICE_LEFT; return a; }
    if (b > a) { c = FH_CHOICE_RIGHT; return b; }
    c = FH_CHOICE_BOTH;
    return (isnan_(a) || isnan_(b)) ? qnan() : b;
}
FH_DEV float f_and(float a, float b, int& c) {
    if (a == 0.0f) { c = FH_CHOICE_LEFT; return a; }
    c = FH_CHOICE_RIGHT;
    return b;
}
FH_DEV float f_or(float a, float b, int& c) {
    if (a != 0.0f) { c = FH_CHOICE_LEFT; return a; }
    c = FH_CHOICE_RIGHT;
    return b;
}

struct F32 {
    typedef float V;
    static FH_DEV V imm(float f) { return f; }
    // FULL = false drops the transcendental / modulo cases (and their f64 code and register
    // pressure) from kernels built for tapes that contain none of them.
    template <bool FULL>
    static FH_DEV V unary(int op, V a) {
        switch (op) {
            case FH_NEG: return -a;
            case FH_ABS: return fabsf(a);
            case FH_RECIP: return 1.0f / a;
            case FH_SQRT: return sqrtf(a);
            case FH_SQUARE: return a * a;
            case FH_FLOOR: return floorf(a);
            case FH_CEIL: return ceilf(a);
            case FH_ROUND: return roundf(a);
            case FH_NOT: return a == 0.0f ? 1.0f : 0.0f;
            case FH_RAND: return f_rand(a);
            default: break;
        }
        if constexpr (FULL) {
            switch (op) {
                case FH_SIN: return t_sin(a);
                case FH_COS: return t_cos(a);
                case FH_TAN: return t_tan(a);
                case FH_ASIN: return t_asin(a);
                case FH_ACOS: return t_acos(a);
                case FH_ATAN: return t_atan(a);
                case FH_EXP: return t_exp(a);
                default: return t_ln(a);
            }
        }
        return a;
    }
    // base = op - FH_ADD_RR (0 add .. 11 or)
    template <bool FULL>
    static FH_DEV V binary(int base, V a, V b, int& c) {
        if constexpr (FULL) {
            if (base == 4) return t_atan2(a, b);
            if (base == 7) return rem_euclid(a, b);
        }
        switch (base) {
            case 0: return a + b;
            case 1: return a - b;
            case 2: return a * b;
            case 3: return a / b;
            case 5: return f_compare(a, b);
            case 6: return f_mix(a, b);
            case 8: return f_min(a, b, c);
            case 9: return f_max(a, b, c);
            case 10: return f_and(a, b, c);
            default: return f_or(a, b, c);
        }
    }
    static FH_DEV V mul_imm(V a, float imm) { return a * imm; }
};

// =====================================================================================
// Interval (types/interval.rs).  {lo, hi}; NaN interval = {NaN, NaN}
// =====================================================================================
struct IV {
    float lo, hi;
};
FH_DEV IV iv(float lo, float hi) { IV r; r.lo = lo; r.hi = hi; return r; }
FH_DEV IV iv1(float f) { return iv(f, f); }
FH_DEV IV iv_nan() { return iv(qnan(), qnan()); }
FH_DEV bool iv_has_nan(IV a) { return isnan_(a.lo) || isnan_(a.hi); }
FH_DEV bool iv_contains(IV a, float v) { return v >= a.lo && v <= a.hi; }

FH_DEV IV iv_neg(IV a) { return iv(-a.hi, -a.lo); }                         // 737-744
FH_DEV IV iv_abs(IV a) {                                                    // 68-78
    if (a.lo < 0.0f) {
        if (a.hi > 0.0f) return iv(0.0f, rmax(a.hi, -a.lo));
        return iv(-a.hi, -a.lo);
    }
    return a;
}
FH_DEV IV iv_square(IV a) {                                                 // 84-94
    if (a.hi < 0.0f) return iv(a.hi * a.hi, a.lo * a.lo);
    if (a.lo > 0.0f) return iv(a.lo * a.lo, a.hi * a.hi);
    if (iv_has_nan(a)) return iv_nan();
    float m = rmax(fabsf(a.lo), fabsf(a.hi));
    return iv(0.0f, m * m);
}
FH_DEV IV iv_sqrt(IV a) { return a.lo < 0.0f ? iv_nan() : iv(sqrtf(a.lo), sqrtf(a.hi)); }          // 307-313
FH_DEV IV iv_recip(IV a) { return (a.lo > 0.0f || a.hi < 0.0f) ? iv(1.0f / a.hi, 1.0f / a.lo) : iv_nan(); }  // 318-324
FH_DEV IV iv_add(IV a, IV b) { return iv(a.lo + b.lo, a.hi + b.hi); }       // 650-656
FH_DEV IV iv_sub(IV a, IV b) { return iv(a.lo - b.hi, a.hi - b.lo); }       // 728-735
FH_DEV IV iv_mul(IV a, IV b) {                                              // 658-681
    if (iv_has_nan(a) || iv_has_nan(b)) return iv_nan();
    float p0 = a.lo * b.lo, p1 = a.lo * b.hi, p2 = a.hi * b.lo, p3 = a.hi * b.hi;
    return iv(rmin(rmin(rmin(p0, p1), p2), p3), rmax(rmax(rmax(p0, p1), p2), p3));
}
FH_DEV IV iv_mul_f(IV a, float r) {                                         // 683-696
    if (iv_has_nan(a) || isnan_(r)) return iv_nan();
    return r < 0.0f ? iv(a.hi * r, a.lo * r) : iv(a.lo * r, a.hi * r);
}
FH_DEV IV iv_div(IV a, IV b) {                                              // 698-726
    if (iv_has_nan(a)) return iv_nan();
    if (b.lo > 0.0f || b.hi < 0.0f) {
        float q0 = a.lo / b.lo, q1 = a.lo / b.hi, q2 = a.hi / b.lo, q3 = a.hi / b.hi;
        return iv(rmin(rmin(rmin(q0, q1), q2), q3), rmax(rmax(rmax(q0, q1), q2), q3));
    }
    return iv_nan();
}
FH_DEV int iv_quadrant(float angle) {                                       // 97-105
    const float PI = 3.14159274101257324f;
    float q = rem_euclid(floorf(angle * 2.0f / PI), 4.0f);
    if (!(q > 0.0f)) return 0;  // NaN and <= 0 saturate to 0 (`as u8`)
    return q >= 255.0f ? 255 : (int)q;
}
template <bool IS_SIN>
FH_DEV IV iv_sincos(IV a) {                                                 // 136-235
    const float PI = 3.14159274101257324f, TAU = 6.28318548202514648f;
    if (iv_has_nan(a)) return iv_nan();
    float d = a.hi - a.lo;
    if (d >= TAU) return iv(-1.0f, 1.0f);
    if (a.lo == a.hi) return iv1(IS_SIN ? t_sin(a.lo) : t_cos(a.lo));
    int lq = iv_quadrant(a.lo), uq = iv_quadrant(a.hi);
    // cos(x) = sin(x + pi/2): the reference's cos table is the sin table with
    // quadrants rotated by one, applied to cos of the bounds
    if (!IS_SIN) { lq = (lq + 1) & 3; uq = (uq + 1) & 3; }
    float fl = IS_SIN ? t_sin(a.lo) : t_cos(a.lo);
    float fu = IS_SIN ? t_sin(a.hi) : t_cos(a.hi);
    if (lq == uq) {
        if (d >= PI) return iv(-1.0f, 1.0f);
        if (lq == 1 || lq == 2) return iv(fu, fl);  // decreasing
        return iv(fl, fu);                          // increasing
    }
    if (lq == 3 && uq == 0) return d >= PI ? iv(-1.0f, 1.0f) : iv(fl, fu);
    if (lq == 1 && uq == 2) return d >= PI ? iv(-1.0f, 1.0f) : iv(fu, fl);
    if ((lq == 0 || lq == 3) && (uq == 1 || uq == 2)) return iv(rmin(fl, fu), 1.0f);
    if ((lq == 1 || lq == 2) && (uq == 3 || uq == 0)) return iv(-1.0f, rmax(fl, fu));
    return iv(-1.0f, 1.0f);  // (Q0,Q3) | (Q2,Q1)
}
FH_DEV IV iv_tan(IV a) {                                                    // 240-255
    const float PI = 3.14159274101257324f;
    if ((a.hi - a.lo) >= PI) return iv_nan();
    if (a.lo == a.hi) return iv1(t_tan(a.lo));
    float l = t_tan(a.lo), u = t_tan(a.hi);
    return u >= l ? iv(l, u) : iv_nan();
}
FH_DEV IV iv_asin(IV a) {                                                   // 260-268
    if (a.lo < -1.0f || a.hi > 1.0f) return iv_nan();
    if (a.lo == a.hi) return iv1(t_asin(a.lo));
    return iv(t_asin(a.lo), t_asin(a.hi));
}
FH_DEV IV iv_acos(IV a) {                                                   // 273-281
    if (a.lo < -1.0f || a.hi > 1.0f) return iv_nan();
    if (a.lo == a.hi) return iv1(t_acos(a.lo));
    return iv(t_acos(a.hi), t_acos(a.lo));
}
FH_DEV IV iv_atan(IV a) { return iv(t_atan(a.lo), t_atan(a.hi)); }          // 284-286
FH_DEV IV iv_exp(IV a) { return iv(t_exp(a.lo), t_exp(a.hi)); }             // 289-291
FH_DEV IV iv_ln(IV a) { return a.lo <= 0.0f ? iv_nan() : iv(t_ln(a.lo), t_ln(a.hi)); }  // 296-302
FH_DEV IV iv_floor(IV a) { return iv(floorf(a.lo), floorf(a.hi)); }         // 507-521
FH_DEV IV iv_ceil(IV a) { return iv(ceilf(a.lo), ceilf(a.hi)); }
FH_DEV IV iv_round(IV a) { return iv(roundf(a.lo), roundf(a.hi)); }
FH_DEV IV iv_not(IV a) {                                                    // 529-537
    if (!iv_contains(a, 0.0f) && !iv_has_nan(a)) return iv(0.0f, 0.0f);
    if (a.lo == 0.0f && a.hi == 0.0f) return iv(1.0f, 1.0f);
    return iv(0.0f, 1.0f);
}
FH_DEV IV iv_rand(IV a) {                                                   // 619-627
    if (iv_has_nan(a) || f2u(a.lo) != f2u(a.hi)) return iv(0.0f, 1.0f);
    return iv1(f_rand(a.lo));
}
FH_DEV IV iv_mix(IV a, IV b) {                                              // 600-616
    if (iv_has_nan(a) || iv_has_nan(b) || f2u(a.lo) != f2u(a.hi) || f2u(b.lo) != f2u(b.hi)) return iv_nan();
    return iv1(f_mix(a.lo, b.lo));
}
FH_DEV IV iv_compare(IV l, IV r) {                                          // 115-132
    if (iv_has_nan(l) || iv_has_nan(r)) return iv_nan();
    if (l.hi < r.lo) return iv1(-1.0f);
    if (l.lo > r.hi) return iv1(1.0f);
    if (l.lo == l.hi && r.lo == r.hi && l.lo == r.lo) return iv(0.0f, 0.0f);
    return iv(-1.0f, 1.0f);
}
FH_DEV IV iv_rem_euclid(IV a, IV o) {                                       // 485-503
    if (iv_has_nan(a) || iv_has_nan(o) || iv_contains(o, 0.0f)) return iv_nan();
    if (o.lo == o.hi && o.lo > 0.0f) {
        float x = a.lo / o.lo, y = a.hi / o.lo;
        if (x != floorf(x) && floorf(x) == floorf(y)) return iv(rem_euclid(a.lo, o.lo), rem_euclid(a.hi, o.lo));
    }
    return iv(0.0f, iv_abs(o).hi);
}
FH_DEV IV iv_atan2(IV y, IV x) {                                            // 541-597
    const float PI = 3.14159274101257324f;
    if (iv_has_nan(y) || iv_has_nan(x)) return iv_nan();
    if (y.lo <= 0.0f && y.hi >= 0.0f && x.lo < 0.0f) return iv(-PI, PI);
    float y0, x0, y1, x1;
    if (y.lo >= 0.0f) {
        if (x.lo >= 0.0f) { y0 = y.hi; x0 = x.lo; y1 = y.lo; x1 = x.hi; }
        else if (x.hi <= 0.0f) { y0 = y.lo; x0 = x.lo; y1 = y.hi; x1 = x.hi; }
        else { y0 = y.lo; x0 = x.lo; y1 = y.lo; x1 = x.hi; }
    } else if (y.hi <= 0.0f) {
        if (x.lo >= 0.0f) { y0 = y.lo; x0 = x.lo; y1 = y.hi; x1 = x.hi; }
        else if (x.hi <= 0.0f) { y0 = y.hi; x0 = x.lo; y1 = y.lo; x1 = x.hi; }
        else { y0 = y.hi; x0 = x.lo; y1 = y.hi; x1 = x.hi; }
    } else { y0 = y.lo; x0 = x.lo; y1 = y.hi; x1 = x.lo; }
    float v0 = t_atan2(y0, x0), v1 = t_atan2(y1, x1);
    const float INF = u2f(0x7f800000u);
    return iv(rmin(rmin(INF, v0), v1), rmax(rmax(-INF, v0), v1));
}
FH_DEV IV iv_min(IV a, IV b, int& c) {                                      // 332-347
    c = FH_CHOICE_BOTH;
    if (iv_has_nan(a) || iv_has_nan(b)) return iv_nan();
    c = (a.hi < b.lo) ? FH_CHOICE_LEFT : ((b.hi < a.lo) ? FH_CHOICE_RIGHT : FH_CHOICE_BOTH);
    return iv(rmin(a.lo, b.lo), rmin(a.hi, b.hi));
}
FH_DEV IV iv_max(IV a, IV b, int& c) {                                      // 355-370
    c = FH_CHOICE_BOTH;
    if (iv_has_nan(a) || iv_has_nan(b)) return iv_nan();
    c = (a.lo > b.hi) ? FH_CHOICE_LEFT : ((b.lo > a.hi) ? FH_CHOICE_RIGHT : FH_CHOICE_BOTH);
    return iv(rmax(a.lo, b.lo), rmax(a.hi, b.hi));
}
FH_DEV IV iv_and(IV a, IV b, int& c) {                                      // 378-393
    c = FH_CHOICE_BOTH;
    if (iv_has_nan(a) || iv_has_nan(b)) return iv_nan();
    if (a.lo == 0.0f && a.hi == 0.0f) { c = FH_CHOICE_LEFT; return iv1(0.0f); }
    if (!iv_contains(a, 0.0f)) { c = FH_CHOICE_RIGHT; return b; }
    return iv(rmin(b.lo, 0.0f), rmax(b.hi, 0.0f));
}
FH_DEV IV iv_or(IV a, IV b, int& c) {                                       // 401-418
    c = FH_CHOICE_BOTH;
    if (iv_has_nan(a) || iv_has_nan(b)) return iv_nan();
    if (!iv_contains(a, 0.0f)) { c = FH_CHOICE_LEFT; return a; }
    if (a.lo == 0.0f && a.hi == 0.0f) { c = FH_CHOICE_RIGHT; return b; }
    return iv(rmin(a.lo, b.lo), rmax(a.hi, b.hi));
}

struct IVAL {
    typedef IV V;
    static FH_DEV V imm(float f) { return iv1(f); }
    template <bool FULL>
    static FH_DEV V unary(int op, V a) {
        switch (op) {
            case FH_NEG: return iv_neg(a);
            case FH_ABS: return iv_abs(a);
            case FH_RECIP: return iv_recip(a);
            case FH_SQRT: return iv_sqrt(a);
            case FH_SQUARE: return iv_square(a);
            case FH_FLOOR: return iv_floor(a);
            case FH_CEIL: return iv_ceil(a);
            case FH_ROUND: return iv_round(a);
            case FH_NOT: return iv_not(a);
            case FH_RAND: return iv_rand(a);
            default: break;
        }
        if constexpr (FULL) {
            switch (op) {
                case FH_SIN: return iv_sincos<true>(a);
                case FH_COS: return iv_sincos<false>(a);
                case FH_TAN: return iv_tan(a);
                case FH_ASIN: return iv_asin(a);
                case FH_ACOS: return iv_acos(a);
                case FH_ATAN: return iv_atan(a);
                case FH_EXP: return iv_exp(a);
                default: return iv_ln(a);
            }
        }
        return a;
    }
    template <bool FULL>
    static FH_DEV V binary(int base, V a, V b, int& c) {
        if constexpr (FULL) {
            if (base == 4) return iv_atan2(a, b);
            if (base == 7) return iv_rem_euclid(a, b);
        }
        switch (base) {
            case 0: return iv_add(a, b);
            case 1: return iv_sub(a, b);
            case 2: return iv_mul(a, b);
            case 3: return iv_div(a, b);
            case 5: return iv_compare(a, b);
            case 6: return iv_mix(a, b);
            case 8: return iv_min(a, b, c);
            case 9: return iv_max(a, b, c);
            case 10: return iv_and(a, b, c);
            default: return iv_or(a, b, c);
        }
    }
    static FH_DEV V mul_imm(V a, float imm) { return iv_mul_f(a, imm); }  // vm/mod.rs:410-412
};

// =====================================================================================
// Grad (types/grad.rs): {v, dx, dy, dz}
// =====================================================================================
struct GR {
    float v, dx, dy, dz;
};
FH_DEV GR gr(float v, float a, float b, float c) { GR r; r.v = v; r.dx = a; r.dy = b; r.dz = c; return r; }
FH_DEV GR gr1(float v) { return gr(v, 0.0f, 0.0f, 0.0f); }
FH_DEV GR gr_neg(GR a) { return gr(-a.v, -a.dx, -a.dy, -a.dz); }
FH_DEV GR gr_add(GR a, GR b) { return gr(a.v + b.v, a.dx + b.dx, a.dy + b.dy, a.dz + b.dz); }
FH_DEV GR gr_sub(GR a, GR b) { return gr(a.v - b.v, a.dx - b.dx, a.dy - b.dy, a.dz - b.dz); }
FH_DEV GR gr_mul(GR a, GR b) {                                              // 347-359
    return gr(a.v * b.v, a.v * b.dx + b.v * a.dx, a.v * b.dy + b.v * a.dy, a.v * b.dz + b.v * a.dz);
}
FH_DEV GR gr_mul_f(GR a, float r) { return gr(a.v * r, a.dx * r, a.dy * r, a.dz * r); }   // 361-373
FH_DEV GR gr_div(GR a, GR b) {                                              // 375-388
    float d = b.v * b.v;
    return gr(a.v / b.v, (b.v * a.dx - a.v * b.dx) / d, (b.v * a.dy - a.v * b.dy) / d, (b.v * a.dz - a.v * b.dz) / d);
}
FH_DEV GR gr_scale_div(GR a, float v, float r) { return gr(v, a.dx / r, a.dy / r, a.dz / r); }

struct GRAD {
    typedef GR V;
    static FH_DEV V imm(float f) { return gr1(f); }
    template <bool FULL>
    static FH_DEV V unary(int op, V a) {
        if constexpr (!FULL) {
            if (op >= FH_SIN && op <= FH_LN) return a;
        }
        switch (op) {
            case FH_NEG: return gr_neg(a);
            case FH_ABS: return a.v < 0.0f ? gr_neg(a) : a;                 // 44-55
            case FH_RECIP: return gr_div(gr1(1.0f), a);                     // vm/mod.rs:1127-1132
            case FH_SQRT: { float v = sqrtf(a.v); return gr(v, a.dx / (2.0f * v), a.dy / (2.0f * v), a.dz / (2.0f * v)); }
            case FH_SQUARE: return gr_mul(a, a);                            // vm/mod.rs:1138-1143
            case FH_FLOOR: return gr1(floorf(a.v));
            case FH_CEIL: return gr1(ceilf(a.v));
            case FH_ROUND: return gr1(roundf(a.v));
            case FH_SIN: { float c = t_cos(a.v); return gr(t_sin(a.v), a.dx * c, a.dy * c, a.dz * c); }
            case FH_COS: { float s = -t_sin(a.v); return gr(t_cos(a.v), a.dx * s, a.dy * s, a.dz * s); }
            case FH_TAN: { float c0 = t_cos(a.v); float c = c0 * c0; return gr(t_tan(a.v), a.dx / c, a.dy / c, a.dz / c); }
            case FH_ASIN: { float r = sqrtf(1.0f - a.v * a.v); return gr(t_asin(a.v), a.dx / r, a.dy / r, a.dz / r); }
            case FH_ACOS: { float r = sqrtf(1.0f - a.v * a.v); return gr(t_acos(a.v), -a.dx / r, -a.dy / r, -a.dz / r); }
            case FH_ATAN: { float r = a.v * a.v + 1.0f; return gr(t_atan(a.v), a.dx / r, a.dy / r, a.dz / r); }
            case FH_EXP: { float v = t_exp(a.v); return gr(v, v * a.dx, v * a.dy, v * a.dz); }
            case FH_LN: return gr(t_ln(a.v), a.dx / a.v, a.dy / a.v, a.dz / a.v);
            case FH_NOT: return gr1(a.v == 0.0f ? 1.0f : 0.0f);
            default: return gr1(f_rand(a.v));
        }
    }
    template <bool FULL>
    static FH_DEV V binary(int base, V a, V b, int& c) {
        (void)c;
        if constexpr (!FULL) {
            if (base == 4 || base == 7) return a;
        }
        switch (base) {
            case 0: return gr_add(a, b);
            case 1: return gr_sub(a, b);
            case 2: return gr_mul(a, b);
            case 3: return gr_div(a, b);
            case 4: {                                                       // atan2(y = a, x = b), 260-270
                float d = b.v * b.v + a.v * a.v;
                return gr(t_atan2(a.v, b.v), (b.v * a.dx - a.v * b.dx) / d, (b.v * a.dy - a.v * b.dy) / d,
                          (b.v * a.dz - a.v * b.dz) / d);
            }
            case 5: return gr1(f_compare(a.v, b.v));
            case 6: return gr1(f_mix(a.v, b.v));
            case 7: {                                                       // 199-207
                float e = div_euclid(a.v, b.v);
                return gr(rem_euclid(a.v, b.v), a.dx - b.dx * e, a.dy - b.dy * e, a.dz - b.dz * e);
            }
            case 8: return (isnan_(a.v) || isnan_(b.v)) ? gr1(qnan()) : (a.v < b.v ? a : b);   // 173-181
            case 9: return (isnan_(a.v) || isnan_(b.v)) ? gr1(qnan()) : (a.v > b.v ? a : b);   // 187-195
            case 10: return a.v == 0.0f ? a : b;
            default: return a.v != 0.0f ? a : b;
        }
    }
    static FH_DEV V mul_imm(V a, float imm) { return gr_mul_f(a, imm); }  // vm/mod.rs:1219-1223
};

// ---- screen -> model transform (shape/mod.rs:894-948, nalgebra transform_point) -------
struct Mat4 {
    float m[16];  // row major
};
FH_DEV void xf_point(const Mat4& t, float x, float y, float z, float& ox, float& oy, float& oz) {
    float n = ((t.m[12] * x + t.m[13] * y) + t.m[14] * z) + t.m[15];
    float a = ((t.m[0] * x + t.m[1] * y) + t.m[2] * z) + t.m[3];
    float b = ((t.m[4] * x + t.m[5] * y) + t.m[6] * z) + t.m[7];
    float c = ((t.m[8] * x + t.m[9] * y) + t.m[10] * z) + t.m[11];
    if (n != 0.0f) { a = a / n; b = b / n; c = c / n; }
    ox = a; oy = b; oz = c;
}
FH_DEV void xf_interval(const Mat4& t, IV x, IV y, IV z, IV& ox, IV& oy, IV& oz) {
    IV r[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
        r[i] = iv_add(iv_add(iv_add(iv_mul_f(x, t.m[i * 4 + 0]), iv_mul_f(y, t.m[i * 4 + 1])), iv_mul_f(z, t.m[i * 4 + 2])),
                      iv1(t.m[i * 4 + 3]));
    ox = iv_div(r[0], r[3]); oy = iv_div(r[1], r[3]); oz = iv_div(r[2], r[3]);
}
FH_DEV void xf_grad(const Mat4& t, GR x, GR y, GR z, GR& ox, GR& oy, GR& oz) {
    GR r[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
        r[i] = gr_add(gr_add(gr_add(gr_mul_f(x, t.m[i * 4 + 0]), gr_mul_f(y, t.m[i * 4 + 1])), gr_mul_f(z, t.m[i * 4 + 2])),
                      gr1(t.m[i * 4 + 3]));
    ox = gr_div(r[0], r[3]); oy = gr_div(r[1], r[3]); oz = gr_div(r[2], r[3]);
}

}  // namespace fhd
