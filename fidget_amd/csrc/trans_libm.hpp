// The host libm's f32 routines, restated operation by operation for the device (and the host, for the CPU sweep).
//
// The reference evaluates sin cos tan asin acos atan atan2 exp ln with Rust's f32 methods, which call the platform libm, and its
// own bulk tests assert EXACT equality with those calls (fidget-core/src/eval/test/float_slice.rs:404-412, canonical ops
// eval/test/mod.rs:194-203).  On the deployment image that libm is glibc 2.35 (Ubuntu 2.35-0ubuntu3.11) on x86-64 with FMA + AVX2,
// where the dynamic loader picks the `_fma` ifunc variants of sinf / cosf / expf / logf (sysdeps/x86_64/fpu/multiarch/ifunc-fma.h):
// the same C source as the generic build, compiled with -mfma, so that every `a * b + c` of the source is ONE fused operation.
// The routines below follow
//
//   sinf, cosf    sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, s_sincosf.h, s_sincosf_data.c   (Szabolcs Nagy / ARM optimized routines)
//   expf          sysdeps/ieee754/flt-32/e_expf.c, e_exp2f_data.c                            (same origin)
//   logf          sysdeps/ieee754/flt-32/e_logf.c, e_logf_data.c                             (same origin)
//   tanf          sysdeps/ieee754/flt-32/s_tanf.c, k_tanf.c, e_rem_pio2f.c, k_rem_pio2f.c   (fdlibm, float arithmetic, no FMA variant)
//   asinf, acosf  sysdeps/ieee754/flt-32/e_asinf.c, e_acosf.c                                (fdlibm)
//   atanf, atan2f sysdeps/ieee754/flt-32/s_atanf.c, e_atan2f.c                               (fdlibm)
//
// with the fused operations read off the disassembly of this image's libm.so.6 (__sinf_fma 0x7b2b0, __cosf_fma 0x7b4f0, __expf_fma
// 0x7aba0, __logf_fma 0x7add0; the fdlibm routines are plain SSE2 scalar code: one rounding per C operation).  Everything is IEEE
// binary32 / binary64 arithmetic in round-to-nearest with explicit `fma`s, so the device (v_fma_f64, v_mul_f64, IEEE f32 division and
// square root) returns the host's bits; both sides must be compiled with floating-point contraction OFF.  `tools/libm_sweep.cpp`
// compares every routine with the running libm over all 2^32 arguments on the CPU, `tools/math_sweep.py` on the device.
//
// Shape of the code: the routines are also compiled into the assembly interpreters (trans_funcs.hip, gen_trans.py), where a routine
// has ten scalar registers; a saved execution mask per nested branch would not fit.  So every routine is ONE straight path that all
// lanes run, at most one wave-uniform skip around a rare case, and selects: the library's branches are the same arithmetic on the
// branch's lanes only - evaluating a branch for lanes that do not take it changes nothing a select does not discard (no traps, no
// flags are observed).  Where a small-argument branch of the library is the general path with n = 0 (sinf, cosf, tanf below pi / 4)
// the general path is used: identical operations with exact identities (x - 0 * c, x * 1).
//
// NaN results: the value class is the host's; the payload / sign of a NaN is not modelled (x86 produces the negative default NaN for
// an invalid operation, the GPU the positive one - as for every other opcode).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FHLM __host__ __device__ inline
#else
#define FHLM inline
#endif

namespace fhlm {

FHLM uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
FHLM float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
FHLM uint64_t d2u(double f) { return __builtin_bit_cast(uint64_t, f); }
FHLM double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }
FHLM double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
FHLM float nan_() { return u2f(0x7FC00000u); }
// A value the device compiler must take as it is (an empty asm on a vector register): a comparison recomputed from it is not merged with
// an earlier one, so its lane mask does not occupy scalar registers in between (the four-sample routines of trans_funcs.hip would
// otherwise hold four such masks at once and spill them).  Nothing on the host.
FHLM uint32_t opaque_(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
    return v;
}

// ---- tables (e_exp2f_data.c: 2^(i/32) with the exponent bits of i/32 taken off; e_logf_data.c; s_sincosf_data.c __inv_pio4) ----
struct MemTables {
    FHLM static uint64_t exp2_tab(uint32_t i) {
        static const uint64_t T[32] = {
            0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b, 0x3fef54873168b9aa,
            0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
            0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74,
            0x3feea11473eb0187, 0x3feea589994cce13, 0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
            0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
            0x3fefa4afa2a490da, 0x3fefd0765b6e4540};
        return T[i];
    }
    FHLM static double log_invc(uint32_t i) {
        static const double T[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0,  0x1.3c995b0b80385p+0,
                                     0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,  0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,
                                     0x1.0953f419900a7p+0, 0x1p+0,               0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
                                     0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
        return T[i];
    }
    FHLM static double log_logc(uint32_t i) {
        static const double T[16] = {-0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3,
                                     -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3,   -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4,
                                     -0x1.252f438e10c1ep-5, 0x0p+0,                0x1.aa5aa5df25984p-5,  0x1.c5e53aa362eb4p-4,
                                     0x1.526e57720db08p-3,  0x1.bc2860d22477p-3,   0x1.1058bc8a07ee1p-2,  0x1.4043057b6ee09p-2};
        return T[i];
    }
    // 4 / pi as a bit stream: entry i holds bits [8 i - 24, 8 i + 8) of 0.a2f9836e4e44...
    FHLM static uint32_t inv_pio4(uint32_t i) {
        static const uint32_t T[24] = {0xa2,       0xa2f9,     0xa2f983,   0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529,
                                       0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0,
                                       0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};
        return T[i];
    }
};

// ---- sinf / cosf (s_sincosf.h) ----
// reduce_large: 120 <= |x| < inf; xi = the bits of x.  192 bits of 4 / pi against the 24-bit mantissa, in integers.
template <class Tab>
FHLM double sincosf_reduce_large(uint32_t xi, int* np) {
    const uint32_t i = (xi >> 26) & 15;
    const int shift = (xi >> 23) & 7;
    xi = (xi & 0xffffff) | 0x800000;
    xi <<= shift;
    uint64_t res0 = (uint32_t)(xi * Tab::inv_pio4(i));
    const uint64_t res1 = (uint64_t)xi * Tab::inv_pio4(i + 4);
    const uint64_t res2 = (uint64_t)xi * Tab::inv_pio4(i + 8);
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const uint64_t n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    const double x = (double)(int64_t)res0;
    *np = (int)n;
    return x * 0x1.921FB54442D18p-62;
}

// sinf_poly with the coefficients of __sincosf_table[0] {c0 c1 s1 c2 s2 c3 s3 c4}.  Table 1 (chosen when the quadrant has bit 1 set)
// holds the same sine and the NEGATED cosine coefficients; a chain of fused multiply-adds over negated constants returns exactly
// the negated value, so the cosine branch's sign is flipped instead of switching tables.
template <class Tab, bool IS_COS>
FHLM float sincosf_(float y) {
    const uint32_t yi = f2u(y);
    const uint32_t top = (yi >> 20) & 0x7ff;  // abstop12
    const double xd = (double)y;
    // reduce_fast (|y| < 120): x * (2^24 * 2 / pi), rounded to the nearest multiple of 2^24 by an integer add and shift; x - n * hpi is
    // one fused operation.  For |y| < pi / 4 it gives n = 0 and x unchanged, and the library's small-argument branch is the n = 0 path.
    const double r = xd * 0x1.45F306DC9C883p+23;
    int n = ((int32_t)r + 0x800000) >> 24;  // quadrant: picks the polynomial
    double x = fma_(-(double)n, 0x1.921FB54442D18p0, xd);
    int ns = n;                             // quadrant + sign bit (large arguments are reduced from |y|): picks sign and table
    if (top >= 0x42f) {                     // |y| >= 120 (inf / NaN: overridden below)
        x = sincosf_reduce_large<Tab>(yi, &n);
        ns = n + (int)(yi >> 31);
    }
    const double x2 = x * x;
    x = u2d(d2u(x) ^ ((uint64_t)((ns ^ (ns >> 1)) & 1) << 63));  // x * sign[ns & 3], sign = {1, -1, -1, 1}
    const int np = IS_COS ? n ^ 1 : n;
    float res;
    if ((np & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = fma_(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7);  // s2 + x2 * s3
        const double x7 = x3 * x2;
        const double s = fma_(x3, -0x1.555545995a603p-3, x);  // x + x3 * s1
        res = (float)fma_(x7, s1, s);                         // s + x7 * s1
    } else {
        const double x4 = x2 * x2;
        const double c2 = fma_(x2, 0x1.99343027bf8c3p-16, -0x1.6c087e89a359dp-10);  // c3 + x2 * c4
        const double c1 = fma_(x2, -0x1.ffffffd0c621cp-2, 0x1p0);                   // c0 + x2 * c1
        const double x6 = x4 * x2;
        const double c = fma_(x4, 0x1.55553e1068f19p-5, c1);  // c1 + x4 * c2
        res = (float)fma_(x6, c2, c);                         // c + x6 * c2
        res = u2f(f2u(res) ^ ((uint32_t)(ns & 2) << 30));     // table 1
    }
    res = top < 0x398 ? (IS_COS ? 1.0f : y) : res;  // |y| < 2^-12
    return top >= 0x7f8 ? nan_() : res;             // inf, NaN: __math_invalidf
}

// ---- expf (e_expf.c) ----
// the library's main path, for every argument (what it returns for |x| >= 88 or NaN is replaced by expf_fix_)
template <class Tab>
FHLM float expf_main_(float x) {
    const double xd = (double)x;
    // x * N / ln2 = k + r; the source's `z = InvLn2N * xd; kd = z + SHIFT; ...; r = z - kd` compiles to two fused operations
    const double SHIFT = 0x1.8p+52, InvLn2N = 0x1.71547652b82fep+5;
    double kd = fma_(InvLn2N, xd, SHIFT);
    const uint64_t ki = d2u(kd);
    kd -= SHIFT;
    const double r = fma_(InvLn2N, xd, -kd);
    const uint64_t t = Tab::exp2_tab((uint32_t)(ki & 31)) + (ki << 47);
    const double s = u2d(t);
    const double z = fma_(0x1.c6af84b912394p-20, r, 0x1.ebfce50fac4f3p-13);
    const double r2 = r * r;
    double y = fma_(0x1.62e42ff0c52d6p-6, r, 1.0);
    y = fma_(z, r2, y);
    y = y * s;
    return (float)y;
}
FHLM bool expf_special_(float x) { return ((f2u(x) >> 20) & 0x7ff) >= 0x42b; }  // |x| >= 88 or NaN
// ... as selects, taken by the comparisons themselves (no branch: for arguments that are not special nothing changes)
FHLM float expf_fix_(float x, float res) {
    const uint32_t xi = f2u(x), abstop = (xi >> 20) & 0x7ff;
    res = x < -0x1.9d1d9ep6f ? u2f(1u) : res;          // __math_may_uflowf: 0x1.4p-75f squared = the smallest subnormal
    res = x < -0x1.9fe368p6f ? 0.0f : res;             // underflow
    res = x > 0x1.62e42ep6f ? u2f(0x7f800000u) : res;  // overflow
    res = abstop >= 0x7f8 ? x + x : res;               // inf, NaN
    res = xi == 0xff800000u ? 0.0f : res;
    return res;
}
template <class Tab>
FHLM float expf_(float x) { return expf_fix_(x, expf_main_<Tab>(x)); }
// four samples: the main paths as one block (its table loads are issued together), the rare cases behind ONE test
template <class Tab>
FHLM void expf4_(const float* x, float* r) {
    for (int k = 0; k < 4; k++) r[k] = expf_main_<Tab>(x[k]);
    if (expf_special_(x[0]) | expf_special_(x[1]) | expf_special_(x[2]) | expf_special_(x[3]))
        for (int k = 0; k < 4; k++) r[k] = expf_fix_(x[k], r[k]);
}

// ---- logf (e_logf.c) ----
FHLM bool logf_special_(uint32_t ix) { return ix - 0x00800000u >= 0x7f800000u - 0x00800000u; }  // x < 2^-126, inf or NaN
template <class Tab>
FHLM float logf_main_(float x) {
    const uint32_t ix0 = f2u(x);
    const uint32_t ix = logf_special_(ix0) ? f2u(x * 0x1p23f) - (23u << 23) : ix0;  // subnormal: normalise
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) & 15;
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double invc = Tab::log_invc(i), logc = Tab::log_logc(i);
    const double z = (double)u2f(iz);
    const double r = fma_(z, invc, -1.0);
    const double y0 = fma_((double)k, 0x1.62e42fefa39efp-1, logc);
    const double r2 = r * r;
    double y = fma_(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);  // A[1] * r + A[2]
    y = fma_(-0x1.00ea348b88334p-2, r2, y);                            // A[0] * r2 + y
    y = fma_(y, r2, y0 + r);
    return (float)y;
}
// x < 2^-126, inf, NaN, and x = 1 (which the library answers before anything else) - as selects
FHLM float logf_fix_(float x, float res) {
    const uint32_t ix1 = opaque_(f2u(x));  // (the comparisons below are made here, not kept from the main path)
    res = (logf_special_(ix1) && ((ix1 & 0x80000000u) || ix1 * 2 >= 0xff000000u)) ? nan_() : res;
    res = ix1 == 0x7f800000u ? x : res;
    res = ix1 * 2 == 0 ? u2f(0xff800000u) : res;  // log(+-0) = -inf
    return ix1 == 0x3f800000u ? 0.0f : res;
}
template <class Tab>
FHLM float logf_(float x) { return logf_fix_(x, logf_main_<Tab>(x)); }
template <class Tab>
FHLM void logf4_(const float* x, float* r) {
    for (int k = 0; k < 4; k++) r[k] = logf_main_<Tab>(x[k]);
    bool any = false;
    for (int k = 0; k < 4; k++) { const uint32_t ix = opaque_(f2u(x[k])); any = any | logf_special_(ix) | (ix == 0x3f800000u); }
    if (any)
        for (int k = 0; k < 4; k++) r[k] = logf_fix_(x[k], r[k]);
}

// ---- fdlibm routines: binary32 arithmetic, one rounding per operation (no fused multiply-add anywhere) ----
FHLM float fabsf_(float x) { return u2f(f2u(x) & 0x7fffffffu); }
FHLM float sqrtf_(float x) { return __builtin_sqrtf(x); }           // correctly rounded on both sides
FHLM float trunc12(float x) { return u2f(f2u(x) & 0xfffff000u); }  // SET_FLOAT_WORD(w, i & 0xfffff000)

// __kernel_tanf (k_tanf.c): tan(x + y) for |x + y| <= pi / 4 when iy = 1, -1 / tan when iy = -1.  The library's three divisions
// (|x| >= 0.6744; the accurate -1 / (x + r); |x| < 2^-13) are one division with selected operands.
FHLM float kernel_tanf(float x0, float y0, int iy) {
    const float T0 = 0x1.555556p-2f, T1 = 0x1.111112p-3f, T2 = 0x1.ba1ba2p-5f, T3 = 0x1.664f48p-6f, T4 = 0x1.226e3ep-7f,
                T5 = 0x1.d6d22cp-9f, T6 = 0x1.7dbc9p-10f, T7 = 0x1.344d9p-11f, T8 = 0x1.026f72p-12f, T9 = 0x1.47e88ap-14f,
                T10 = 0x1.2b80f4p-14f, T11 = -0x1.375cbep-16f, T12 = 0x1.b2a708p-16f;
    const float pio4 = 0x1.921fb4p-1f, pio4lo = 0x1.4442dp-25f;
    const int32_t hx = (int32_t)f2u(x0);
    const int32_t ix = hx & 0x7fffffff;
    const bool tiny = ix < 0x39000000;  // |x| < 2^-13: (int)x == 0
    const bool big = ix >= 0x3f2ca140;  // |x| >= 0.6744
    const float sgn = (float)(1 - ((hx >> 30) & 2)), fiy = (float)iy;
    // big: x, y = |x|, y * sign; z = pio4 - x; w = pio4lo - y; x = z + w; y = 0
    const float ax = u2f(f2u(x0) & 0x7fffffffu), ay = u2f(f2u(y0) ^ (f2u(x0) & 0x80000000u));
    const float x = big ? (pio4 - ax) + (pio4lo - ay) : x0;
    const float y = big ? 0.0f : y0;
    const float z = x * x;
    float w = z * z;
    float r = T1 + w * (T3 + w * (T5 + w * (T7 + w * (T9 + w * T11))));
    float v = z * (T2 + w * (T4 + w * (T6 + w * (T8 + w * (T10 + w * T12)))));
    float s = z * x;
    r = y + z * (s * (r + v) + y);
    r += T0 * s;
    w = x + r;
    // big: w * w / (w + v) with v = iy;  otherwise -1 / w;  tiny: -1 / x (1 / |x| when x = 0 and iy = -1: -1 / -0)
    const float num = big ? w * w : -1.0f;
    const float den = tiny ? (ix == 0 ? -0.0f : x0) : big ? w + fiy : w;
    const float a = num / den;
    const float res_big = fabsf_(x) < 0x1p-13f ? (float)((1 - ((hx >> 30) & 2)) * iy) * (1.0f - (float)(2 * iy) * x)
                                                : sgn * (fiy - 2.0f * (x - (a - r)));
    // -1 / (x + r), accurately
    const float zz = trunc12(w);
    v = r - (zz - x);
    const float t = trunc12(a);
    s = 1.0f + t * zz;
    const float res_inv = t + a * (s + t * v);
    float res = iy == 1 ? w : res_inv;
    res = big ? res_big : res;
    return tiny ? (iy == 1 ? x0 : a) : res;
}

// tanf (s_tanf.c; since glibc 2.33 the reduction is sincosf's, in binary64 WITHOUT fused operations - this file has no _fma variant).
// Below pi / 4 the reduction gives n = 0, y[0] = x and y[1] = 0: the library's first branch.
template <class Tab>
FHLM float tanf_(float x) {
    const uint32_t xi = f2u(x);
    const uint32_t top = (xi >> 20) & 0x7ff;
    double dx = (double)x;
    const double r = dx * 0x1.45F306DC9C883p+23;
    int n = ((int32_t)r + 0x800000) >> 24;
    dx = dx - (double)n * 0x1.921FB54442D18p0;
    if (top >= 0x42f) {  // |x| >= 120
        dx = sincosf_reduce_large<Tab>(xi, &n);
        dx = u2d(d2u(dx) ^ ((uint64_t)(xi >> 31) << 63));
    }
    const float y0 = (float)dx;
    const float y1 = (float)(dx - (double)y0);
    const float res = kernel_tanf(y0, y1, 1 - ((n & 1) << 1));
    return top >= 0x7f8 ? nan_() : res;
}

// asinf (e_asinf.c, glibc's polynomial)
FHLM float asinf_(float x) {
    const float pio2_hi = 0x1.921fb6p+0f, pio2_lo = -0x1.777a5cp-25f, pio4_hi = 0x1.921fb6p-1f;
    const float p0 = 0x1.5555c8p-3f, p1 = 0x1.3301e4p-4f, p2 = 0x1.747e4ap-5f, p3 = 0x1.8c283cp-6f, p4 = 0x1.596d28p-5f;
    const int32_t hx = (int32_t)f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    const bool small = ix < 0x3f000000;  // |x| < 0.5
    const float t = small ? x * x : (1.0f - fabsf_(x)) * 0.5f;
    const float p = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
    const float res_small = x + x * p;
    const float s = sqrtf_(t);
    const float res_hi = pio2_hi - (2.0f * (s + s * p) - pio2_lo);  // |x| > 0.975
    const float w = trunc12(s);
    const float c = (t - w * w) / (s + w);
    const float pp = 2.0f * s * p - (pio2_lo - 2.0f * c);
    const float q = pio4_hi - 2.0f * w;
    const float res_mid = pio4_hi - (pp - q);
    float res = ix >= 0x3F79999A ? res_hi : res_mid;
    res = hx > 0 ? res : -res;
    res = small ? (ix < 0x32000000 ? x : res_small) : res;  // |x| < 2^-27: x
    res = ix == 0x3f800000 ? x * pio2_hi + x * pio2_lo : res;
    return ix > 0x3f800000 ? nan_() : res;
}

// acosf (e_acosf.c)
FHLM float acosf_(float x) {
    const float pi = 0x1.921fb4p+1f, pio2_hi = 0x1.921fb4p+0f, pio2_lo = 0x1.4442dp-24f;
    const float pS0 = 0x1.555556p-3f, pS1 = -0x1.4d612p-2f, pS2 = 0x1.9c155p-3f, pS3 = -0x1.48228cp-5f, pS4 = 0x1.9efe08p-11f,
                pS5 = 0x1.23de1p-15f, qS1 = -0x1.33a272p+1f, qS2 = 0x1.02ae5ap+1f, qS3 = -0x1.6066c2p-1f, qS4 = 0x1.3b8c5cp-4f;
    const int32_t hx = (int32_t)f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    const bool small = ix < 0x3f000000;  // |x| < 0.5
    // z = x^2, (1 + x) / 2 or (1 - x) / 2: 1 - |x| is 1 + x for negative x
    const float z = small ? x * x : (1.0f - fabsf_(x)) * 0.5f;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    const float res_small = pio2_hi - (x - (pio2_lo - x * r));
    const float s = sqrtf_(z);
    const float res_neg = pi - 2.0f * (s + (r * s - pio2_lo));  // x < -0.5
    const float df = trunc12(s);
    const float c = (z - df * df) / (s + df);
    const float res_pos = 2.0f * (df + (r * s + c));  // x > 0.5
    float res = hx < 0 ? res_neg : res_pos;
    res = small ? (ix <= 0x32800000 ? pio2_hi + pio2_lo : res_small) : res;
    res = ix == 0x3f800000 ? (hx > 0 ? 0.0f : pi + 2.0f * pio2_lo) : res;
    return ix > 0x3f800000 ? nan_() : res;
}

// atanf (s_atanf.c): the four reductions' divisions are one division with selected operands
FHLM float atanf_(float x0) {
    const float hi0 = 0x1.dac67p-2f, hi1 = 0x1.921fb4p-1f, hi2 = 0x1.f730bcp-1f, hi3 = 0x1.921fb4p+0f;
    const float lo0 = 0x1.586ed2p-28f, lo1 = 0x1.4442dp-25f, lo2 = 0x1.281f68p-25f, lo3 = 0x1.4442dp-24f;
    const float aT0 = 0x1.555556p-2f, aT1 = -0x1.99999ap-3f, aT2 = 0x1.24924ap-3f, aT3 = -0x1.c71c7p-4f, aT4 = 0x1.745cdcp-4f,
                aT5 = -0x1.3b0f2ap-4f, aT6 = 0x1.10d66ap-4f, aT7 = -0x1.dde2d6p-5f, aT8 = 0x1.97b4b2p-5f, aT9 = -0x1.2b4442p-5f,
                aT10 = 0x1.0ad3aep-6f;
    const int32_t hx = (int32_t)f2u(x0);
    const int32_t ix = hx & 0x7fffffff;
    const float ax = fabsf_(x0);
    const bool direct = ix < 0x3ee00000;  // |x| < 0.4375: no reduction
    const bool r0 = ix < 0x3f300000, r1 = ix < 0x3f980000, r2 = ix < 0x401c0000;
    // (every candidate is computed, then selected: nested conditional expressions come back from the compiler as nested branches)
    const float n0 = 2.0f * ax - 1.0f, n1 = ax - 1.0f, n2 = ax - 1.5f, d0 = 2.0f + ax, d1 = ax + 1.0f, d2 = 1.0f + 1.5f * ax;
    float num = -1.0f, den = ax, hi = hi3, lo = lo3;
    num = r2 ? n2 : num; den = r2 ? d2 : den; hi = r2 ? hi2 : hi; lo = r2 ? lo2 : lo;
    num = r1 ? n1 : num; den = r1 ? d1 : den; hi = r1 ? hi1 : hi; lo = r1 ? lo1 : lo;
    num = r0 ? n0 : num; den = r0 ? d0 : den; hi = r0 ? hi0 : hi; lo = r0 ? lo0 : lo;
    num = direct ? x0 : num; den = direct ? 1.0f : den;  // x / 1 = x: one unconditional division instead of a branch around it
    const float x = num / den;
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    const float rr = hi - ((x * (s1 + s2) - lo) - x);
    const float res_direct = x - x * (s1 + s2), res_huge = u2f(f2u(hi3 + lo3) | (f2u(x0) & 0x80000000u));
    float res = u2f(f2u(rr) ^ (f2u(x0) & 0x80000000u));  // hx < 0 ? -rr : rr
    res = direct ? res_direct : res;
    res = ix < 0x31000000 ? x0 : res;        // |x| < 2^-29: x
    res = ix >= 0x4c000000 ? res_huge : res;  // |x| >= 2^25: +-(hi3 + lo3)
    return ix > 0x7f800000 ? x0 + x0 : res;
}

// atan2f (e_atan2f.c)
FHLM float atan2f_(float y, float x) {
    const float pi_o_4 = 0x1.921fb6p-1f, pi_o_2 = 0x1.921fb6p+0f, pi = 0x1.921fb6p+1f, pi_lo = -0x1.777a5cp-24f;
    const int32_t hx = (int32_t)f2u(x), hy = (int32_t)f2u(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);  // 2 * sign(x) + sign(y)
    const bool x_one = hx == 0x3f800000;
    const float t = atanf_(x_one ? y : fabsf_(y / x));
    const int k = (iy - ix) >> 23;
    float z = t;
    z = (hx < 0 && k < -60) ? 0.0f : z;          // |y| / x < -2^60
    z = k > 60 ? pi_o_2 + 0.5f * pi_lo : z;      // |y / x| > 2^60
    const float zl = z - pi_lo;
    const float q0 = z, q1 = u2f(f2u(z) ^ 0x80000000u), q2 = pi - zl, q3 = zl - pi;
    float res = q3;
    res = m == 2 ? q2 : res;
    res = m == 1 ? q1 : res;
    res = m == 0 ? q0 : res;
    const float ysign = u2f(f2u(y) & 0x80000000u);
    const float ypi2 = u2f(f2u(pi_o_2) | f2u(ysign));  // hy < 0 ? -pi_o_2 : pi_o_2
    const bool y_inf = iy == 0x7f800000, x_inf = ix == 0x7f800000;
    res = y_inf ? ypi2 : res;
    // x is inf: +-pi/4, +-3pi/4 when y is too; +-0, +-pi otherwise - the sign is y's
    const float both = u2f(f2u(hx < 0 ? 3.0f * pi_o_4 : pi_o_4) | f2u(ysign));
    const float one = u2f(f2u(hx < 0 ? pi : 0.0f) | f2u(ysign));
    const float xinf_res = y_inf ? both : one;
    res = x_inf ? xinf_res : res;
    res = ix == 0 ? ypi2 : res;  // x = 0
    res = iy == 0 ? one : res;   // y = 0: +-0 for positive x (= y), +-pi for negative
    res = x_one ? t : res;
    const float nan_res = x + y;
    return (ix > 0x7f800000 || iy > 0x7f800000) ? nan_res : res;
}

}  // namespace fhlm
