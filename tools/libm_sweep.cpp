// CPU sweep: fidget_amd/csrc/trans_libm.hpp against the running libm over all 2^32 arguments (a NaN equals any NaN; everything else
// bit for bit, the sign of zero included).  Build and run: tools/libm_sweep.sh
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../fidget_amd/csrc/trans_libm.hpp"

using namespace fhlm;
typedef float (*fn1)(float);
// the four-sample routines of the leaf interpreter, through one of their samples (the others: fixed ordinary arguments)
static float exp4_first(float x) { const float a[4] = {x, 1.5f, -3.25f, 20.0f}; float r[4]; expf4_<MemTables>(a, r); return r[0]; }
static float ln4_last(float x) { const float a[4] = {2.5f, 0.75f, 1.0e-3f, x}; float r[4]; logf4_<MemTables>(a, r); return r[3]; }
struct Case { const char* name; fn1 mine, ref; };

int main(int argc, char** argv) {
    const Case cases[] = {
        {"sin", sincosf_<MemTables, false>, sinf}, {"cos", sincosf_<MemTables, true>, cosf},
        {"exp", expf_<MemTables>, expf},           {"ln", logf_<MemTables>, logf},
        {"exp4", exp4_first, expf},                {"ln4", ln4_last, logf},
#ifdef FHLM_HAVE_FDLIBM
        {"tan", tanf_<MemTables>, tanf},           {"asin", asinf_, asinf}, {"acos", acosf_, acosf}, {"atan", atanf_, atanf},
#endif
    };
    const unsigned stride = argc > 2 ? atoi(argv[2]) : 1;
    int bad = 0;
    for (const Case& c : cases) {
        if (argc > 1 && strcmp(argv[1], "all") && strcmp(argv[1], "unary") && strcmp(argv[1], c.name)) continue;
        unsigned long long differ = 0;
        unsigned first = 0;
        bool have = false;
#pragma omp parallel for reduction(+ : differ) schedule(dynamic, 1)
        for (long long blk = 0; blk < 4096; blk++) {
            for (unsigned long long j = 0; j < (1ull << 20); j += stride) {
                const uint32_t b = (uint32_t)((blk << 20) + j);
                const float a = c.mine(u2f(b)), r = c.ref(u2f(b));
                if (a != a && r != r) continue;
                if (f2u(a) != f2u(r)) {
                    differ++;
#pragma omp critical
                    if (!have || b < first) { have = true; first = b; }
                }
            }
        }
        printf("%-5s differ %llu", c.name, differ);
        if (have) printf("  first 0x%08x: mine %a (0x%08x) libm %a (0x%08x)", first, c.mine(u2f(first)), f2u(c.mine(u2f(first))), c.ref(u2f(first)), f2u(c.ref(u2f(first))));
        printf("\n");
        bad += differ != 0;
    }
#ifdef FHLM_HAVE_FDLIBM
    if (argc <= 1 || !strcmp(argv[1], "all") || !strcmp(argv[1], "atan2")) {
        // atan2: every pair of 4096 x 4096 arguments spread over the exponent range (special values included) + 2^31 random pairs
        static uint32_t v[4096];
        int nv = 0;
        const uint32_t sp[] = {0, 0x80000000u, 0x3f800000u, 0xbf800000u, 0x7f800000u, 0xff800000u, 0x7fc00000u, 1, 0x80000001u, 0x007fffffu, 0x00800000u, 0x7f7fffffu, 0xff7fffffu};
        for (uint32_t x : sp) v[nv++] = x;
        uint32_t st = 12345;
        while (nv < 4096) { st = st * 1664525u + 1013904223u; v[nv] = (uint32_t)(((uint64_t)nv << 20) ^ (st >> 9)); nv++; }
        unsigned long long differ = 0;
#pragma omp parallel for reduction(+ : differ) schedule(dynamic, 16)
        for (int i = 0; i < 4096; i++)
            for (int j = 0; j < 4096; j++) {
                const float a = atan2f_(u2f(v[i]), u2f(v[j])), r = atan2f(u2f(v[i]), u2f(v[j]));
                if (!(a != a && r != r) && f2u(a) != f2u(r)) differ++;
            }
#pragma omp parallel for reduction(+ : differ) schedule(dynamic, 16)
        for (long long blk = 0; blk < 2048; blk++) {
            uint64_t s = 0x9E3779B97F4A7C15ull * (blk + 1);
            for (int j = 0; j < (1 << 20); j++) {
                s ^= s << 13; s ^= s >> 7; s ^= s << 17;
                const float y = u2f((uint32_t)s), x = u2f((uint32_t)(s >> 32));
                const float a = atan2f_(y, x), r = atan2f(y, x);
                if (!(a != a && r != r) && f2u(a) != f2u(r)) differ++;
            }
        }
        printf("atan2 differ %llu (2^24 structured + 2^31 random pairs)\n", differ);
        bad += differ != 0;
    }
#endif
    return bad != 0;
}
