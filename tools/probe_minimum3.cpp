// GPU box: what gfx950's v_minimum3_f32 / v_maximum3_f32 return, against fidget's min / max (types/float.rs: NaN if either operand is
// one, else a < b ? a : b / a > b ? a : b - dev_ops.hpp f_min / f_max) - all pairs of a table of special values + random bit patterns.
// Prints per category how many results differ in bits (a NaN equals any NaN).    hipcc --offload-arch=gfx950 -O2 -o /tmp/pm3 tools/probe_minimum3.cpp
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
__global__ void k(const float* a, const float* b, float* mn, float* mx, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b[i], r, s;
    asm volatile("v_minimum3_f32 %0, %1, %2, %2" : "=v"(r) : "v"(x), "v"(y));
    asm volatile("v_maximum3_f32 %0, %1, %2, %2" : "=v"(s) : "v"(x), "v"(y));
    mn[i] = r; mx[i] = s;
}
static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float fb(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
int main() {
    std::vector<uint32_t> sp = {0x00000000, 0x80000000, 0x00000001, 0x80000001, 0x007fffff, 0x807fffff, 0x00800000, 0x80800000, 0x3f800000, 0xbf800000,
                                0x7f7fffff, 0xff7fffff, 0x7f800000, 0xff800000, 0x7fc00000, 0xffc00000, 0x7f800001, 0xff800001, 0x7fc12345, 0x3dcccccd, 0xbdcccccd};
    std::vector<float> a, b;
    for (uint32_t x : sp) for (uint32_t y : sp) { a.push_back(fb(x)); b.push_back(fb(y)); }
    size_t n_special = a.size();
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < 1 << 22; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint32_t x = (uint32_t)s, y = (uint32_t)(s >> 32);
        if (i & 1) y = (y & 0x80000000u) | (x & 0x7fffffffu);          // equal magnitudes
        if ((i & 6) == 2) { x &= 0x807fffffu; y &= 0x807fffffu; }       // denormals
        a.push_back(fb(x)); b.push_back(fb(y));
    }
    int n = (int)a.size();
    float *da, *db, *dmn, *dmx;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dmn, n * 4); hipMalloc(&dmx, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, dmn, dmx, n);
    std::vector<float> mn(n), mx(n);
    hipMemcpy(mn.data(), dmn, n * 4, hipMemcpyDeviceToHost); hipMemcpy(mx.data(), dmx, n * 4, hipMemcpyDeviceToHost);
    long bad[2][4] = {{0}};      // [min / max][category]: 0 a NaN involved, 1 both zero, 2 a zero operand (not both), 3 anything else
    long badq[2] = {0, 0};
    for (int i = 0; i < n; i++) {
        float x = a[i], y = b[i];
        bool nan = std::isnan(x) || std::isnan(y);
        float want[2] = {nan ? NAN : (x < y ? x : y), nan ? NAN : (x > y ? x : y)};
        float got[2] = {mn[i], mx[i]};
        for (int m = 0; m < 2; m++) {
            bool same = std::isnan(want[m]) ? std::isnan(got[m]) : bits(want[m]) == bits(got[m]);
            if (same) continue;
            int cat = nan ? 0 : (x == 0 && y == 0) ? 1 : (x == 0 || y == 0) ? 2 : 3;
            if (bad[m][cat]++ < 6 && (size_t)i < n_special + 64) printf("%s(%08x, %08x) = %08x, want %08x\n", m ? "max" : "min", bits(x), bits(y), bits(got[m]), bits(want[m]));
            if (nan && !std::isnan(got[m])) badq[m]++;
        }
    }
    printf("pairs %d (special %zu)\n", n, n_special);
    for (int m = 0; m < 2; m++)
        printf("%s: differ with a NaN operand %ld (result not a NaN: %ld), both operands zero %ld, one zero %ld, other %ld\n", m ? "v_maximum3_f32" : "v_minimum3_f32", bad[m][0], badq[m], bad[m][1], bad[m][2], bad[m][3]);
    return 0;
}
