// Transcendental / modulo routines for the assembly interpreters, as out-of-line device functions.
//
// The reference evaluates sin, cos, ... with the platform libm; the device's definition of those opcodes is dev_ops.hpp
// t_* (trans_libm.hpp: the host libm's routines restated) and the HIP kernels inline exactly these functions.  The assembly
// interpreters must return the same bits, so they CALL the same code: this file is compiled to gfx950 assembly (hipcc to LLVM IR,
// llc with the functions limited to eight scalar registers: fidget_amd.build; no scratch; the tables of expf / logf / the large
// argument reduction are loaded from .rodata), gen_trans.py renames the registers of each function into a window the interpreters
// keep free (v0.. -> v128.., s0.. -> s86.., return address s[30:31] -> s[96:97]) and embeds the bodies and the tables in the
// interpreters' code object.
#include <hip/hip_runtime.h>

#include "dev_ops.hpp"

#define FH_NI extern "C" __device__ __attribute__((noinline, used))
FH_NI float fh_t_sin(float a) { return fhd::t_sin(a); }
FH_NI float fh_t_cos(float a) { return fhd::t_cos(a); }
FH_NI float fh_t_tan(float a) { return fhd::t_tan(a); }
FH_NI float fh_t_asin(float a) { return fhd::t_asin(a); }
FH_NI float fh_t_acos(float a) { return fhd::t_acos(a); }
FH_NI float fh_t_atan(float a) { return fhd::t_atan(a); }
FH_NI float fh_t_exp(float a) { return fhd::t_exp(a); }
FH_NI float fh_t_ln(float a) { return fhd::t_ln(a); }
FH_NI float fh_t_atan2(float y, float x) { return fhd::t_atan2(y, x); }
// Four samples per call (arguments and results in v0..v3) for the leaf interpreter, which evaluates an op for the 8 voxels of a lane: the
// call, the constants and - for expf / logf - the latency of the table loads are paid once per four samples instead of once per sample.
typedef float fh_f4 __attribute__((ext_vector_type(4)));
#define FH_T4(name, fn) FH_NI fh_f4 fh_t_##name##4(fh_f4 a) { fh_f4 r; for (int k = 0; k < 4; k++) r[k] = fhd::fn(a[k]); return r; }
FH_T4(sin, t_sin)
FH_T4(cos, t_cos)
// (exp / ln: the four main paths as one block - the table loads go out together - and the rare cases behind one test, trans_libm.hpp)
FH_NI fh_f4 fh_t_exp4(fh_f4 a) { float x[4] = {a[0], a[1], a[2], a[3]}, r[4]; fhlm::expf4_<fhlm::MemTables>(x, r); return fh_f4{r[0], r[1], r[2], r[3]}; }
FH_NI fh_f4 fh_t_ln4(fh_f4 a) { float x[4] = {a[0], a[1], a[2], a[3]}, r[4]; fhlm::logf4_<fhlm::MemTables>(x, r); return fh_f4{r[0], r[1], r[2], r[3]}; }
FH_NI float fh_t_mod(float a, float b) { return fhd::rem_euclid(a, b); }
// (a kernel that references them keeps the functions in the device image)
__global__ void fh_trans_keep(float* p) {
    p[0] = fh_t_sin(p[0]) + fh_t_cos(p[0]) + fh_t_tan(p[1]) + fh_t_asin(p[2]) + fh_t_acos(p[2]) + fh_t_atan(p[2]) + fh_t_exp(p[1]) + fh_t_ln(p[2]) +
           fh_t_atan2(p[3], p[4]) + fh_t_mod(p[3], p[4]);
    const fh_f4 a = {p[5], p[6], p[7], p[8]}, r = fh_t_sin4(a) + fh_t_cos4(a) + fh_t_exp4(a) + fh_t_ln4(a);
    p[1] = r[0] + r[1] + r[2] + r[3];
}
