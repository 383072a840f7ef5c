"""The device's transcendental opcodes against the host libm: identical bits.

The reference evaluates sin / cos / tan / asin / acos / atan / atan2 / exp / ln with the platform's f32 libm (Rust std -> glibc here)
and its bulk tests assert exact equality with those calls (eval/test/float_slice.rs:404-412, canonical ops eval/test/mod.rs:194-203).
The device runs that libm's routines restated operation by operation (fidget_amd/csrc/trans_libm.hpp), so every result must equal
the host's bit for bit - a NaN equals any NaN, +0 and -0 differ.  By default every 64th f32 bit pattern per function (2^26 inputs
each); FHIP_FULL_SWEEP=1 takes all 2^32 (tools/math_sweep.py writes the table committed under profiles/).  The same comparison on the
CPU - the restatement compiled for the host against the running libm, all 2^32 arguments - is tools/libm_sweep.cpp; its sample
here is part of the CPU suite."""
import os
import subprocess

import numpy as np
import pytest

OPS = ["sin", "cos", "tan", "asin", "acos", "atan", "exp", "ln", "atan2"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sweep(F, O, hip, op, first, stride, count):
    ref = O.math_unary(op, first, stride, count)
    out = np.zeros(4, np.uint64)
    hip.check(F.lib().fhip_debug_math_sweep(hip._h, OPS.index(op), first, stride, count, ref.ctypes.data_as(F.C.c_void_p),
                                            out.ctypes.data_as(F.C.c_void_p)))
    return {"max_ulp": int(out[0]), "differ": int(out[1]), "over_1_ulp": int(out[2]), "worst_input_bits": int(out[3])}


def test_oracle_libm_is_glibc_f32(oracle_mod):
    """CPU leg: the sweep's reference values are the f32 libm calls, NaN / domain behaviour included"""
    O = oracle_mod
    x = np.array([0.5, -2.0, 1e10, np.inf, np.nan], np.float32).view(np.uint32)
    for k, v in enumerate(x):
        got = O.math_unary("sin", int(v), 1, 1)[0]
        want = np.float32(np.sin(np.float64(np.array([v], np.uint32).view(np.float32)[0])))
        assert (np.isnan(got) and np.isnan(want)) or abs(float(got) - float(want)) <= 2 * np.spacing(abs(want))
    assert np.isnan(O.math_unary("ln", np.float32(-1).view(np.uint32).item(), 1, 1)[0])
    assert O.math_unary("exp", 0, 1, 1)[0] == 1.0


def test_restated_libm_equals_the_running_libm_on_the_host(tmp_path):
    """trans_libm.hpp compiled for the host (no contraction, explicit fused operations) against the running glibc: every 1021st
    argument of the eight unary routines and of the four-sample exp / ln here, all 2^32 with `tools/libm_sweep.cpp all` (0 differ, profiles/r04*/libm_sweep_cpu.txt)"""
    exe = str(tmp_path / "libm_sweep")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-mfma", "-ffp-contract=off", "-fno-builtin", "-fopenmp", "-DFHLM_HAVE_FDLIBM",
                           os.path.join(ROOT, "tools", "libm_sweep.cpp"), "-o", exe, "-lm"])
    out = subprocess.run([exe, "unary", "1021"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert out.stdout.count("differ 0") == 10, out.stdout      # the eight unary routines + the four-sample forms of exp and ln


@pytest.mark.gpu
@pytest.mark.parametrize("op", OPS)
def test_transcendental_bits_equal_libm(op, oracle_mod):
    import fidget_amd as F
    hip = F.default_context()
    full = os.environ.get("FHIP_FULL_SWEEP") == "1"
    stride, chunks = (1, 64) if full else (64, 1)
    n = 1 << 26
    worst = {"max_ulp": 0, "differ": 0, "over_1_ulp": 0}
    for c in range(chunks):
        r = sweep(F, oracle_mod, hip, op, (c * n * stride + (0 if full else OPS.index(op) * 7)) & 0xFFFFFFFF, stride, n)
        worst["differ"] += r["differ"]
        worst["over_1_ulp"] += r["over_1_ulp"]
        if r["max_ulp"] > worst["max_ulp"]:
            worst["max_ulp"], worst["worst_input_bits"] = r["max_ulp"], r["worst_input_bits"]
    assert worst["differ"] == 0, f"{op}: {worst}"


# the copies of the compiled routines inside the assembly kernels (gen_trans.COPIES, in the order gen_interp.py's main() embeds them)
# and the routines each holds (gen_trans.FUNCS + FUNCS4).  fh_columns_t: the one-sample routines, and under the four-sample routines'
# numbers the two-sample sinf, cosf, expf and logf written by hand that its SIN / COS / EXP / LN handlers hold (gen_trans.sincos_pair,
# exp_pair, ln_pair: the probe runs them on the four arguments); the bulk kernels: sinf and cosf that way
ROUTINES = ["sin", "cos", "tan", "asin", "acos", "atan", "exp", "ln", "atan2", "mod", "sin4", "cos4", "exp4", "ln4"]
COPIES = [("fh_columns_t", 14), ("fh_normals_t", 10), ("fh_tiles_t", 10), ("fh_tiles_v32_t", 10), ("fh_tiles_v64_t", 10),
          ("fh_float_eval_16x4_t", 12), ("fh_float_eval_32x2_t", 12)]


def test_the_probe_reaches_every_embedded_copy():
    """CPU leg: the generator's list of copies is the one the GPU test walks"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "fidget_amd", "csrc"))
    import importlib
    gen_trans = importlib.import_module("gen_trans")
    assert gen_trans.FUNCS + gen_trans.FUNCS4 == ROUTINES
    src = open(os.path.join(ROOT, "fidget_amd", "csrc", "_gen", "interp_gfx950.s")).read()
    for prefix, n in (("fh_t_", 10), ("fh_tn_", 10), ("fh_til_", 10), ("fh_ti32_", 10), ("fh_ti64_", 10), ("fh_tb16_", 10), ("fh_tb32_", 10)):
        for r in ROUTINES[:n]:
            assert f"s_mov_b32 s100, {prefix}{r} - " in src, (prefix, r)      # fh_trans_probe's jump to that copy
    # (a one-sample routine is called four times, a four-sample routine once; the hand-written expf's special cases: four calls of exp)
    assert src.count(" - .Lfar_") == sum(4 * min(n, 10) + 4 * (n - 10) for _, n in COPIES)
    assert "v_fma_f64" in src[src.index(".Lfh_trans_probe_store") - 60000:src.index(".Lfh_trans_probe_store")]      # ... and the probe holds exp_pair itself


@pytest.mark.gpu
@pytest.mark.parametrize("copy", range(len(COPIES)), ids=[c[0] for c in COPIES])
def test_embedded_routines_equal_the_inlined_ones(copy):
    """The text the hot kernels execute - the routines compiled apart (SGPR budget), renamed into each kernel's register window and
    compacted by gen_trans.py - against the routines as hipcc inlines them into the HIP kernels, which the test above holds to the
    host libm: all 2^32 arguments of every routine of every copy (second arguments of atan2 / mod: a hash of the first)."""
    import fidget_amd as F
    hip = F.default_context()
    n = 1 << 28
    for fn in range(COPIES[copy][1]):
        differ, where = 0, None
        for c in range(16):
            out = np.zeros(2, np.uint64)
            hip.check(F.lib().fhip_debug_trans_probe(hip._h, copy, fn, (c * n) & 0xFFFFFFFF, n, out.ctypes.data_as(F.C.c_void_p)))
            differ += int(out[0])
            if out[0]:
                where = hex(int(out[1]))
        assert differ == 0, f"{COPIES[copy][0]} {ROUTINES[fn]}: {differ} results differ, e.g. at input bits {where}"


@pytest.mark.gpu
def test_the_probe_reports_a_copy_that_is_not_there():
    import fidget_amd as F
    hip = F.default_context()
    out = np.zeros(2, np.uint64)
    hip.check(F.lib().fhip_debug_trans_probe(hip._h, 1, 12, 0x3F000000, 1 << 16, out.ctypes.data_as(F.C.c_void_p)))   # fh_normals_t has no exp4
    assert out[0] > 60000
