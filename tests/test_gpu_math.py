"""Accuracy of the device's transcendental opcodes against the host libm.

The reference evaluates sin / cos / tan / asin / acos / atan / exp / ln with the platform's f32 libm (Rust std -> glibc here,
eval/test/mod.rs:194-203 compares against the same calls, i.e. pins nothing at the ulp level); the device evaluates them in
f64 and rounds once (dev_ops.hpp t_*).  The north star allows 1 ulp on f32 point values: this sweep measures it.  By default
every 64th f32 bit pattern per function (2^26 inputs each); FHIP_FULL_SWEEP=1 takes all 2^32 (tools/math_sweep.py writes the
table committed under profiles/)."""
import os

import numpy as np
import pytest

OPS = ["sin", "cos", "tan", "asin", "acos", "atan", "exp", "ln"]


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


@pytest.mark.gpu
@pytest.mark.parametrize("op", OPS)
def test_transcendental_within_one_ulp_of_libm(op, oracle_mod):
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
    assert worst["max_ulp"] <= 1, f"{op}: {worst}"
