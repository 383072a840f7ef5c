"""Helpers shared by the ported known-answer tests."""
import ctypes
import math

import numpy as np

NAN = float("nan")
PI = float(np.float32(math.pi))

_libm = ctypes.CDLL("libm.so.6")
for _n in ("sinf", "cosf", "tanf", "asinf", "acosf", "atanf", "expf", "logf", "floorf", "ceilf", "roundf", "sqrtf", "fabsf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]
_libm.atan2f.restype = ctypes.c_float
_libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]
_libm.fmodf.restype = ctypes.c_float
_libm.fmodf.argtypes = [ctypes.c_float, ctypes.c_float]


def libm(name, *a):
    """Host libm (glibc), the same functions Rust's std calls on Linux."""
    return getattr(_libm, name)(*a)


def f32(x):
    return float(np.float32(x))


def spicy_args_n(n):
    """eval/test/mod.rs:48-63"""
    tau = np.float32(np.float32(math.pi) * np.float32(2.0))
    args = [f32(tau * np.float32(i) / np.float32(n)) for i in range(-n, n + 1)]
    args += [1.0, 5.0, 0.5, 1.5, 10.0, f32(math.pi), f32(math.pi / 2), f32(1 / math.pi), f32(math.sqrt(2)), NAN]
    return args


def spicy_args():
    return spicy_args_n(32)


def rng_hash(v):
    """rng/mod.rs:8-13"""
    v &= 0xFFFFFFFF
    state = (v * 747796405 + 2891336453) & 0xFFFFFFFF
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
    return ((word >> 22) ^ word) & 0xFFFFFFFF


def bits(f):
    return int(np.float32(f).view(np.uint32))


def from_bits(u):
    return float(np.uint32(u).view(np.float32))


def f_rand(a):
    h = rng_hash(bits(a))
    return f32(np.float32(from_bits((h >> 9) | 0x3F800000)) - np.float32(1.0))


def f_mix(a, b):
    return from_bits(rng_hash((bits(a) + rng_hash(bits(b))) & 0xFFFFFFFF))


def f_compare(a, b):
    if a < b:
        return -1.0
    if a == b:
        return 0.0
    if a > b:
        return 1.0
    return NAN


def f_min(a, b):
    if a < b:
        return a
    if b < a:
        return b
    return NAN if (math.isnan(a) or math.isnan(b)) else b


def f_max(a, b):
    if a > b:
        return a
    if b > a:
        return b
    return NAN if (math.isnan(a) or math.isnan(b)) else b


def rem_euclid(a, b):
    r = libm("fmodf", a, b)
    return f32(np.float32(r) + np.float32(abs(b))) if r < 0 else r


np.seterr(all="ignore")  # the sweeps deliberately hit inf/NaN


def _div(a, b):
    return f32(np.float32(a) / np.float32(b))


def _op(fn):
    def g(a, b):
        return f32(fn(np.float32(a), np.float32(b)))
    return g


# canonical op definitions, eval/test/mod.rs:190-243
UNARY_DEFS = {
    "neg": lambda a: -a,
    "recip": lambda a: _div(1.0, a),
    "abs": lambda a: abs(a),
    "sin": lambda a: libm("sinf", a),
    "cos": lambda a: libm("cosf", a),
    "tan": lambda a: libm("tanf", a),
    "asin": lambda a: libm("asinf", a),
    "acos": lambda a: libm("acosf", a),
    "atan": lambda a: libm("atanf", a),
    "exp": lambda a: libm("expf", a),
    "ln": lambda a: libm("logf", a),
    "square": lambda a: f32(np.float32(a) * np.float32(a)),
    "sqrt": lambda a: libm("sqrtf", a),
    "floor": lambda a: libm("floorf", a),
    "ceil": lambda a: libm("ceilf", a),
    "round": lambda a: libm("roundf", a),
    "not_": lambda a: 1.0 if a == 0.0 else 0.0,
    "rand": f_rand,
}
BINARY_DEFS = {
    "add": _op(lambda a, b: a + b),
    "sub": _op(lambda a, b: a - b),
    "mul": _op(lambda a, b: a * b),
    "div": _div,
    "min": f_min,
    "max": f_max,
    "compare": f_compare,
    "modulo": rem_euclid,
    "and_": lambda a, b: a if a == 0.0 else b,
    "or_": lambda a, b: a if a != 0.0 else b,
    "atan2": lambda y, x: libm("atan2f", y, x),
    "mix": f_mix,
}


def same(a, b):
    """float equality treating NaN == NaN (the reference's `o == v || both NaN`)."""
    return a == b or (math.isnan(a) and math.isnan(b))


def build_stress_fn(be, n):
    """eval/test/mod.rs:20-45"""
    ctx = be.Context()
    inputs = []
    s = ctx.constant(0.0)
    x, y, z = ctx.x(), ctx.y(), ctx.z()
    for i in range(1, n + 1):
        d = ctx.mul(float(i), [x, y, z][i % 3])
        inputs.append(d)
        s = ctx.add(s, d)
    s = ctx.sin(s)
    for i in reversed(inputs):
        s = ctx.add(s, i)
    return ctx, s
