"""Point / bulk-f32 / bulk-gradient known-answer tests, ported from the
reference's conformance suite:
  /root/reference/fidget-core/src/eval/test/point.rs
  /root/reference/fidget-core/src/eval/test/float_slice.rs
  /root/reference/fidget-core/src/eval/test/grad_slice.rs
Each test names the lines it restates.  Oracle on CPU; HIP backend with -m gpu.
"""
import math

import numpy as np
import pytest

from kat_util import (BINARY_DEFS, NAN, UNARY_DEFS, build_stress_fn, f32, f_mix, f_rand, libm, same, spicy_args)

L, R, B = 1, 2, 3
TRANSCENDENTAL = {"sin", "cos", "tan", "asin", "acos", "atan", "exp", "ln", "atan2"}


def ulp_close(a, b, ulps):
    if same(a, b):
        return True
    if math.isnan(a) or math.isnan(b) or math.isinf(a) or math.isinf(b):
        return False
    return abs(a - b) <= ulps * float(np.spacing(np.float32(abs(b))))


def value_ok(be, name, got, want):
    """Exact for every opcode on every backend: the transcendental opcodes are the host libm's routines restated
    (trans_libm.hpp), the reference's own bar (`o == v`, eval/test/float_slice.rs:404-412)."""
    return same(got, want)


def pt(shape, x=0.0, y=0.0, z=0.0):
    return shape.eval_point(x, y, z)


# ---- point.rs ---------------------------------------------------------------
def test_constant(be):  # point.rs:26-33
    ctx = be.Context()
    s = be.Shape(ctx, ctx.constant(1.5))
    assert s.eval_point_raw([])[0][0] == 1.5


def test_constant_push(be):  # point.rs:35-57
    ctx = be.Context()
    s = be.Shape(ctx, ctx.min(1.5, ctx.x()))
    r, trace = pt(s, 2.0)
    assert r == 1.5 and trace is not None
    nxt = s.simplify(trace)
    assert nxt.size() == 2  # constant, output
    assert nxt.eval_point_raw([2.0])[0][0] == 1.5
    assert nxt.eval_point_raw([1.0])[0][0] == 1.5
    with pytest.raises(ValueError):
        nxt.eval_point_raw([])  # vars are inherited from the parent (BadVarSlice)


def test_circle(be):  # point.rs:59-73
    ctx = be.Context()
    x, y = ctx.x(), ctx.y()
    c = ctx.sub(ctx.add(ctx.mul(x, x), ctx.mul(y, y)), 1.0)
    s = be.Shape(ctx, c)
    assert pt(s, 0.0, 0.0)[0] == -1.0
    assert pt(s, 1.0, 0.0)[0] == 0.0


def test_p_min_max(be):  # point.rs:75-137
    ctx = be.Context()
    x, y = ctx.x(), ctx.y()
    s = be.Shape(ctx, ctx.min(x, y))
    r, t = pt(s, 0.0, 0.0)
    assert r == 0.0 and t is None
    r, t = pt(s, 0.0, 1.0)
    assert r == 0.0 and list(t) == [L]
    r, t = pt(s, 2.0, 0.0)
    assert r == 0.0 and list(t) == [R]
    r, t = pt(s, NAN, 0.0)
    assert math.isnan(r) and t is None
    r, t = pt(s, 0.0, NAN)
    assert math.isnan(r) and t is None
    s = be.Shape(ctx, ctx.max(x, y))
    r, t = pt(s, 0.0, 0.0)
    assert r == 0.0 and t is None
    r, t = pt(s, 0.0, 1.0)
    assert r == 1.0 and list(t) == [R]
    r, t = pt(s, 2.0, 0.0)
    assert r == 2.0 and list(t) == [L]
    r, t = pt(s, NAN, 0.0)
    assert math.isnan(r) and t is None
    r, t = pt(s, 0.0, NAN)
    assert math.isnan(r) and t is None


def test_p_and_or(be):  # point.rs:139-209
    ctx = be.Context()
    x, y = ctx.x(), ctx.y()
    s = be.Shape(ctx, ctx.and_(x, y))
    for (a, b), (want, ch) in [((0.0, 0.0), (0.0, L)), ((0.0, 1.0), (0.0, L)), ((0.0, NAN), (0.0, L)),
                               ((f32(0.1), 1.0), (1.0, R)), ((f32(0.1), 0.0), (0.0, R)),
                               ((NAN, f32(1.2)), (f32(1.2), R))]:
        r, t = pt(s, a, b)
        assert r == want and list(t) == [ch]
    s = be.Shape(ctx, ctx.or_(x, y))
    for (a, b), (want, ch) in [((0.0, 0.0), (0.0, R)), ((0.0, 1.0), (1.0, R)), ((0.0, NAN), (NAN, R)),
                               ((f32(0.1), 1.0), (f32(0.1), L)), ((f32(0.1), 0.0), (f32(0.1), L)),
                               ((NAN, f32(1.2)), (NAN, L))]:
        r, t = pt(s, a, b)
        assert same(r, want) and list(t) == [ch]


def test_p_sin(be):  # point.rs:211-245
    ctx = be.Context()
    x = ctx.x()
    sn = ctx.sin(x)
    s = be.Shape(ctx, sn)
    for v in [0.0, 1.0, 2.0]:
        r, t = pt(s, v)
        assert value_ok(be, "sin", r, libm("sinf", v)) and t is None
    s = be.Shape(ctx, ctx.add(sn, ctx.y()))
    for a, b in [(0.0, 1.0), (1.0, 3.0), (2.0, 8.0)]:
        r, t = pt(s, a, b)
        assert ulp_close(r, f32(np.float32(libm("sinf", a)) + np.float32(b)), 0)
        assert t is None


def test_basic_interpreter(be):  # point.rs:247-260
    ctx = be.Context()
    x, y = ctx.x(), ctx.y()
    s = be.Shape(ctx, ctx.min(ctx.add(x, 1.0), y))
    assert pt(s, 1.0, 2.0)[0] == 2.0
    assert pt(s, 1.0, 3.0)[0] == 2.0
    assert pt(s, 3.0, 3.5)[0] == 3.5


def test_push(be):  # point.rs:262-327
    ctx = be.Context()
    x, y = ctx.x(), ctx.y()
    s = be.Shape(ctx, ctx.min(x, y))
    assert pt(s, 1.0, 2.0)[0] == 1.0
    assert pt(s, 3.0, 2.0)[0] == 2.0
    n = s.simplify([L])
    assert pt(n, 1.0, 2.0)[0] == 1.0
    assert pt(n, 3.0, 2.0)[0] == 3.0
    n = s.simplify([R])
    assert pt(n, 1.0, 2.0)[0] == 2.0
    assert pt(n, 3.0, 2.0)[0] == 2.0
    s = be.Shape(ctx, ctx.min(x, 1.0))
    assert pt(s, 0.5)[0] == 0.5
    assert pt(s, 3.0)[0] == 1.0
    n = s.simplify([L])
    assert pt(n, 0.5)[0] == 0.5
    assert pt(n, 3.0)[0] == 3.0
    n = s.simplify([R])
    assert pt(n, 0.5)[0] == 1.0
    assert pt(n, 3.0)[0] == 1.0
    with pytest.raises(ValueError):
        s.simplify([L, R])  # BadChoiceSlice (vm/data.rs:129-134)


def test_basic(be):  # point.rs:329-354
    ctx = be.Context()
    x, y = ctx.x(), ctx.y()
    assert pt(be.Shape(ctx, x), 1.0)[0] == 1.0
    assert pt(be.Shape(ctx, y), y=4.0)[0] == 4.0
    s = be.Shape(ctx, ctx.add(x, ctx.mul(y, 2.5)))
    assert pt(s, 1.0, 2.0)[0] == 6.0


def test_p_rand_mix(be):  # point.rs:445-548
    ctx = be.Context()
    a, b = ctx.x(), ctx.y()
    s = be.Shape(ctx, ctx.rand(a))
    for v in [0.0, 1.0, 2.0, f32(math.pi / 2)]:
        assert pt(s, v)[0] == f_rand(v)
    s = be.Shape(ctx, ctx.mix(a, b))
    assert pt(s, 0.0, 2.0)[0] == f_mix(0.0, 2.0)
    s = be.Shape(ctx, ctx.mix(a, 5.0))
    assert pt(s, 1.0)[0] == f_mix(1.0, 5.0)
    s = be.Shape(ctx, ctx.mix(5.0, a))
    assert pt(s, 1.0)[0] == f_mix(5.0, 1.0)


@pytest.mark.parametrize("n", [4, 8, 12, 16, 32, 256, 512])
def test_p_f_stress(be, oracle_mod, n):  # point.rs:389-443, float_slice.rs:265-316
    args = [f32(np.float32(i) / np.float32(32)) for i in range(32)]
    x, y, z = args, args[1:] + args[:1], args[2:] + args[:2]
    ctx, node = build_stress_fn(be, n)
    s = be.Shape(ctx, node)
    out = s.eval_float_slice(x, y, z)
    octx, onode = build_stress_fn(oracle_mod, n)
    ref = oracle_mod.Shape(octx, onode).eval_float_slice(x, y, z)
    for i in range(32):
        q = octx.eval_xyz(onode, x[i], y[i], z[i])
        assert abs(out[i] - q) < 1e-2
        assert out[i] == ref[i]  # `a == b` vs VmShape, on every backend
    if n <= 32:
        for i in range(0, 32, 5):
            assert ulp_close(pt(s, x[i], y[i], z[i])[0], float(out[i]), 0)


# ---- float_slice.rs ---------------------------------------------------------
def test_vectorized(be):  # float_slice.rs:48-93
    ctx = be.Context()
    x, y = ctx.x(), ctx.y()
    s = be.Shape(ctx, x)
    for n in (4, 8, 9):
        a = np.arange(n, dtype=np.float32)
        assert list(s.eval_float_slice(a, a * 0, a * 0)) == list(a)
    s = be.Shape(ctx, ctx.mul(y, 2.0))
    assert list(s.eval_float_slice([0] * 4, [3, 2, 1, 0], [0] * 4)) == [6, 4, 2, 0]
    assert list(s.eval_float_slice([0] * 3, [1, 4, 8], [0] * 3)) == [2, 8, 16]
    assert list(s.eval_float_slice([0] * 7, [1, 4, 4, -1, -2, -3, 0], [0] * 7)) == [2, 8, 8, -2, -4, -6, 0]


def test_f_sin_rand_mix(be):  # float_slice.rs:95-187
    ctx = be.Context()
    a, b = ctx.x(), ctx.y()
    args = [0.0, 1.0, 2.0, f32(math.pi / 2)]
    zeros = [0.0] * 4
    out = be.Shape(ctx, ctx.sin(a)).eval_float_slice(args, zeros, zeros)
    assert all(value_ok(be, "sin", float(o), libm("sinf", v)) for o, v in zip(out, args))
    out = be.Shape(ctx, ctx.rand(a)).eval_float_slice(args, zeros, zeros)
    assert [float(o) for o in out] == [f_rand(v) for v in args]
    out = be.Shape(ctx, ctx.mix(a, b)).eval_float_slice([2.0, 3.0], [0.0, 1.0], [0, 0])
    assert [float(o) for o in out] == [f_mix(2.0, 0.0), f_mix(3.0, 1.0)]
    out = be.Shape(ctx, ctx.mix(a, 5.0)).eval_float_slice([0.0, 1.0], [0, 0], [0, 0])
    assert [float(o) for o in out] == [f_mix(0.0, 5.0), f_mix(1.0, 5.0)]
    out = be.Shape(ctx, ctx.mix(5.0, a)).eval_float_slice([0.0, 1.0], [0, 0], [0, 0])
    assert [float(o) for o in out] == [f_mix(5.0, 0.0), f_mix(5.0, 1.0)]
    out = be.Shape(ctx, ctx.mix(a, NAN)).eval_float_slice([0.0, 1.0], [0, 0], [0, 0])
    assert all(same(float(o), f_mix(v, NAN)) for o, v in zip(out, [0.0, 1.0]))


def test_f_shape_var(be):  # float_slice.rs:189-263 (argument-shape errors)
    ctx = be.Context()
    v = 0xABCDEF
    a = ctx.add(ctx.add(ctx.x(), ctx.y()), ctx.var(v))
    s = be.Shape(ctx, a)
    out = s.eval_float_slice([1.0, 2.0], [2.0, 3.0], [0.0, 0.0], extra={v: np.array([4.0, 5.0], np.float32)})
    assert list(out) == [7.0, 10.0]
    with pytest.raises(ValueError):  # too few variable slices -> BadVarSlice (var/mod.rs:167-197)
        s.eval_float_slice_raw([np.zeros(2, np.float32)])
    with pytest.raises(ValueError):  # MismatchedSlices
        s.eval_float_slice_raw([np.zeros(2, np.float32), np.zeros(3, np.float32), np.zeros(2, np.float32)])
    # extra variables are fine
    vs = [np.zeros(2, np.float32)] * 5
    s.eval_float_slice_raw(vs)


def test_f_multiple_outputs(be):  # float_slice.rs:318-383
    ctx = be.Context()
    s = be.Shape(ctx, roots=[ctx.x(), ctx.y(), ctx.z()])
    vs = [None] * 3
    for axis, val in enumerate([0.0, 1.0, 2.0]):
        vs[s.axis_index(axis)] = np.full(8, val, np.float32)
    out = s.eval_float_slice_raw(vs)
    assert (out[0] == 0).all() and (out[1] == 1).all() and (out[2] == 2).all()
    s = be.Shape(ctx, roots=[ctx.x(), ctx.constant(4.0), ctx.constant(5.0)])
    out = s.eval_float_slice_raw([np.zeros(8, np.float32)])
    assert (out[0] == 0).all() and (out[1] == 4).all() and (out[2] == 5).all()


@pytest.mark.parametrize("name", list(UNARY_DEFS))
def test_f_unary(be, name):  # float_slice.rs:391-413
    args = spicy_args()
    ctx = be.Context()
    v = ctx.var(9)
    s = be.Shape(ctx, getattr(ctx, name)(v))
    out = s.eval_float_slice_raw([np.array(args, np.float32)])[0]
    for a, o in zip(args, out):
        assert value_ok(be, name, float(o), UNARY_DEFS[name](a)), f"{name} at {a}: {UNARY_DEFS[name](a)} != {o}"
    # the tracing point evaluator shares the definition (point.rs:550-592)
    for a in args[::7]:
        r, t = s.eval_point_raw([a])
        assert value_ok(be, name, float(r[0]), UNARY_DEFS[name](a)) and t is None


@pytest.mark.parametrize("name", list(BINARY_DEFS))
def test_f_binary(be, name):  # float_slice.rs:415-533
    args = spicy_args()
    fn = BINARY_DEFS[name]
    ctx = be.Context()
    va, vb = ctx.var(1), ctx.var(2)
    s = be.Shape(ctx, getattr(ctx, name)(va, vb))
    ia, ib = s.var_index(1), s.var_index(2)
    for rot in range(0, len(args), 3):
        rgsa = args[rot:] + args[:rot]
        vs = [None, None]
        vs[ia], vs[ib] = np.array(args, np.float32), np.array(rgsa, np.float32)
        out = s.eval_float_slice_raw(vs)[0]
        for a, b, o in zip(args, rgsa, out):
            assert value_ok(be, name, float(o), fn(a, b)), f"{name}(reg,reg) at {a} {b}: {fn(a, b)} != {o}"
    for imm in args[::5]:
        for imm_first in (False, True):
            ctx = be.Context()
            v = ctx.var(1)
            node = getattr(ctx, name)(imm, v) if imm_first else getattr(ctx, name)(v, imm)
            s = be.Shape(ctx, node)
            if s.var_count() == 0:
                continue  # constant-folded
            out = s.eval_float_slice_raw([np.array(args, np.float32)])[0]
            for a, o in zip(args, out):
                want = fn(imm, a) if imm_first else fn(a, imm)
                assert value_ok(be, name, float(o), want) or (math.isnan(want) and s.ssa_len() <= 2), \
                    f"{name}({'imm,reg' if imm_first else 'reg,imm'}) at {a} imm {imm}: {want} != {o}"


# ---- grad_slice.rs ----------------------------------------------------------
def g(shape, x, y=None, z=None):
    n = len(x)
    y = [0.0] * n if y is None else y
    z = [0.0] * n if z is None else z
    return [tuple(float(v) for v in row) for row in shape.eval_grad_slice(x, y, z)]


def test_g_xyz(be):  # grad_slice.rs:54-88
    ctx = be.Context()
    assert g(be.Shape(ctx, ctx.x()), [2.0], [3.0], [4.0])[0] == (2.0, 1.0, 0.0, 0.0)
    assert g(be.Shape(ctx, ctx.y()), [2.0], [3.0], [4.0])[0] == (3.0, 0.0, 1.0, 0.0)
    assert g(be.Shape(ctx, ctx.z()), [2.0], [3.0], [4.0])[0] == (4.0, 0.0, 0.0, 1.0)


def test_g_square_abs_sqrt(be):  # grad_slice.rs:90-147
    ctx = be.Context()
    x = ctx.x()
    s = be.Shape(ctx, ctx.square(x))
    assert g(s, [0.0])[0] == (0, 0, 0, 0)
    assert g(s, [1.0])[0] == (1, 2, 0, 0)
    assert g(s, [2.0])[0] == (4, 4, 0, 0)
    assert g(s, [3.0])[0] == (9, 6, 0, 0)
    s = be.Shape(ctx, ctx.abs(x))
    assert g(s, [2.0])[0] == (2, 1, 0, 0)
    assert g(s, [-2.0])[0] == (2, -1, 0, 0)
    s = be.Shape(ctx, ctx.sqrt(x))
    assert g(s, [1.0])[0] == (1, 0.5, 0, 0)
    assert g(s, [4.0])[0] == (2, 0.25, 0, 0)


def test_g_sin(be):  # grad_slice.rs:149-170 (compare_eq: 1e-6)
    ctx = be.Context()
    x = ctx.x()
    v = g(be.Shape(ctx, ctx.sin(x)), [1.0, 2.0, 3.0])
    for got, a in zip(v, [1.0, 2.0, 3.0]):
        want = (math.sin(a), math.cos(a), 0, 0)
        assert max(abs(p - q) for p, q in zip(got, want)) < 1e-6
    y = ctx.mul(ctx.y(), 2.0)
    v = g(be.Shape(ctx, ctx.sin(y)), [0.0] * 3, [1.0, 2.0, 3.0])
    for got, a in zip(v, [2.0, 4.0, 6.0]):
        want = (math.sin(a), 0, 2 * math.cos(a), 0)
        assert max(abs(p - q) for p, q in zip(got, want)) < 1e-6


def test_g_mul_div_recip(be):  # grad_slice.rs:172-226
    ctx = be.Context()
    x, y = ctx.x(), ctx.y()
    s = be.Shape(ctx, ctx.mul(x, y))
    assert g(s, [1.0], [0.0])[0] == (0, 0, 1, 0)
    assert g(s, [0.0], [1.0])[0] == (0, 1, 0, 0)
    assert g(s, [4.0], [1.0])[0] == (4, 1, 4, 0)
    assert g(s, [4.0], [2.0])[0] == (8, 2, 4, 0)
    assert g(be.Shape(ctx, ctx.div(x, 2.0)), [1.0])[0] == (0.5, 0.5, 0, 0)
    s = be.Shape(ctx, ctx.recip(x))
    assert g(s, [1.0])[0] == (1, -1, 0, 0)
    assert g(s, [2.0])[0] == (0.5, -0.25, 0, 0)


def test_g_min_max(be):  # grad_slice.rs:228-286
    ctx = be.Context()
    x, y, z = ctx.x(), ctx.y(), ctx.z()
    mn = ctx.min(x, y)
    s = be.Shape(ctx, mn)
    assert g(s, [2.0], [3.0])[0] == (2, 1, 0, 0)
    assert g(s, [4.0], [3.0])[0] == (3, 0, 1, 0)
    s = be.Shape(ctx, ctx.max(mn, z))
    assert g(s, [2.0], [3.0], [0.0])[0] == (2, 1, 0, 0)
    assert g(s, [4.0], [3.0], [0.0])[0] == (3, 0, 1, 0)
    assert g(s, [4.0], [3.0], [5.0])[0] == (5, 0, 0, 1)
    s = be.Shape(ctx, ctx.max(x, y))
    assert g(s, [2.0], [3.0])[0] == (3, 0, 1, 0)
    assert g(s, [4.0], [3.0])[0] == (4, 1, 0, 0)


def test_g_add_not_rand_mix(be):  # grad_slice.rs:288-390
    ctx = be.Context()
    a, b = ctx.x(), ctx.y()
    s = be.Shape(ctx, ctx.add(a, b))
    vs = [None, None]
    vs[s.axis_index(0)] = np.array([[0, 0, 0, 0], [1, 0, 0, 0]], np.float32)
    vs[s.axis_index(1)] = np.array([[2, 0, 0, 0], [3, 0, 0, 0]], np.float32)
    out = s.eval_grad_slice_raw(vs)[0]
    assert [tuple(r) for r in out] == [(2, 0, 0, 0), (4, 0, 0, 0)]
    assert g(be.Shape(ctx, ctx.not_(a)), [0.0])[0] == (1, 0, 0, 0)
    args = [0.0, 1.0, 2.0, f32(math.pi / 2)]
    out = g(be.Shape(ctx, ctx.rand(a)), args)
    assert out == [(f_rand(v), 0, 0, 0) for v in args]
    out = g(be.Shape(ctx, ctx.mix(a, b)), [2.0, 3.0], [0.0, 1.0])
    assert out == [(f_mix(2.0, 0.0), 0, 0, 0), (f_mix(3.0, 1.0), 0, 0, 0)]
    out = g(be.Shape(ctx, ctx.mix(a, 5.0)), [0.0, 1.0])
    assert out == [(f_mix(0.0, 5.0), 0, 0, 0), (f_mix(1.0, 5.0), 0, 0, 0)]
    out = g(be.Shape(ctx, ctx.mix(5.0, a)), [0.0, 1.0])
    assert out == [(f_mix(5.0, 0.0), 0, 0, 0), (f_mix(5.0, 1.0), 0, 0, 0)]


def test_g_circle_modulo(be):  # grad_slice.rs:392-445
    ctx = be.Context()
    x, y = ctx.x(), ctx.y()
    c = ctx.sub(ctx.sqrt(ctx.add(ctx.square(x), ctx.square(y))), 0.5)
    s = be.Shape(ctx, c)
    assert g(s, [1.0], [0.0])[0] == (0.5, 1, 0, 0)
    assert g(s, [0.0], [1.0])[0] == (0.5, 0, 1, 0)
    assert g(s, [2.0], [0.0])[0] == (1.5, 1, 0, 0)
    assert g(s, [0.0], [2.0])[0] == (1.5, 0, 1, 0)
    s = be.Shape(ctx, ctx.sub(y, ctx.modulo(x, 1.0)))
    assert g(s, [0.0], [0.5])[0] == (0.5, -1, 1, 0)
    assert g(s, [f32(-0.01)], [0.5])[0] == (f32(np.float32(0.5) - np.float32(rem(f32(-0.01)))), -1, 1, 0)
    assert g(s, [f32(0.01)], [0.5])[0] == (f32(np.float32(0.5) - np.float32(f32(0.01))), -1, 1, 0)


def rem(a):
    from kat_util import rem_euclid
    return rem_euclid(a, 1.0)


@pytest.mark.parametrize("n", [4, 8, 12, 16, 32])
def test_g_stress(be, oracle_mod, n):  # grad_slice.rs:447-493
    args = [f32(np.float32(i) / np.float32(32)) for i in range(32)]
    x, y, z = args, args[1:] + args[:1], args[2:] + args[:2]
    ctx, node = build_stress_fn(be, n)
    out = be.Shape(ctx, node).eval_grad_slice(x, y, z)
    octx, onode = build_stress_fn(oracle_mod, n)
    ref = oracle_mod.Shape(octx, onode).eval_grad_slice(x, y, z)
    assert np.abs(out - ref).max() < 1e-6 * max(1.0, float(n))  # compare_eq, scaled by the gradient magnitude


@pytest.mark.parametrize("name", [n for n in UNARY_DEFS if n not in ("rand",)])
def test_g_unary(be, name):  # grad_slice.rs:495-644: exact value + finite-difference gradient
    args = [a for a in spicy_args()[::3] if not math.isnan(a)]
    ctx = be.Context()
    s = be.Shape(ctx, getattr(ctx, name)(ctx.x()))
    out = g(s, args)
    fn = UNARY_DEFS[name]
    for a, o in zip(args, out):
        assert value_ok(be, name, o[0], fn(a)), f"{name} value at {a}"
        if name in ("floor", "ceil", "round", "not_"):
            assert o[1:] == (0, 0, 0)
            continue
        eps = 1e-3
        d = (fn(f32(a + eps)) - fn(f32(a - eps))) / (f32(a + eps) - f32(a - eps))
        if math.isfinite(d) and math.isfinite(o[1]) and abs(d) < 1e3 and not (name == "abs" and abs(a) < 2 * eps):
            if name in ("tan", "recip", "ln", "sqrt", "asin", "acos") and (abs(o[1]) > 50 or abs(a) < 0.05):
                continue  # near a pole: the reference skips these via its error scaling
            assert abs(o[1] - d) < 1e-2 * max(1.0, abs(d)), f"{name} d/dx at {a}: {o[1]} vs {d}"
        assert all(v == 0 or math.isnan(v) for v in o[2:])  # 0/NaN outside a function's domain
