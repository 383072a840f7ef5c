"""Harness for running the assembled gfx950 kernels on the CPU emulator (tools/gfx950_emu.py) and numpy
restatements of what they compute, on the device tape format (tape_format.h).

TEST INFRASTRUCTURE ONLY.  The numpy evaluators follow fidget_amd/csrc/dev_ops.hpp (which cites
fidget-core/src/types/{float,interval}.rs line by line) and kernels.hip prune_sweep (vm/data.rs:123-318)."""
import ctypes
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gfx950_emu as E  # noqa: E402

F32 = np.float32
U32 = np.uint32
GEN = os.path.join(ROOT, "fidget_amd", "csrc", "_gen")

OPS = ["OUTPUT", "INPUT", "COPY_REG", "COPY_IMM", "NEG", "ABS", "RECIP", "SQRT", "SQUARE", "FLOOR", "CEIL", "ROUND", "SIN", "COS", "TAN",
       "ASIN", "ACOS", "ATAN", "EXP", "LN", "NOT", "RAND"]
BIN = ["ADD", "SUB", "MUL", "DIV", "ATAN2", "COMPARE", "MIX", "MOD", "MIN", "MAX", "AND", "OR"]
OPS += [b + "_RR" for b in BIN] + [b + "_RI" for b in BIN] + [b + "_IR" for b in ("SUB", "DIV", "ATAN2", "COMPARE", "MIX", "MOD")]
OPN = {n: i for i, n in enumerate(OPS)}

_prog = None


def program():
    global _prog
    if _prog is None:
        alt = os.environ.get("FH_EMU_CO")       # (a variant's code object, tools/build_variant.py: the emulator tests without the library build)
        if alt:
            _prog = E.Program(alt)
            return _prog
        import fidget_amd
        fidget_amd.build()
        _prog = E.Program(os.path.join(GEN, "interp_gfx950.co"))
    return _prog


def offsets():
    return json.load(open(os.path.join(GEN, "offsets.json")))


def f2u(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def u2f(x):
    return struct.unpack("<f", struct.pack("<I", x & 0xFFFFFFFF))[0]


def decode(w):
    w = int(w)
    w0, w1 = w & 0xFFFFFFFF, w >> 32
    return w0 & 0xFF, (w0 >> 8) & 0xFFF, w0 >> 20, w1


def pack(op, out=0, a=0, w1=0):
    return (op | (out << 8) | (a << 20)) | (w1 << 32)


def is_choice(op):
    return 30 <= op <= 33 or 42 <= op <= 45


def is_rr(op):
    return 22 <= op <= 33


def split(op):
    """(base index 0..11 or None, form)"""
    if op >= 46:
        return [1, 3, 4, 5, 6, 7][op - 46], "IR"
    if op >= 34:
        return op - 34, "RI"
    if op >= 22:
        return op - 22, "RR"
    return None, None


# ---- f32 ----------------------------------------------------------------------------------------
def _round(x):
    t = np.trunc(x)
    d = x - t
    return (t + np.copysign((np.abs(d) >= 0.5).astype(F32), x)).astype(F32)


def _f_compare(a, b):
    with np.errstate(all="ignore"):
        return np.where(a < b, F32(-1), np.where(a == b, F32(0), np.where(a > b, F32(1), F32(np.nan)))).astype(F32)


def pcg(v):
    """rng::hash (fidget-core/src/rng/mod.rs:8-13) on uint32 arrays"""
    v = np.asarray(v, np.uint64)
    s = (v * 747796405 + 2891336453) & 0xFFFFFFFF
    w = (((s >> ((s >> 28) + 4)) ^ s) * 277803737) & 0xFFFFFFFF
    return (((w >> 22) ^ w) & 0xFFFFFFFF).astype(U32)


_libm = ctypes.CDLL("libm.so.6")
for _n in ("sinf", "cosf", "tanf", "asinf", "acosf", "atanf", "expf", "logf", "atan2f"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float] * (2 if _n == "atan2f" else 1)


def t64(fn, *a):
    """the device's definition of a transcendental opcode: the host libm's f32 routine (dev_ops.hpp t_* = trans_libm.hpp, which
    restates it bit for bit); fn: the libm function's name"""
    a = [np.asarray(x, F32) for x in a]
    shape = np.broadcast(*a).shape
    a = [np.broadcast_to(x, shape).reshape(-1) for x in a]
    f = getattr(_libm, fn)
    return np.array([f(*[float(x[k]) for x in a]) for k in range(a[0].size)], F32).reshape(shape)


def rem_euclid(a, b):
    with np.errstate(all="ignore"):
        r = np.fmod(np.asarray(a, F32), np.asarray(b, F32)).astype(F32)
        return np.where(r < 0, (r + np.abs(b)).astype(F32), r).astype(F32)


TRANS = {"SIN": "sinf", "COS": "cosf", "TAN": "tanf", "ASIN": "asinf", "ACOS": "acosf", "ATAN": "atanf", "EXP": "expf", "LN": "logf"}


def trans_hooks(prog, prefix="fh_t_", v_base=128, sregs=tuple(range(86, 96)), window=26):
    """native stand-ins for the compiled routines embedded by gen_trans.py (the emulator has no f64 ISA): argument(s) in v<v_base> (,
    v<v_base + 1>), result in v<v_base> for the lanes in exec, return to s[96:97]; the routines' register window and scalar
    registers come back clobbered.  The four-sample routines (gen_trans.FUNCS4, where embedded): v<v_base> .. v<v_base + 3> in and out."""
    def mk(fn, nargs, nres=1):
        def hook(w):
            m = w._bits(w.exec)
            args = [w.v[v_base + k].view(F32).copy() for k in range(nargs)]
            if nres == 1:
                res = [fn(*args)]
            else:
                res = [fn(x) for x in args]
            for k, r in enumerate(res):
                w.v[v_base + k][m] = r.astype(F32).view(U32)[m]
            w.v[v_base + nres:v_base + window] = 0xDEADBEEF        # the routines may clobber their whole register window
            for sr in sregs:
                w.s[sr] = 0xDEADBEEF
            w.vcc = 0xDEADBEEFDEADBEEF
            return int(w.s[96]) | (int(w.s[97]) << 32)
        return hook
    h = {}
    for name, fn in TRANS.items():
        h[prog.symbols[prefix + name.lower()]] = mk(lambda x, fn=fn: t64(fn, x), 1)
        if prefix + name.lower() + "4" in prog.symbols:
            h[prog.symbols[prefix + name.lower() + "4"]] = mk(lambda x, fn=fn: t64(fn, x), 4, 4)
    h[prog.symbols[prefix + "atan2"]] = mk(lambda y, x: t64("atan2f", y, x), 2)
    h[prog.symbols[prefix + "mod"]] = mk(rem_euclid, 2)
    return h


def ref_f32(tape, inputs, n):
    """inputs: slot -> array[n]; returns {output slot: array}"""
    regs = {}
    out = {}
    qn = F32(np.nan)
    with np.errstate(all="ignore"):
        for w in tape:
            op, ro, ra, w1 = decode(w)
            imm = F32(u2f(w1))
            name = OPS[op]
            if name == "OUTPUT":
                out[w1] = regs[ra].copy()
                continue
            if name == "INPUT":
                regs[ro] = np.asarray(inputs[w1], F32).copy()
                continue
            if name == "COPY_REG":
                regs[ro] = regs[ra].copy()
                continue
            if name == "COPY_IMM":
                regs[ro] = np.full(n, imm, F32)
                continue
            base, form = split(op)
            if base is None and name in TRANS:
                regs[ro] = t64(TRANS[name], regs[ra])
                continue
            if base is None and name == "RAND":
                regs[ro] = (((pcg(regs[ra].view(U32)) >> U32(9)) | U32(0x3F800000)).view(F32) - F32(1)).astype(F32)
                continue
            if base is None:
                a = regs[ra]
                r = {"NEG": lambda: -a, "ABS": lambda: np.abs(a), "RECIP": lambda: F32(1) / a, "SQRT": lambda: np.sqrt(a),
                     "SQUARE": lambda: a * a, "FLOOR": lambda: np.floor(a), "CEIL": lambda: np.ceil(a), "ROUND": lambda: _round(a),
                     "NOT": lambda: (a == 0).astype(F32)}[name]()
                regs[ro] = r.astype(F32)
                continue
            if form == "RR":
                a, b = regs[ra], regs[w1]
            elif form == "RI":
                a, b = regs[ra], np.full(n, imm, F32)
            else:
                a, b = np.full(n, imm, F32), regs[ra]
            bn = BIN[base]
            un = np.isnan(a) | np.isnan(b)
            if bn == "ADD":
                r = a + b
            elif bn == "SUB":
                r = a - b
            elif bn == "MUL":
                r = a * b
            elif bn == "DIV":
                r = a / b
            elif bn == "COMPARE":
                r = _f_compare(a, b)
            elif bn == "ATAN2":
                r = t64("atan2f", a, b)
            elif bn == "MOD":
                r = rem_euclid(a, b)
            elif bn == "MIX":
                r = pcg((a.view(U32).astype(np.uint64) + pcg(b.view(U32)).astype(np.uint64)) & 0xFFFFFFFF).view(F32)
            elif bn == "MIN":
                r = np.where(a < b, a, np.where(b < a, b, np.where(un, qn, b)))
            elif bn == "MAX":
                r = np.where(a > b, a, np.where(b > a, b, np.where(un, qn, b)))
            elif bn == "AND":
                r = np.where(a == 0, a, b)
            elif bn == "OR":
                r = np.where(a != 0, a, b)
            else:
                raise NotImplementedError(bn)
            regs[ro] = r.astype(F32)
    return out


# ---- intervals -----------------------------------------------------------------------------------
def _rmin(a, b):
    return np.where(np.isnan(a), b, np.where(np.isnan(b), a, np.minimum(a, b))).astype(F32)


def _rmax(a, b):
    return np.where(np.isnan(a), b, np.where(np.isnan(b), a, np.maximum(a, b))).astype(F32)


def _nan(lo, hi):
    return np.isnan(lo) | np.isnan(hi)


def _iv_quadrant(x):
    """dev_ops.hpp iv_quadrant (interval.rs:143-147): rem_euclid(floorf(x * 2 / PI), 4) as u8, NaN -> 0"""
    PI = F32(3.14159274101257324)
    q = rem_euclid(np.floor(((x * F32(2)).astype(F32) / PI).astype(F32)).astype(F32), F32(4))
    return np.where(q > 0, q, F32(0)).astype(np.int64)


def _iv_trans(name, al, ah):
    """interval rules of the transcendental opcodes (types/interval.rs:136-302 as dev_ops.hpp iv_sincos .. iv_ln restates them)"""
    NAN = F32(np.nan)
    PI, TAU = F32(3.14159274101257324), F32(6.28318548202514648)
    f = lambda x: t64(TRANS[name], x)
    fl, fu = f(al), f(ah)
    if name in ("EXP", "ATAN"):
        return fl, fu
    if name == "LN":
        bad = al <= 0
        return np.where(bad, NAN, fl), np.where(bad, NAN, fu)
    if name in ("ASIN", "ACOS"):
        bad = (al < -1) | (ah > 1)
        lo, hi = (fl, fu) if name == "ASIN" else (fu, fl)
        return np.where(bad, NAN, lo), np.where(bad, NAN, hi)
    d = (ah - al).astype(F32)
    if name == "TAN":
        bad = (d >= PI) | ~(fu >= fl)
        return np.where(bad, NAN, fl), np.where(bad, NAN, fu)
    lq, uq = _iv_quadrant(al), _iv_quadrant(ah)
    if name == "COS":
        lq, uq = (lq + 1) & 3, (uq + 1) & 3
    n = len(al)
    lo, hi = np.full(n, -1, F32), np.full(n, 1, F32)
    dec_l, dec_u = (lq == 1) | (lq == 2), (uq == 1) | (uq == 2)
    small = ~(d >= PI)
    inc = (((lq == uq) & ~dec_l) | ((lq == 3) & (uq == 0))) & small
    dec = (((lq == uq) & dec_l) | ((lq == 1) & (uq == 2))) & small
    up, down = ~dec_l & dec_u, dec_l & ~dec_u
    lo = np.where(up, _rmin(fl, fu), lo); hi = np.where(down, _rmax(fl, fu), hi)
    lo = np.where(inc, fl, lo); hi = np.where(inc, fu, hi)
    lo = np.where(dec, fu, lo); hi = np.where(dec, fl, hi)
    single = al == ah
    lo = np.where(single, fl, lo); hi = np.where(single, fl, hi)
    full = (d >= TAU)
    lo = np.where(full, F32(-1), lo); hi = np.where(full, F32(1), hi)
    nn = np.isnan(al) | np.isnan(ah)
    return np.where(nn, NAN, lo).astype(F32), np.where(nn, NAN, hi).astype(F32)


def ref_interval(tape, inputs, n):
    """inputs: slot -> (lo[n], hi[n]).  Returns (result lo, hi, choices [n_choice_ops][n] u8, {slot: (lo, hi)} of all outputs)."""
    regs = {}
    res = (np.full(n, np.nan, F32), np.full(n, np.nan, F32))
    outs = {}
    choices = []
    NAN = F32(np.nan)
    with np.errstate(all="ignore"):
        for w in tape:
            op, ro, ra, w1 = decode(w)
            imm = F32(u2f(w1))
            name = OPS[op]
            if name == "OUTPUT":
                res = (regs[ra][0].copy(), regs[ra][1].copy())
                outs[w1] = res
                continue
            if name == "INPUT":
                lo, hi = inputs[w1]
                regs[ro] = (np.asarray(lo, F32).copy(), np.asarray(hi, F32).copy())
                continue
            if name == "COPY_REG":
                regs[ro] = (regs[ra][0].copy(), regs[ra][1].copy())
                continue
            if name == "COPY_IMM":
                regs[ro] = (np.full(n, imm, F32), np.full(n, imm, F32))
                continue
            base, form = split(op)
            if base is None:
                al, ah = regs[ra]
                if name == "NEG":
                    r = (-ah, -al)
                elif name == "ABS":
                    neg, pos = al < 0, ah > 0
                    r = (np.where(neg, np.where(pos, F32(0), -ah), al), np.where(neg, np.where(pos, _rmax(ah, -al), -al), ah))
                elif name == "RECIP":
                    ok = (al > 0) | (ah < 0)
                    r = (np.where(ok, F32(1) / ah, NAN), np.where(ok, F32(1) / al, NAN))
                elif name == "SQRT":
                    bad = al < 0
                    r = (np.where(bad, NAN, np.sqrt(al)), np.where(bad, NAN, np.sqrt(ah)))
                elif name == "SQUARE":
                    m = _rmax(np.abs(al), np.abs(ah))
                    lo = np.where(ah < 0, ah * ah, np.where(al > 0, al * al, np.where(_nan(al, ah), NAN, F32(0))))
                    hi = np.where(ah < 0, al * al, np.where(al > 0, ah * ah, np.where(_nan(al, ah), NAN, m * m)))
                    r = (lo, hi)
                elif name in ("FLOOR", "CEIL", "ROUND"):
                    f = {"FLOOR": np.floor, "CEIL": np.ceil, "ROUND": _round}[name]
                    r = (f(al), f(ah))
                elif name == "NOT":
                    contains = (al <= 0) & (ah >= 0)
                    zero = (al == 0) & (ah == 0)
                    first = ~contains & ~_nan(al, ah)
                    r = (np.where(first, F32(0), np.where(zero, F32(1), F32(0))), np.where(first, F32(0), F32(1)))
                elif name in TRANS:
                    r = _iv_trans(name, al, ah)
                elif name == "RAND":        # dev_ops.hpp iv_rand (interval.rs:619-627): one non-NaN bit pattern -> the point, else [0, 1]
                    point = ~_nan(al, ah) & (al.view(U32) == ah.view(U32))
                    v = (((pcg(al.view(U32)) >> U32(9)) | U32(0x3F800000)).view(F32) - F32(1)).astype(F32)
                    r = (np.where(point, v, F32(0)), np.where(point, v, F32(1)))
                else:
                    raise NotImplementedError(name)
                regs[ro] = (r[0].astype(F32), r[1].astype(F32))
                continue
            if form == "RR":
                (al, ah), (bl, bh) = regs[ra], regs[w1]
            elif form == "RI":
                (al, ah), (bl, bh) = regs[ra], (np.full(n, imm, F32), np.full(n, imm, F32))
            else:
                (al, ah), (bl, bh) = (np.full(n, imm, F32), np.full(n, imm, F32)), regs[ra]
            bn = BIN[base]
            nn = _nan(al, ah) | _nan(bl, bh)
            c = None
            if bn == "ADD":
                r = (al + bl, ah + bh)
            elif bn == "SUB":
                r = (al - bh, ah - bl)
            elif bn == "MUL" and form == "RI":
                neg = imm < 0
                bad = _nan(al, ah) | np.isnan(imm)
                r = (np.where(bad, NAN, ah * imm if neg else al * imm), np.where(bad, NAN, al * imm if neg else ah * imm))
            elif bn == "MUL":
                p = [al * bl, al * bh, ah * bl, ah * bh]
                lo = _rmin(_rmin(_rmin(p[0], p[1]), p[2]), p[3])
                hi = _rmax(_rmax(_rmax(p[0], p[1]), p[2]), p[3])
                r = (np.where(nn, NAN, lo), np.where(nn, NAN, hi))
            elif bn == "DIV":
                ok = ((bl > 0) | (bh < 0)) & ~_nan(al, ah)
                q = [al / bl, al / bh, ah / bl, ah / bh]
                lo = _rmin(_rmin(_rmin(q[0], q[1]), q[2]), q[3])
                hi = _rmax(_rmax(_rmax(q[0], q[1]), q[2]), q[3])
                r = (np.where(ok, lo, NAN), np.where(ok, hi, NAN))
            elif bn == "COMPARE":
                less, greater = ah < bl, al > bh
                eq = (al == ah) & (bl == bh) & (al == bl)
                lo = np.where(nn, NAN, np.where(less, F32(-1), np.where(greater, F32(1), np.where(eq, F32(0), F32(-1)))))
                hi = np.where(nn, NAN, np.where(less, F32(-1), np.where(greater, F32(1), np.where(eq, F32(0), F32(1)))))
                r = (lo, hi)
            elif bn == "MIX":           # iv_mix (interval.rs:600-616): both one non-NaN bit pattern -> the point hash(a + hash(b)), else NaN
                al, ah, bl, bh = (np.ascontiguousarray(v, F32) for v in (al, ah, bl, bh))
                point = ~nn & (al.view(U32) == ah.view(U32)) & (bl.view(U32) == bh.view(U32))
                v = pcg((al.view(U32) + pcg(bl.view(U32))).astype(U32)).view(F32)
                r = (np.where(point, v, NAN), np.where(point, v, NAN))
            elif bn == "MOD":           # iv_rem_euclid (interval.rs:485-503)
                bad = nn | ((bl <= 0) & (bh >= 0))
                x, y = (al / bl).astype(F32), (ah / bl).astype(F32)
                same = (bl == bh) & (bl > 0) & (x != np.floor(x)) & (np.floor(x) == np.floor(y))
                top = np.where(bl < 0, -bl, bh)
                r = (np.where(bad, NAN, np.where(same, rem_euclid(al, bl), F32(0))), np.where(bad, NAN, np.where(same, rem_euclid(ah, bl), top)))
            elif bn == "ATAN2":         # iv_atan2 (interval.rs:541-597): a = y, b = x
                PI = F32(3.14159274101257324)
                ypos, xpos = al >= 0, bl >= 0
                yneg, xneg = ~ypos & (ah <= 0), ~xpos & (bh <= 0)
                y0 = np.where(ypos, np.where(xpos, ah, al), np.where(yneg, np.where(xpos, al, ah), al))
                y1 = np.where(ypos, np.where(xneg, ah, al), np.where(yneg, np.where(xneg, al, ah), ah))
                x1 = np.where(ypos | yneg, bh, bl)
                v0, v1 = t64("atan2f", y0, bl), t64("atan2f", y1, x1)
                full = (al <= 0) & (ah >= 0) & (bl < 0)
                r = (np.where(nn, NAN, np.where(full, -PI, _rmin(v0, v1))), np.where(nn, NAN, np.where(full, PI, _rmax(v0, v1))))
            elif bn == "MIN":
                c = np.where(nn, 3, np.where(ah < bl, 1, np.where(bh < al, 2, 3)))
                r = (np.where(nn, NAN, _rmin(al, bl)), np.where(nn, NAN, _rmin(ah, bh)))
            elif bn == "MAX":
                c = np.where(nn, 3, np.where(al > bh, 1, np.where(bl > ah, 2, 3)))
                r = (np.where(nn, NAN, _rmax(al, bl)), np.where(nn, NAN, _rmax(ah, bh)))
            elif bn == "AND":
                zero = (al == 0) & (ah == 0)
                contains = (al <= 0) & (ah >= 0)
                c = np.where(nn, 3, np.where(zero, 1, np.where(~contains, 2, 3)))
                lo = np.where(nn, NAN, np.where(zero, F32(0), np.where(~contains, bl, _rmin(bl, F32(0)))))
                hi = np.where(nn, NAN, np.where(zero, F32(0), np.where(~contains, bh, _rmax(bh, F32(0)))))
                r = (lo, hi)
            elif bn == "OR":
                zero = (al == 0) & (ah == 0)
                contains = (al <= 0) & (ah >= 0)
                c = np.where(nn, 3, np.where(~contains, 1, np.where(zero, 2, 3)))
                lo = np.where(nn, NAN, np.where(~contains, al, np.where(zero, bl, _rmin(al, bl))))
                hi = np.where(nn, NAN, np.where(~contains, ah, np.where(zero, bh, _rmax(ah, bh))))
                r = (lo, hi)
            else:
                raise NotImplementedError(bn)
            regs[ro] = (np.asarray(r[0], F32), np.asarray(r[1], F32))
            if c is not None:
                choices.append(np.asarray(c, np.uint8))
    ch = np.array(choices, dtype=np.uint8).reshape(len(choices), n)
    return res[0], res[1], ch, outs


# ---- prune (kernels.hip prune_sweep<true>) -----------------------------------------------------------
def ref_prune(tape, choices):
    """choices: one value per choice op, evaluation order.  Returns (ops list, n_regs, kept choices)."""
    DEAD = -1
    m = {}
    free = []          # pool: lowest free first
    high = [0]
    used = set()

    def take():
        r = 0
        while r in used:
            r += 1
        used.add(r)
        high[0] = max(high[0], r + 1)
        return r

    def give(r):
        used.discard(r)

    def use(r):
        if m.get(r, DEAD) == DEAD:
            m[r] = take()
        return m[r]

    rev = []
    ci = len(choices)
    kept = 0
    for w in reversed(list(tape)):
        op, ro, ra, w1 = decode(w)
        c = 3
        if is_choice(op):
            ci -= 1
            c = int(choices[ci])
        if op == 0:
            rev.append(pack(0, 0, use(ra), w1))
            continue
        no = m.get(ro, DEAD)
        if no == DEAD:
            continue
        m[ro] = DEAD
        alias, copy_imm = None, False
        if op == 2:
            alias = ra
        elif is_choice(op) and c == 1:
            alias = ra
        elif is_choice(op) and c == 2:
            if is_rr(op):
                alias = w1
            else:
                copy_imm = True
        if alias is not None:
            if m.get(alias, DEAD) == DEAD:
                m[alias] = no
                continue
            give(no)
            rev.append(pack(2, no, m[alias], 0))
            continue
        give(no)
        if copy_imm:
            rev.append(pack(3, no, 0, w1))
            continue
        na = nb = 0
        if op not in (1, 3):
            na = use(ra)
        if is_rr(op):
            nb = use(w1)
        if is_choice(op):
            kept += 1
        rev.append(pack(op, no, na, nb if is_rr(op) else w1))
    return rev[::-1], high[0], kept


# ---- device state in emulator memory -------------------------------------------------------------------
class Blob:
    """a zeroed struct image with typed pokes at byte offsets"""

    def __init__(self, size):
        self.b = np.zeros(size, dtype=np.uint8)

    def u32(self, off, v):
        self.b[off:off + 4] = np.frombuffer(struct.pack("<I", int(v) & 0xFFFFFFFF), np.uint8)

    def u64(self, off, v):
        self.b[off:off + 8] = np.frombuffer(struct.pack("<Q", int(v)), np.uint8)

    def f32(self, off, v):
        self.b[off:off + 4] = np.frombuffer(struct.pack("<f", float(v)), np.uint8)

    def arr(self, off, a):
        a = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        self.b[off:off + len(a)] = a

    def get_u32(self, off, n=1):
        return self.b[off:off + 4 * n].view(U32).copy()


def shape_tape(shape):
    """device-format ops (uint64 array) of a fidget_amd.Shape (host only, no GPU)"""
    import fidget_amd as F
    n = shape.size()
    w = np.zeros(max(n, 1), dtype=np.uint64)
    F.lib().fhip_tape_ops(shape._h, w.ctypes.data_as(F.C.c_void_p), n)
    return w[:n]
