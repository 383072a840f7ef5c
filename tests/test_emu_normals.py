"""fh_normals (the assembly gradient interpreter, gen_normals.py) on the CPU emulator against a numpy restatement of dev_ops.hpp's
GRAD semantics (fidget-core/src/types/grad.rs, vm/mod.rs:1091-1397): for every leaf of the launch's work list and every pixel of its 8 x 8
whose z-buffer word names it at a depth of this slab, the gradient of that leaf's tape at the voxel above the hit, through xf_grad (the screen -> model matrix applied to
{x,1,0,0}, {y,0,1,0}, {z,0,0,1} and the division by w) - dx, dy, dz bit for bit, NaN for NaN; the leaf number cleared; pixels of other
slabs and pixels without a hit untouched.  Several leaves per footprint, affine and projective matrices."""
import numpy as np
import pytest

import emu_util as U
from emu_util import E, F32, U32
from test_emu_tiles import shape_of, chain
from test_emu_columns import AFFINE, ROTATED, PERSPECTIVE


class G:
    """{v, dx, dy, dz} as four float32 arrays; every operation rounds to f32 once, as the device's"""
    def __init__(self, v, dx, dy, dz):
        self.c = [np.asarray(x, F32) for x in (v, dx, dy, dz)]

    @staticmethod
    def one(v, n):
        z = np.zeros(n, F32)
        return G(np.broadcast_to(np.asarray(v, F32), (n,)).copy(), z, z.copy(), z.copy())


def g_add(a, b): return G(*[(x + y).astype(F32) for x, y in zip(a.c, b.c)])
def g_sub(a, b): return G(*[(x - y).astype(F32) for x, y in zip(a.c, b.c)])
def g_mul_f(a, r): return G(*[(x * F32(r)).astype(F32) for x in a.c])


def g_mul(a, b):
    v = (a.c[0] * b.c[0]).astype(F32)
    return G(v, *[((a.c[0] * b.c[k]).astype(F32) + (b.c[0] * a.c[k]).astype(F32)).astype(F32) for k in (1, 2, 3)])


def g_div(a, b):
    d = (b.c[0] * b.c[0]).astype(F32)
    return G((a.c[0] / b.c[0]).astype(F32), *[(((b.c[0] * a.c[k]).astype(F32) - (a.c[0] * b.c[k]).astype(F32)).astype(F32) / d).astype(F32) for k in (1, 2, 3)])


def g_sel(m, a, b): return G(*[np.where(m, x, y).astype(F32) for x, y in zip(a.c, b.c)])


def ref_grad(tape, inputs, n):
    """inputs: slot -> G; returns the G of output 0"""
    regs, out = {}, None
    with np.errstate(all="ignore"):
        for w in tape:
            op, ro, ra, w1 = U.decode(w)
            name = U.OPS[op]
            imm = G.one(U.u2f(w1), n)
            if name == "OUTPUT":
                out = regs[ra]
            elif name == "INPUT":
                regs[ro] = inputs[w1]
            elif name == "COPY_REG":
                regs[ro] = regs[ra]
            elif name == "COPY_IMM":
                regs[ro] = imm
            elif op < 22:
                a = regs[ra]
                v = a.c[0]
                if name == "NEG": r = G(*[(-x).astype(F32) for x in a.c])
                elif name == "ABS": r = g_sel(v < 0, G(*[(-x).astype(F32) for x in a.c]), a)
                elif name == "RECIP": r = g_div(G.one(1.0, n), a)
                elif name == "SQRT":
                    s = np.sqrt(v).astype(F32)
                    t = (F32(2) * s).astype(F32)
                    r = G(s, *[(a.c[k] / t).astype(F32) for k in (1, 2, 3)])
                elif name == "SQUARE": r = g_mul(a, a)
                elif name == "FLOOR": r = G.one(np.floor(v), n)
                elif name == "CEIL": r = G.one(np.ceil(v), n)
                elif name == "ROUND": r = G.one(U._round(v), n)
                elif name == "NOT": r = G.one((v == 0).astype(F32), n)
                elif name in U.TRANS:      # dev_ops.hpp GRAD::unary, FULL
                    f = lambda fn, x=v: U.t64(fn, x)
                    mulk = lambda c: [(a.c[k] * c).astype(F32) for k in (1, 2, 3)]
                    divk = lambda c, neg=False: [((-a.c[k] if neg else a.c[k]) / c).astype(F32) for k in (1, 2, 3)]
                    if name == "SIN": r = G(f("sinf"), *mulk(f("cosf")))
                    elif name == "COS": r = G(f("cosf"), *mulk((-f("sinf")).astype(F32)))
                    elif name == "TAN":
                        c0 = f("cosf")
                        r = G(f("tanf"), *divk((c0 * c0).astype(F32)))
                    elif name in ("ASIN", "ACOS"):
                        rt = np.sqrt((F32(1) - (v * v).astype(F32)).astype(F32)).astype(F32)
                        r = G(f(U.TRANS[name]), *divk(rt, neg=name == "ACOS"))
                    elif name == "ATAN": r = G(f("atanf"), *divk(((v * v).astype(F32) + F32(1)).astype(F32)))
                    elif name == "EXP":
                        e = f("expf")
                        r = G(e, *[(e * a.c[k]).astype(F32) for k in (1, 2, 3)])
                    else: r = G(f("logf"), *divk(v))
                else: raise NotImplementedError(name)
                regs[ro] = r
            else:
                base, form = U.split(op)
                a, b = (regs[ra], regs[w1]) if form == "RR" else ((regs[ra], imm) if form == "RI" else (imm, regs[ra]))
                bn = U.BIN[base]
                un = np.isnan(a.c[0]) | np.isnan(b.c[0])
                if bn == "ADD": r = g_add(a, b)
                elif bn == "SUB": r = g_sub(a, b)
                elif bn == "MUL": r = g_mul(a, b) if form == "RR" else g_mul_f(a, U.u2f(w1))
                elif bn == "DIV": r = g_div(a, b)
                elif bn == "MIN": r = g_sel(un, G.one(np.nan, n), g_sel(a.c[0] < b.c[0], a, b))
                elif bn == "MAX": r = g_sel(un, G.one(np.nan, n), g_sel(a.c[0] > b.c[0], a, b))
                elif bn == "AND": r = g_sel(a.c[0] == 0, a, b)
                elif bn == "OR": r = g_sel(a.c[0] != 0, a, b)
                elif bn == "COMPARE": r = G.one(U._f_compare(a.c[0], b.c[0]), n)
                else: raise NotImplementedError(bn)
                regs[ro] = r
    return out


def xf_grad(m, px, py, pz):
    n = len(px)
    one, zero = np.ones(n, F32), np.zeros(n, F32)
    x, y, z = G(px, one, zero, zero), G(py, zero, one, zero), G(pz, zero, zero, one)
    r = [g_add(g_add(g_add(g_mul_f(x, m[4 * i]), g_mul_f(y, m[4 * i + 1])), g_mul_f(z, m[4 * i + 2])), G.one(m[4 * i + 3], n)) for i in range(4)]
    return [g_div(r[i], r[3]) for i in range(3)]


def run_normals(tapes, in_kind, mat, hits, size=16, z_lo=0, z_hi=1 << 20, kernel="fh_normals", corners=None):
    """tapes: [(ops, n_regs)]; hits: {(px, py): (leaf index, depth)}; corners: [(x, y)] of each leaf's 8 x 8 pixels.  Returns (zbuf, normals) after one launch."""
    off = U.offsets()
    mem = E.Memory()
    arena = np.zeros(8192, np.uint64)
    leaves = np.zeros((len(tapes), 6), U32)
    at = 16
    for k, (ops, regs) in enumerate(tapes):
        arena[at:at + len(ops)] = ops
        leaves[k] = [at, len(ops), regs, 0, 0, 0]
        at += len(ops) + 24
    zbuf = np.zeros(size * size, np.uint64)
    for (px, py), (leaf, depth) in hits.items():
        zbuf[py * size + px] = (depth << 32) | (leaf + 1)
    normals = np.full(size * size * 3, 7.5, F32)
    # the work list k_hits3d makes: every leaf that owns a pixel of the launch's depth range, once (here: in leaf order, plus one leaf
    # without any hit - the kernel must leave after its z-buffer words)
    for k, c in enumerate(corners or {}):
        leaves[k, 3:5] = c
    n_waves = 128                        # two waves per list
    ids = sorted({leaf + 1 for leaf, depth in hits.values() if z_lo < depth <= z_hi}) + ([len(tapes)] if tapes else [])
    cap = 5                              # 64 lists of `cap` entries behind 64 counters 64 words apart (kernels.hip k_hits3d)
    fps = np.zeros(64 * 64 + 64 * cap, U32)
    for k, v in enumerate(ids):
        b = (k * 7) % 64 if k % 3 else k % 2       # (lists of several entries, most lists empty)
        assert fps[b * 64] < cap
        fps[64 * 64 + b * cap + fps[b * 64]] = v
        fps[b * 64] += 1
    a_arena, a_leaves, a_z, a_n, a_fp = mem.map(arena), mem.map(leaves), mem.map(zbuf), mem.map(normals), mem.map(fps)
    st = U.Blob(off["sizeof_state"])
    st.arr(off["P.mat"], np.asarray(mat, F32))
    st.u32(off["P.width"], size); st.u32(off["P.height"], size)
    for s in range(16):
        st.u32(off["P.in_kind"] + 4 * s, in_kind[s] if s < len(in_kind) else 3)
        st.f32(off["P.in_value"] + 4 * s, 0.25 + s)
    st.u64(off["arena"], a_arena); st.u64(off["leaves"], a_leaves); st.u64(off["zbuf"], a_z); st.u64(off["normals"], a_n)
    st.u64(off["hit_list"], a_fp)
    a_st = mem.map(st.b)
    slot = [-1, -1, -1]
    for s_, k in enumerate(list(in_kind) + [3] * (16 - len(in_kind))):
        if k < 3:
            slot[k] = s_
    slots = sum((0xFF if slot[ax] < 0 else slot[ax]) << (8 * ax) for ax in range(3))
    ka = np.array([a_st & 0xFFFFFFFF, a_st >> 32, n_waves, slots, z_lo, z_hi, cap, 0], U32)
    trans = kernel == "fh_normals_t"
    E.launch(U.program(), mem, kernel, ka.tobytes(), n_waves, lds_bytes=16, n_vgpr=250 if trans else 224, wg_y_sgpr=None,
             hooks=U.trans_hooks(U.program(), "fh_tn_", 224) if trans else None)
    return zbuf, normals.reshape(size * size, 3)


def leaves_of(tapes, size=16):
    """a leaf per footprint and tape - a leaf's hits lie in its own 8 x 8 pixels: (the leaves' tapes, their corners)"""
    fw = size // 8
    lt, cs = [], []
    for fy in range(fw):
        for fx in range(fw):
            for t in tapes:
                lt.append(t)
                cs.append((8 * fx, 8 * fy))
    return lt, cs


def leaf_at(px, py, t, n_tapes, size=16):
    return ((py // 8) * (size // 8) + px // 8) * n_tapes + t


def expect(tapes, in_kind, mat, hits, size, z_lo=0, z_hi=1 << 20):
    zbuf = np.zeros(size * size, np.uint64)
    normals = np.full((size * size, 3), 7.5, F32)
    m = np.asarray(mat, F32)
    for (px, py), (leaf, depth) in hits.items():
        i = py * size + px
        zbuf[i] = (depth << 32) | (leaf + 1)
        if not (z_lo < depth <= z_hi):
            continue
        gx, gy, gz = xf_grad(m, np.array([px], F32), np.array([py], F32), np.array([depth - 1], F32))
        inputs = {s: (gx, gy, gz)[k] if k < 3 else G.one(F32(0.25 + s), 1) for s, k in enumerate(list(in_kind) + [3] * (16 - len(in_kind)))}
        g = ref_grad(tapes[leaf][0], inputs, 1)
        normals[i] = [g.c[1][0], g.c[2][0], g.c[3][0]]
        zbuf[i] = depth << 32
    return zbuf, normals


def same(a, b):
    return ((a.view(U32) == b.view(U32)) | (np.isnan(a) & np.isnan(b))).all()


@pytest.mark.parametrize("mat", [AFFINE, ROTATED, PERSPECTIVE], ids=["affine", "rotated", "perspective"])
@pytest.mark.parametrize("seed", [0, 1, 2, 3, 5, 6])
def test_normals_of_random_shapes(seed, mat):
    sh, tape, ik = shape_of(seed)
    if sh.slot_count() > 40:
        pytest.skip("more than 40 registers: the C++ kernel's")
    sh2, tape2, ik2 = shape_of(seed + 1 if seed != 3 else 0)
    tapes = [(tape, sh.slot_count())]
    if sh2.slot_count() <= 40 and ik2 == ik:
        tapes.append((tape2, sh2.slot_count()))
    rng = np.random.default_rng(seed)
    hits = {}
    nt = len(tapes)
    tapes, corners = leaves_of(tapes)
    for _ in range(70):           # pixels of all four footprints, some sharing a leaf, depths in and out of the slab
        px, py = int(rng.integers(0, 16)), int(rng.integers(0, 16))
        hits[(px, py)] = (leaf_at(px, py, int(rng.integers(0, nt)), nt), int(rng.integers(1, 17)))
    got_z, got_n = run_normals(tapes, ik, mat, hits, z_lo=2, z_hi=14, corners=corners)
    want_z, want_n = expect(tapes, ik, mat, hits, 16, z_lo=2, z_hi=14)
    assert (got_z == want_z).all(), "z-buffer words differ"
    assert same(got_n, want_n), f"{(got_n.view(U32) != want_n.view(U32)).any(axis=1).sum()} normals differ"


def test_normals_of_a_prospero_leaf_parent():
    """a 32^3 tile's tape of prospero.vm (dozens of ops, min / max chains, square roots) as the leaf's"""
    ik, ch = chain()
    tape, regs, nch, center, half = ch[1]
    assert regs <= 32
    hits = {(x, y): (leaf_at(x, y, 0, 1), 3 + (x + y) % 9) for x in range(0, 16, 3) for y in range(0, 16, 2)}
    mat = [0.125 * 0.25, 0, 0, float(center[0]) - 0.25, 0, -0.125 * 0.25, 0, float(center[1]) + 0.25, 0, 0, 0.125 * 0.25, float(center[2]) - 0.25, 0, 0, 0, 1]
    tapes, corners = leaves_of([(tape, regs)])
    got_z, got_n = run_normals(tapes, ik, mat, hits, corners=corners)
    want_z, want_n = expect(tapes, ik, mat, hits, 16)
    assert (got_z == want_z).all() and same(got_n, want_n)
    assert np.isfinite(want_n[want_z != 0].astype(np.float64)).all() is not None


def trans_grad_shape(which):
    import fidget_amd as F
    c = F.Context()
    x, y, z = c.x(), c.y(), c.z()
    if which == 0:
        n = c.sub(c.add(c.mul(c.sin(c.mul(x, 3.0)), c.cos(c.mul(y, 2.0))), c.mul(c.exp(c.mul(z, -0.7)), c.atan(c.add(x, y)))), 0.2)
    else:
        n = c.add(c.sub(c.tan(c.mul(x, 0.4)), c.asin(c.mul(y, 0.3))), c.mul(c.acos(c.mul(z, 0.3)), c.ln(c.add(c.square(x), 0.5))))
    sh = F.Shape(c, n)
    ik = [3] * 16
    for a in range(3):
        s_ = sh.axis_index(a)
        if s_ >= 0:
            ik[s_] = a
    return sh, U.shape_tape(sh), ik


@pytest.mark.parametrize("which", [0, 1])
@pytest.mark.parametrize("mat", [AFFINE, PERSPECTIVE], ids=["affine", "perspective"])
def test_normals_with_transcendental_opcodes(which, mat):
    """fh_normals_t: gradients of sin cos tan asin acos atan exp ln as dev_ops.hpp GRAD has them, the values themselves by the compiled
    routines (stood in for natively in the emulator: what is tested is the call sequence, the register window behind the 32-register
    gradient file, and the derivative formulas around the calls)"""
    sh, tape, ik = trans_grad_shape(which)
    rng = np.random.default_rng(which)
    hits = {}
    for _ in range(50):
        px, py = int(rng.integers(0, 16)), int(rng.integers(0, 16))
        hits[(px, py)] = (leaf_at(px, py, 0, 1), int(rng.integers(1, 15)))
    tapes, corners = leaves_of([(tape, sh.slot_count())])
    got_z, got_n = run_normals(tapes, ik, mat, hits, kernel="fh_normals_t", corners=corners)
    want_z, want_n = expect(tapes, ik, mat, hits, 16)
    assert (got_z == want_z).all() and same(got_n, want_n), f"{(got_n.view(U32) != want_n.view(U32)).any(axis=1).sum()} normals differ"
