"""Tape parallelism (host_graph.hpp split_root / plan_terms): the independent sub-tapes of a root
min / max, combined in order, are the same function as the whole tape - bit for bit, for points and
intervals; and the 3D renderer, which evaluates its root level that way (the root tree's terms by
independent groups, then the tree), draws the same image."""
import numpy as np
import pytest

from conftest import model_path


def _fmin(a, b):
    """types/float.rs:93-108: a < b -> a; b < a -> b; else NaN if either is NaN, else b"""
    out = np.where(a < b, a, b)
    return np.where(np.isnan(a) | np.isnan(b), np.float32(np.nan), out).astype(np.float32)


def test_split_exists_for_prospero():
    import fidget_amd as F
    s = F.Shape.from_vm(model_path("prospero.vm"))
    op, gs = s.groups()
    assert op == "min" and 2 <= len(gs) <= 32           # (FH_MAX_GROUPS)
    assert max(g.size() for g in gs) < s.size() // 4          # the point: short independent chains
    assert F.Shape.from_vm(model_path("hi.vm")).groups() == ("", [])


@pytest.mark.gpu
def test_groups_points_bit_exact():
    import fidget_amd as F
    s = F.Shape.from_vm(model_path("prospero.vm"))
    op, gs = s.groups()
    rng = np.random.default_rng(7)
    n = 100_000
    x, y, z = (rng.uniform(-1, 1, n).astype(np.float32) for _ in range(3))
    want = s.eval_float_slice(x, y, z)
    acc = None
    for g in gs:
        v = g.eval_float_slice(x, y, z)
        acc = v if acc is None else _fmin(acc, v)
    assert (acc.view(np.uint32) == np.asarray(want).view(np.uint32)).all()


@pytest.mark.gpu
def test_groups_intervals_bit_exact():
    import fidget_amd as F
    s = F.Shape.from_vm(model_path("prospero.vm"))
    op, gs = s.groups()
    rng = np.random.default_rng(11)
    boxes = []
    for _ in range(2000):
        c = rng.uniform(-1, 1, 3)
        h = rng.uniform(0.001, 0.3, 3)
        boxes.append([(float(np.float32(c[i] - h[i])), float(np.float32(c[i] + h[i]))) for i in range(3)])
    want = [o for o, _ in s.eval_interval_batch(boxes)]
    parts = [[o for o, _ in g.eval_interval_batch(boxes)] for g in gs]
    for i, w in enumerate(want):
        lo = hi = None
        nan = False
        for p in parts:
            a, b = np.float32(p[i][0]), np.float32(p[i][1])
            nan |= bool(np.isnan(a) or np.isnan(b))
            lo = a if lo is None else min(lo, a)
            hi = b if hi is None else min(hi, b)
        if nan:
            assert np.isnan(w[0]) and np.isnan(w[1])
        else:
            assert (np.float32(w[0]), np.float32(w[1])) == (lo, hi), (i, w, lo, hi)


def test_term_plan_for_prospero():
    """The renderer's form of the split: every choice of the full tape has a source, the tree is a chain."""
    import fidget_amd as F
    s = F.Shape.from_vm(model_path("prospero.vm"))
    p = s.term_plan()
    assert 2 <= p["groups"] <= 32 and p["terms"] >= 600
    assert p["choices"] == s.choice_count()
    assert p["tree_regs"] == 1 and p["tree_ops"] >= p["terms"] - 1
    assert F.Shape.from_vm(model_path("hi.vm")).term_plan()["groups"] == 0


_SMALL_TREES = r"""
import os, sys
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["FHIP_GROUPS_MIN_OPS"] = "0"      # every shape with a min / max root goes down the grouped path
os.environ["FHIP_GROUPS_MIN_TERMS"] = "2"
import numpy as np, fidget_amd as F, oracle as O
from test_render_random import build

def balanced(ctx, op, n, seed):
    import random
    rng = random.Random(seed)
    x, y, z = ctx.x(), ctx.y(), ctx.z()
    def ball():
        cx, cy, cz, r = (rng.uniform(-0.7, 0.7) for _ in range(3)), None, None, rng.uniform(0.15, 0.4)
        cx = list(cx)
        d = ctx.add(ctx.add(ctx.square(ctx.sub(x, cx[0])), ctx.square(ctx.sub(y, cx[1]))), ctx.square(ctx.sub(z, cx[2])))
        return ctx.sub(ctx.sqrt(d), r)
    level = [ball() for _ in range(n)]
    shared = level[1]
    while len(level) > 1:   # a balanced tree (not a chain), one part used twice
        level = [getattr(ctx, op)(level[i], level[i + 1]) if i + 1 < len(level) else level[i] for i in range(0, len(level), 2)]
    return getattr(ctx, op)(level[0], ctx.add(shared, 0.05))

def check(fa, fb, n, what):
    a = F.render3d(fa, n)[0]
    b = O.render3d(fb, n)[0]
    assert (a["depth"] == b["depth"]).all(), (what, n, int((a["depth"] != b["depth"]).sum()))
    same = (a["normal"] == b["normal"]) | (np.isnan(a["normal"]) & np.isnan(b["normal"]))   # add / square / sqrt only: bit-exact
    assert same.all(), (what, n, int((~same).any(axis=2).sum()))

split = 0
for seed in range(12):
    cf, co = F.Context(), O.Context()
    sf = F.Shape(cf, build(cf, seed))
    split += sf.term_plan()["groups"] > 0
    check(sf, O.Shape(co, build(co, seed)), 128, f"random {seed}")
assert split >= 6, split
for op, n in (("min", 13), ("max", 9), ("min", 40)):
    cf, co = F.Context(), O.Context()
    sf = F.Shape(cf, balanced(cf, op, n, 5))
    p = sf.term_plan()
    assert p["groups"] > 0 and p["tree_regs"] > 1, p      # not a chain: the op-by-op tree kernel
    # a max of balls is mostly empty space: still the same image
    check(sf, O.Shape(co, balanced(co, op, n, 5)), 128, f"balanced {op} {n}")
print("ok")
"""


@pytest.mark.gpu
def test_render3d_small_trees_through_groups():
    """Random CSG shapes and balanced (non-chain) trees, with the thresholds lowered so that all of them
    take the grouped root level: same image as the oracle."""
    import subprocess, sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + _SMALL_TREES], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


# ---- plan_terms on the CPU: a point interpreter of the device tape format in numpy f32 -------------------
def _run_tape(ops, pts):
    """Evaluate a device tape (Shape.ops()) at points pts[n, 3] (x, y, z in input slot order 0, 1, 2).
    Returns ({output slot: f32[n]}, [choice arrays in tape order]); choices as the f32 evaluators record
    them: 1 = left, 2 = right, 3 = both (tie or NaN).  Only the opcodes prospero uses."""
    f = np.float32
    r, outs, choices = {}, {}, []
    def minmax(a, b, is_min):
        lt = (a < b) if is_min else (a > b)
        gt = (b < a) if is_min else (b > a)
        nan = np.isnan(a) | np.isnan(b)
        v = np.where(lt, a, np.where(gt, b, np.where(nan, f(np.nan), b))).astype(f)
        choices.append(np.where(lt, 1, np.where(gt, 2, 3)).astype(np.uint8))
        return v
    with np.errstate(all="ignore"):
        for name, out, a, b, imm in ops:
            iv = np.full(len(pts), np.array([imm], np.uint32).view(f)[0], f)
            if name == "Output": outs[imm] = r[a]
            elif name == "Input": r[out] = pts[:, imm].astype(f)
            elif name == "CopyReg": r[out] = r[a]
            elif name == "CopyImm": r[out] = iv
            elif name == "Neg": r[out] = -r[a]
            elif name == "Abs": r[out] = np.abs(r[a])
            elif name == "Square": r[out] = r[a] * r[a]
            elif name == "Sqrt": r[out] = np.sqrt(r[a])
            elif name == "AddRR": r[out] = r[a] + r[b]
            elif name == "SubRR": r[out] = r[a] - r[b]
            elif name == "MulRR": r[out] = r[a] * r[b]
            elif name == "AddRI": r[out] = r[a] + iv
            elif name == "SubRI": r[out] = r[a] - iv
            elif name == "MulRI": r[out] = r[a] * iv
            elif name == "SubIR": r[out] = iv - r[a]
            elif name == "Recip": r[out] = (f(1.0) / r[a]).astype(f)
            elif name == "Floor": r[out] = np.floor(r[a])
            elif name == "Ceil": r[out] = np.ceil(r[a])
            elif name == "Round": r[out] = (np.trunc(r[a]) + np.where(np.abs(r[a] - np.trunc(r[a])) >= f(0.5), np.copysign(f(1), r[a]), f(0))).astype(f)
            elif name == "Not": r[out] = np.where(r[a] == 0, f(1), f(0)).astype(f)
            elif name in ("DivRR", "DivRI", "DivIR"):
                x, y = (r[a], r[b]) if name == "DivRR" else ((r[a], iv) if name == "DivRI" else (iv, r[a]))
                r[out] = (x / y).astype(f)
            elif name in ("CompareRR", "CompareRI", "CompareIR"):
                x, y = (r[a], r[b]) if name == "CompareRR" else ((r[a], iv) if name == "CompareRI" else (iv, r[a]))
                r[out] = np.where(x < y, f(-1), np.where(x == y, f(0), np.where(x > y, f(1), f(np.nan)))).astype(f)
            elif name in ("AndRR", "AndRI", "OrRR", "OrRI"):
                y = r[b] if name.endswith("RR") else iv
                first = (r[a] == 0) if name.startswith("And") else (r[a] != 0)
                r[out] = np.where(first, r[a], y).astype(f)
                choices.append(np.where(first, 1, 2).astype(np.uint8))
            elif name == "MinRR": r[out] = minmax(r[a], r[b], True)
            elif name == "MaxRR": r[out] = minmax(r[a], r[b], False)
            elif name == "MinRI": r[out] = minmax(r[a], iv, True)
            elif name == "MaxRI": r[out] = minmax(r[a], iv, False)
            else: raise NotImplementedError(name)
    return outs, choices


def _check_plan(s, seed=3):
    """the groups' terms folded by the tree are the root tape's value bit for bit, and every choice of the
    root tape is found where the plan says it is recorded (point semantics, numpy f32)"""
    groups, tree, src = s.term_parts()
    assert len(groups) >= 2 and len(src) == s.choice_count()
    rng = np.random.default_rng(seed)
    pts = np.concatenate([rng.uniform(-1, 1, (48, 3)), [[0, 0, 0], [0.5, -0.25, 0.1], [-1, 1, 0]]]).astype(np.float32)
    want, want_ch = _run_tape(s.ops(), pts)
    assert len(want_ch) == len(src)
    terms, group_ch = {}, []
    for g in groups:
        o, ch = _run_tape(g.ops(), pts)
        assert not (set(o) & set(terms))           # a term belongs to one group
        terms.update(o)
        group_ch.append(ch)
    assert sorted(terms) == list(range(s.term_plan()["terms"]))
    # the tree over the terms
    regs, tree_ch = {}, []
    def operand(kind, ref):
        if kind == 0: return regs[ref]
        if kind == 1: return terms[ref]
        return np.full(len(pts), np.array([ref], np.uint32).view(np.float32)[0], np.float32)
    for w0, a, b in tree:
        op, out, ak, bk = int(w0) & 0xFF, (int(w0) >> 8) & 0xFF, (int(w0) >> 16) & 0xFF, int(w0) >> 24
        assert op in (30, 31, 42, 43)             # MIN / MAX, reg,reg or reg,imm
        va, vb = operand(ak, int(a)), operand(bk, int(b))
        is_min = op in (30, 42)
        lt = (va < vb) if is_min else (va > vb)
        gt = (vb < va) if is_min else (vb > va)
        nan = np.isnan(va) | np.isnan(vb)
        regs[out] = np.where(lt, va, np.where(gt, vb, np.where(nan, np.float32(np.nan), vb))).astype(np.float32)
        tree_ch.append(np.where(lt, 1, np.where(gt, 2, 3)).astype(np.uint8))
    root = regs[(int(tree[-1][0]) >> 8) & 0xFF]
    assert (root.view(np.uint32) == want[0].view(np.uint32)).all()
    # every choice of the full tape, from where the plan says it is recorded
    for c, e in enumerate(src):
        g, j = int(e) >> 24, int(e) & 0xFFFFFF
        got = tree_ch[j] if g == 255 else group_ch[g][j]
        assert (got == want_ch[c]).all(), (c, g, j)


def test_term_plan_is_the_same_function_cpu():
    """Without a GPU: plan_terms on prospero.vm (a chain of 665 terms)."""
    import fidget_amd as F
    _check_plan(F.Shape.from_vm(model_path("prospero.vm")))


def test_term_plan_small_trees_cpu(monkeypatch):
    """... and on small shapes sent down the same path: random CSG (min and max roots, reg,imm tree ops,
    every opcode of the assembly set inside the terms) and a balanced tree with a shared part."""
    import random
    import fidget_amd as F
    from test_render_random import build
    monkeypatch.setenv("FHIP_GROUPS_MIN_OPS", "0")
    monkeypatch.setenv("FHIP_GROUPS_MIN_TERMS", "2")
    split = 0
    for seed in range(12):
        ctx = F.Context()
        s = F.Shape(ctx, build(ctx, seed))
        if s.term_plan()["groups"] >= 2:
            split += 1
            _check_plan(s, seed)
    assert split >= 6
    for op, n in (("min", 13), ("max", 9)):
        ctx = F.Context()
        rng = random.Random(5)
        x, y, z = ctx.x(), ctx.y(), ctx.z()
        def ball():
            c = [rng.uniform(-0.7, 0.7) for _ in range(3)]
            d = ctx.add(ctx.add(ctx.square(ctx.sub(x, c[0])), ctx.square(ctx.sub(y, c[1]))), ctx.square(ctx.sub(z, c[2])))
            return ctx.sub(ctx.sqrt(d), rng.uniform(0.15, 0.4))
        level = [ball() for _ in range(n)]
        shared = level[1]
        while len(level) > 1:
            level = [getattr(ctx, op)(level[i], level[i + 1]) if i + 1 < len(level) else level[i] for i in range(0, len(level), 2)]
        s = F.Shape(ctx, getattr(ctx, op)(level[0], ctx.add(shared, 0.05)))
        p = s.term_plan()
        assert p["groups"] >= 2 and p["tree_regs"] > 1
        _check_plan(s)
