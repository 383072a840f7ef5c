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
    assert op == "min" and 2 <= len(gs) <= 16
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
    assert 2 <= p["groups"] <= 16 and p["terms"] >= 600
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
    assert np.abs(a["normal"] - b["normal"]).max() <= 1e-5, (what, n)

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
