"""Tape parallelism (host_graph.hpp split_root): the independent sub-tapes of a root min / max,
combined in order, are the same function as the whole tape - bit for bit, for points and intervals."""
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
