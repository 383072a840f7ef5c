"""Tapes that need more registers than any register file holds.  The reference's VM has 255 registers and spills
the rest to memory slots (RegOp::Load / Store, fidget-core/src/compiler/alloc.rs:116-125, KAT vm/data.rs:415-436); the
device tape keeps 12-bit register numbers instead of Load / Store ops, and a register file that does not fit LDS lives in
HBM (render_state.h gscratch) - a slow path with the same results: here a shape with 303 simultaneously live values,
through the trait-level evaluators and both renderers, against the oracle (whose RegTape does spill: 698 Load / Store ops)."""
import numpy as np
import pytest

import fidget_amd as F
import oracle as O


def many_live_values(be, n=300):
    """min over n small spheres + a tiny multiple of the sum of the same n terms: every term is needed twice, first by one chain
    then by the other, so all n are live at once"""
    c = be.Context()
    x, y, z = c.x(), c.y(), c.z()
    ts = []
    for i in range(n):
        cx, cy, cz = 0.6 * np.cos(0.5 * i) * (i / n), 0.6 * np.sin(0.5 * i) * (i / n), -0.5 + i / n
        r2 = c.add(c.add(c.square(c.sub(x, float(cx))), c.square(c.sub(y, float(cy)))), c.square(c.sub(z, float(cz))))
        ts.append(c.sub(r2, 0.01 + 0.0001 * i))
    a = ts[0]
    for t in ts[1:]:
        a = c.min(a, t)
    b = ts[0]
    for t in ts[1:]:
        b = c.add(b, c.mul(t, 1e-6))
    return be.Shape(c, c.add(a, c.mul(b, 1e-9)))


def test_the_shape_needs_more_than_255_registers():
    s, o = many_live_values(F), many_live_values(O)
    assert s.slot_count() == o.slot_count() == 303          # simultaneously live values (the reference: 255 registers + spill slots)
    # Function::size = VmData<255>::len(): the RegTape with its Load / Store ops on both sides; the device tape itself carries none
    assert s.size() == o.size() > s.device_len() == 3600


@pytest.mark.gpu
def test_trait_level_evaluators_with_303_registers():
    s, o = many_live_values(F), many_live_values(O)
    rng = np.random.default_rng(3)
    n = 500
    pts = [rng.uniform(-1, 1, n).astype(np.float32) for _ in range(3)]
    a, b = s.eval_float_slice(*pts), o.eval_float_slice(*pts)
    assert (np.asarray(a, np.float32).view(np.uint32) == np.asarray(b, np.float32).view(np.uint32)).all()
    ga, gb = s.eval_grad_slice(*pts), o.eval_grad_slice(*pts)
    assert (np.asarray(ga, np.float32).view(np.uint32) == np.asarray(gb, np.float32).view(np.uint32)).all()
    for k in range(40):
        lo = rng.uniform(-1, 0.9, 3)
        box = [(float(l), float(l + rng.uniform(0, 0.2))) for l in lo]
        (ra, ta), (rb, tb) = s.eval_interval(*box), o.eval_interval(*box)
        assert np.array(ra, np.float32).view(np.uint32).tolist() == np.array(rb, np.float32).view(np.uint32).tolist()
        assert (ta is None) == (tb is None) and (ta is None or (np.asarray(ta) == np.asarray(tb)).all())
        (pa, qa), (pb, qb) = s.eval_point(*[b_[0] for b_ in box]), o.eval_point(*[b_[0] for b_ in box])
        assert np.float32(pa).view(np.uint32) == np.float32(pb).view(np.uint32)
        assert (qa is None) == (qb is None) and (qa is None or (np.asarray(qa) == np.asarray(qb)).all())


@pytest.mark.gpu
@pytest.mark.parametrize("size", [64, 160])
def test_renders_with_303_registers(size):
    s, o = many_live_values(F), many_live_values(O)
    a, b = F.render2d(s, size)[0], O.render2d(o, size, tile_sizes=F.HIP_TILES_2D)[0]
    assert (np.asarray(a).view(np.uint32) == np.asarray(b).view(np.uint32)).all(), "2D image differs"
    a, b = F.render3d(s, size)[0], O.render3d(o, size)[0]
    assert (a["depth"] == b["depth"]).all(), f"{(a['depth'] != b['depth']).sum()} depths differ"
    assert (a["normal"].view(np.uint32) == b["normal"].view(np.uint32)).all(), "normals differ"
    assert a["depth"].max() > 0


def many_choices(be, n=6500):
    """min over n small spheres, one after the other: few live values, n - 1 choices - more than the LDS choice area of the assembly
    tile kernels holds beside a register file (~5 600), so the tile stage must leave them (capi.hip: asm_tiles only without big_hbm)"""
    c = be.Context()
    x, y, z = c.x(), c.y(), c.z()
    a = None
    for i in range(n):
        t = i / n
        cx, cy, cz = 0.7 * np.cos(40.0 * t) * t, 0.7 * np.sin(40.0 * t) * t, -0.8 + 1.6 * t
        r2 = c.add(c.add(c.square(c.sub(x, float(cx))), c.square(c.sub(y, float(cy)))), c.square(c.sub(z, float(cz))))
        d = c.sub(r2, 0.004)
        a = d if a is None else c.min(a, d)
    return be.Shape(c, a)


def test_the_many_choices_shape_has_few_registers():
    s = many_choices(F)
    assert s.choice_count() == 6499 and s.slot_count() <= 16


@pytest.mark.gpu
@pytest.mark.parametrize("size", [64, 128])
def test_renders_with_6499_choices_and_few_registers(size):
    s, o = many_choices(F), many_choices(O)
    a, b = F.render2d(s, size)[0], O.render2d(o, size, tile_sizes=F.HIP_TILES_2D)[0]
    assert (np.asarray(a).view(np.uint32) == np.asarray(b).view(np.uint32)).all(), "2D image differs"
    a, b = F.render3d(s, size)[0], O.render3d(o, size)[0]
    assert (a["depth"] == b["depth"]).all(), f"{(a['depth'] != b['depth']).sum()} depths differ"
    assert (a["normal"].view(np.uint32) == b["normal"].view(np.uint32)).all(), "normals differ"
    assert a["depth"].max() > 0
