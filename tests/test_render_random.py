"""Random shapes through the renderers: every opcode of the assembly interpreters (f32 leaves,
interval tile stage, prune) in reg,reg / reg,imm / imm,reg form, GPU against the oracle.

The reference has no such test (its renders are pinned by a handful of golden bitmaps); this is
the size-independent property the prompt asks for where fixtures run out: the product and the
CPU restatement must agree bit for bit on shapes neither has seen before."""
import random

import numpy as np
import pytest


def build(ctx, seed):
    """A random CSG-ish expression over x, y, z (deterministic in `seed`, same calls on any backend)."""
    rng = random.Random(seed)
    x, y, z = ctx.x(), ctx.y(), ctx.z()
    axes = [x, y, z]

    def coord():
        a = rng.choice(axes)
        k = rng.choice([0, 1, 2, 3])
        if k == 0:
            return ctx.sub(a, rng.uniform(-0.6, 0.6))                      # reg - imm
        if k == 1:
            return ctx.sub(rng.uniform(-0.6, 0.6), a)                      # imm - reg
        if k == 2:
            return ctx.mul(ctx.add(a, rng.uniform(-0.5, 0.5)), rng.uniform(0.7, 1.6))
        return ctx.div(ctx.add(a, rng.uniform(-0.5, 0.5)), rng.uniform(0.6, 1.4))   # reg / imm

    def sphere():
        r = rng.uniform(0.2, 0.7)
        s = ctx.add(ctx.add(ctx.square(coord()), ctx.square(coord())), ctx.square(coord()))
        return ctx.sub(ctx.sqrt(s), r)

    def box():
        d = [ctx.sub(ctx.abs(coord()), rng.uniform(0.15, 0.6)) for _ in range(3)]
        return ctx.max(ctx.max(d[0], d[1]), d[2])

    def slab():
        return ctx.sub(ctx.abs(coord()), rng.uniform(0.05, 0.4))

    def odd():
        k = rng.choice(range(8))
        c = coord()
        if k == 0:   # steps
            return ctx.sub(ctx.sub(c, ctx.mul(ctx.floor(ctx.mul(c, 4.0)), 0.25)), 0.1)
        if k == 1:
            return ctx.sub(ctx.sub(ctx.mul(ctx.ceil(ctx.mul(c, 3.0)), 1.0 / 3.0), c), 0.15)
        if k == 2:
            return ctx.sub(ctx.abs(ctx.sub(c, ctx.mul(ctx.round(ctx.mul(c, 2.0)), 0.5))), 0.12)
        if k == 3:   # imm / reg and recip, kept away from the pole
            return ctx.sub(ctx.div(0.3, ctx.add(ctx.square(c), 0.5)), rng.uniform(0.3, 0.5))
        if k == 4:
            return ctx.sub(ctx.recip(ctx.add(ctx.abs(c), 0.8)), rng.uniform(0.7, 1.1))
        if k == 5:   # compare / and / or / not: a half space selected by a predicate
            half = ctx.compare(coord(), coord())                           # -1 / 0 / 1
            return ctx.add(ctx.mul(half, rng.uniform(0.2, 0.5)), sphere())
        if k == 6:
            cond = ctx.max(ctx.compare(coord(), 0.1), 0.0)                 # 1 where coord > 0.1 (reg, imm compare)
            return ctx.or_(ctx.and_(cond, sphere()), ctx.and_(ctx.not_(cond), box()))
        cond = ctx.max(ctx.compare(0.0, coord()), 0.0)                     # imm, reg compare
        return ctx.or_(ctx.and_(cond, slab()), ctx.and_(ctx.not_(cond), sphere()))

    prims = [sphere, box, slab, odd]
    node = rng.choice(prims)()
    for _ in range(rng.randint(4, 10)):
        p = rng.choice(prims)()
        k = rng.random()
        if k < 0.45:
            node = ctx.min(node, p)                                        # union
        elif k < 0.7:
            node = ctx.max(node, p)                                        # intersection
        elif k < 0.85:
            node = ctx.max(node, ctx.neg(p))                               # difference
        elif k < 0.93:
            node = ctx.min(node, rng.uniform(0.05, 0.3))                   # reg, imm choices
        else:
            node = ctx.max(node, rng.uniform(-0.3, -0.05))
    return node


SEEDS = list(range(12))


@pytest.mark.parametrize("seed", SEEDS[:4])
def test_random_shapes_oracle_is_deterministic(seed, oracle_mod):
    """CPU leg: the generator drives the oracle's Context the same way twice (guards the test itself)."""
    O = oracle_mod
    a = O.render3d(O.Shape(*_shape(O, seed)), 32)[0]
    b = O.render3d(O.Shape(*_shape(O, seed)), 32)[0]
    assert (a["depth"] == b["depth"]).all() and a["depth"].max() > 0


def _shape(mod, seed):
    ctx = mod.Context()
    return ctx, build(ctx, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("size", [64, 128, 200])
def test_random_shapes_3d(seed, size, oracle_mod):
    import fidget_amd as F
    O = oracle_mod
    a = F.render3d(F.Shape(*_shape(F, seed)), size)[0]
    b = O.render3d(O.Shape(*_shape(O, seed)), size)[0]
    assert (a["depth"] == b["depth"]).all(), f"{(a['depth'] != b['depth']).sum()} depths differ"
    # no transcendental opcode in these shapes: the normals are bit-exact too (NaN == NaN, +0 == -0 by float equality).
    # Round 1 allowed 1e-5 here; tools/bisect_normals.py (profiles/r02a) found no differing pixel at 64 / 128 / 200.
    na, nb = a["normal"], b["normal"]
    same = (na == nb) | (np.isnan(na) & np.isnan(nb))
    assert same.all(), f"{(~same).any(axis=2).sum()} pixels with different normals"


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS[:6])
def test_random_shapes_2d(seed, oracle_mod):
    import fidget_amd as F
    O = oracle_mod
    a = F.render2d(F.Shape(*_shape(F, seed)), 256, z=0.1)[0]
    b = O.render2d(O.Shape(*_shape(O, seed)), 256, z=0.1, tile_sizes=F.HIP_TILES_2D)[0]
    assert (a.view(np.uint32) == b.view(np.uint32)).all(), f"{(a.view(np.uint32) != b.view(np.uint32)).sum()} pixels differ"


def build_full(ctx, seed):
    """... and the opcodes the shapes above leave out: sin cos tan asin acos atan exp ln, atan2 / modulo / mix in reg,reg / reg,imm / imm,reg form,
    rand - the set whose INTERVAL handlers the assembly tile kernels got in rounds 3 (unary) and 5 (atan2, modulo, rand, mix)."""
    rng = random.Random(1000 + seed)
    x, y, z = ctx.x(), ctx.y(), ctx.z()
    axes = [x, y, z]

    def coord():
        return ctx.add(ctx.mul(rng.choice(axes), rng.uniform(0.6, 1.8)), rng.uniform(-0.4, 0.4))

    def term():
        k = rng.choice(range(12))
        c, d = coord(), coord()
        if k == 0:
            return ctx.sub(ctx.mul(ctx.sin(ctx.mul(c, rng.uniform(3.0, 9.0))), ctx.cos(ctx.mul(d, rng.uniform(2.0, 7.0)))), rng.uniform(-0.2, 0.3))
        if k == 1:
            return ctx.sub(ctx.exp(ctx.neg(ctx.add(ctx.square(c), ctx.square(d)))), rng.uniform(0.3, 0.7))
        if k == 2:
            return ctx.add(ctx.ln(ctx.add(ctx.add(ctx.square(c), ctx.square(d)), 0.05)), rng.uniform(0.2, 1.2))
        if k == 3:
            return ctx.sub(ctx.atan2(c, d), ctx.mul(coord(), 2.0))                           # reg, reg
        if k == 4:
            return ctx.sub(ctx.add(ctx.atan2(c, rng.uniform(-0.5, 0.5)), ctx.atan2(rng.uniform(-0.5, 0.5), d)), 0.4)    # reg, imm and imm, reg
        if k == 5:
            return ctx.sub(ctx.modulo(ctx.mul(c, 3.0), rng.uniform(0.4, 0.9)), rng.uniform(0.15, 0.35))      # reg % imm: the same-floor rule
        if k == 6:
            return ctx.sub(ctx.modulo(c, ctx.add(ctx.square(d), 0.3)), 0.2)                 # reg % reg
        if k == 7:
            return ctx.sub(ctx.modulo(rng.uniform(1.0, 3.0), ctx.add(ctx.abs(c), 0.4)), 0.3)   # imm % reg
        if k == 8:
            return ctx.sub(ctx.add(ctx.square(c), ctx.mul(ctx.rand(ctx.floor(ctx.mul(d, 6.0))), 0.3)), 0.35)   # rand of a step function
        if k == 9:
            cell = ctx.floor(ctx.mul(c, 5.0))
            h = ctx.mix(cell, ctx.floor(ctx.mul(d, 5.0)))                                  # a hash per grid cell: NaN over tiles that straddle cells
            return ctx.sub(ctx.add(ctx.abs(coord()), ctx.mul(ctx.min(ctx.abs(h), 1.0), 1.0e-3)), 0.3)
        if k == 10:
            return ctx.sub(ctx.add(ctx.tan(ctx.mul(c, 0.9)), ctx.asin(ctx.mul(d, 0.5))), rng.uniform(-0.2, 0.4))
        return ctx.sub(ctx.add(ctx.acos(ctx.mul(c, 0.45)), ctx.atan(ctx.mul(d, 3.0))), rng.uniform(1.0, 2.0))

    node = term()
    for _ in range(rng.randint(3, 7)):
        p = term()
        node = ctx.min(node, p) if rng.random() < 0.6 else ctx.max(node, ctx.neg(p))
    r = ctx.sub(ctx.sqrt(ctx.add(ctx.add(ctx.square(x), ctx.square(y)), ctx.square(z))), 0.9)
    return ctx.max(node, r)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(10)))
def test_random_shapes_with_every_opcode(seed, oracle_mod):
    """3D (depth, normals) and 2D renders of shapes with the transcendental, atan2, modulo, rand and mix opcodes against the oracle, and the
    tile stage stays on the assembly kernels: fhip_render_counters' count of frames that took the HIP tile kernels implicitly does not move
    (until round 5 a tape with atan2 / modulo / rand / mix fell to k_teval3d)."""
    import torch
    import fidget_amd as F
    O = oracle_mod
    hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
    cf, co = F.Context(), O.Context()
    p, o = F.Shape(cf, build_full(cf, seed), hip=hip), O.Shape(co, build_full(co, seed))
    for size in (96, 160):
        a = F.render3d(p, size)[0]
        b = O.render3d(o, size)[0]
        assert (a["depth"] == b["depth"]).all(), f"{size}: {(a['depth'] != b['depth']).sum()} depths differ"
        na, nb = a["normal"], b["normal"]
        same = (na.view(np.uint32) == nb.view(np.uint32)) | (np.isnan(na) & np.isnan(nb)) | ((na == 0) & (nb == 0))
        assert same.all(), f"{size}: {(~same).any(axis=2).sum()} pixels with different normals"
    a = F.render2d(p, 200, z=0.05)[0]
    b = O.render2d(o, 200, z=0.05, tile_sizes=F.HIP_TILES_2D)[0]
    same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b) & (F.pixel_fill_depth(a) == O.pixel_fill_depth(b)))
    assert same.all(), f"{(~same).sum()} pixels differ"
    assert hip.counters()["hip_tile_stage_frames"] == 0
    del p, hip
