"""The reference's small unit tests next to the hot path that the other files do not carry yet, on both back ends
(oracle here, the HIP library on a GPU):
  fidget-core/src/lib.rs:  it_works, test_constant_folding, test_eval
  fidget-core/src/shape/mod.rs (tests): shape_vars, shape_bind, shape_eval_bulk_size
  fidget-core/src/types/interval.rs (tests): test_interval (min_choice)
  fidget-raster/src/pixel.rs, voxel.rs (tests): shape_with_var
  fidget-raster/src/lib.rs (tests): image sizes (what a render of W x H returns)
"""
import numpy as np
import pytest


def test_it_works(be):  # fidget-core/src/lib.rs it_works
    ctx = be.Context()
    assert ctx.x() == ctx.x()
    a, b = ctx.constant(1.0), ctx.constant(1.0)
    assert a == b
    c = ctx.add(a, b)
    assert c == ctx.constant(2.0)                    # get_const(c) == 2
    assert ctx.neg(c) == ctx.constant(-2.0)
    assert ctx.x() != a                              # NotAConst


def test_constant_folding(be):  # fidget-core/src/lib.rs test_constant_folding
    ctx = be.Context()
    a = ctx.constant(1.0)
    assert len(ctx) == 1
    b = ctx.constant(-1.0)
    assert len(ctx) == 2
    ctx.add(a, b)
    assert len(ctx) == 3
    ctx.add(a, b)
    assert len(ctx) == 3
    ctx.mul(a, b)                                    # -1: there already
    assert len(ctx) == 3


def test_eval(be):  # fidget-core/src/lib.rs test_eval (Context::eval through the shape's point evaluator)
    ctx = be.Context()
    s = be.Shape(ctx, ctx.add(ctx.x(), ctx.y()))
    assert s.eval_point(1.0, 2.0, 0.0)[0] == 3.0
    assert s.eval_point(2.0, 3.0, 0.0)[0] == 5.0


def test_shape_vars(be):  # shape/mod.rs shape_vars
    ctx = be.Context()
    v = 0x51234
    s = be.Shape(ctx, ctx.add(ctx.add(ctx.x(), ctx.y()), ctx.var(v)))
    assert s.var_count() == 3
    ix, iy, iz, iv = s.axis_index(0), s.axis_index(1), s.axis_index(2), s.var_index(v)
    assert ix >= 0 and iy >= 0 and iz == -1 and iv >= 0
    assert sorted([ix, iy, iv]) == [0, 1, 2]         # every slot of the map is taken by exactly one of them
    assert s.var_index(v + 1) == -1


def test_shape_bind(be):  # shape/mod.rs shape_bind: MissingVar until the shape's own variable is in the map
    ctx = be.Context()
    v = 0x51234
    s = be.Shape(ctx, ctx.add(ctx.add(ctx.x(), ctx.y()), ctx.var(v)))
    with pytest.raises(ValueError):
        be.render2d(s, 16, 16)                       # TryFrom<Shape> for BoundShape: no vars at all
    with pytest.raises(ValueError):
        be.render2d(s, 16, 16, vars={})
    with pytest.raises(ValueError):
        be.render2d(s, 16, 16, vars={v + 1: 1.0})    # an unrelated var does not help
    img = be.render2d(s, 16, 16, vars={v + 1: 1.0, v: 2.0})[0]     # extra vars are allowed (var/mod.rs:155-157)
    assert img.shape == (16, 16)
    with pytest.raises(ValueError):
        be.render3d(s, 16, vars={v + 1: 1.0})
    be.render3d(s, 16, vars={v: 2.0})


def test_shape_eval_bulk_size(be):  # shape/mod.rs shape_eval_bulk_size: a constant shape still answers per point
    ctx = be.Context()
    s = be.Shape(ctx, ctx.constant(1.0))
    out = s.eval_float_slice([1.0, 2.0, 3.0], [4.0, 5.0, 6.0], [7.0, 8.0, 9.0])
    assert out.tolist() == [1.0, 1.0, 1.0]


def test_interval_min_choice(be):  # types/interval.rs test_interval: min([0, 1], [0.5, 1.5]) = [0, 1], Choice::Both
    ctx = be.Context()
    s = be.Shape(ctx, ctx.min(ctx.x(), ctx.y()))
    (lo, hi), trace = s.eval_interval((0.0, 1.0), (0.5, 1.5), (0.0, 0.0))
    assert (lo, hi) == (0.0, 1.0)
    assert trace is None                             # nothing decided: the tracing evaluator hands out no trace (vm/mod.rs:529-536)
    (lo, hi), trace = s.eval_interval((0.0, 1.0), (2.0, 3.0), (0.0, 0.0))
    assert (lo, hi) == (0.0, 1.0) and trace is not None and trace.tolist() == [1]      # Choice::Left


@pytest.mark.parametrize("dim", [2, 3])
def test_shape_with_var(be, dim):  # fidget-raster/src/pixel.rs, voxel.rs shape_with_var: x - v with v = 1 at size 64
    ctx = be.Context()
    v = 0x77
    s = be.Shape(ctx, ctx.sub(ctx.x(), ctx.var(v)))
    if dim == 2:
        img = be.render2d(s, 64, 64, vars={v: 1.0})[0]
        assert img.shape == (64, 64)
        # x - 1 < 0 for every pixel centre of [-1, 1]^2
        assert be.pixel_inside(img).all()
        assert not be.pixel_inside(be.render2d(s, 64, 64, vars={v: -1.0})[0]).any()
    else:
        img = be.render3d(s, 64, vars={v: 1.0})[0]
        w = np.asarray(img).view(np.uint32).reshape(64, 64, 4)
        assert (w[..., 3] == 64).all()               # filled to the top: (depth, [0, 0, 1]) (voxel.rs:536-542)
        assert (np.asarray(img).view(np.float32).reshape(64, 64, 4)[..., :3] == np.array([0, 0, 1], np.float32)).all()
        w = np.asarray(be.render3d(s, 64, vars={v: -1.0})[0]).view(np.uint32).reshape(64, 64, 4)
        assert (w == 0).all()                        # empty


def test_image_sizes(be):  # fidget-raster/src/lib.rs image_construction: Image::new(ImageSize) is height rows of width pixels
    ctx = be.Context()
    s = be.Shape(ctx, ctx.sub(ctx.sqrt(ctx.add(ctx.square(ctx.x()), ctx.square(ctx.y()))), 0.5))
    img = be.render2d(s, 40, 24)[0]
    assert np.asarray(img).shape[:2] == (24, 40)
