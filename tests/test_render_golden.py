"""Golden-image and geometry known-answer tests for the tile renderers, from
the reference's own tests:
  fidget/tests/pixel_render.rs        (ASCII bitmaps -> tests/golden/*.txt via extract_goldens.py)
  fidget/tests/voxel_render.rs:13-75  (analytic sphere)
  fidget-core/src/render/region.rs:204-228, fidget-raster/src/pixel.rs:538-609,
  fidget/tests/pixel_render.rs:429-474 (screen->world matrices, exact point equality)
  fidget-raster/src/voxel.rs:561-570  (single-tile render)
Instantiated for N=255 (VmFunction) and N=3 (GenericVmFunction<3>, forces
spills) like render_tests!(vm, ..) / render_tests!(vm3, ..), pixel_render.rs:422-423.
"""
import os

import numpy as np
import pytest

from conftest import model_path

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    rows = [r for r in open(os.path.join(GOLD, name)).read().split("\n") if r and not r.startswith("#")]
    return np.array([[c == "#" for c in r] for r in rows])


def bitmap(be, shape, w, h, **kw):
    img = be.render2d(shape, w, h, **kw)[0]
    return be.pixel_inside(img)


def show(a):
    return "\n".join("".join("#" if v else "." for v in r) for r in a)


def check(be, shape, name, wide=False, **kw):
    want = golden(name)
    got = bitmap(be, shape, 64 if wide else 32, 32, **kw)
    assert got.shape == want.shape
    assert (got == want).all(), "image mismatch\n" + show(got) + "\nexpected\n" + show(want)


T_S = np.array([[0.5, 0, 0.5], [0, 0.5, 0.5], [0, 0, 1]], np.float32)  # translate(0.5,0.5) * scale(0.5)


@pytest.mark.parametrize("n_regs", [255, 3])
def test_render_hi(be, n_regs):  # pixel_render.rs:71-106, 108-150
    s = be.Shape.from_vm(model_path("hi.vm"), n_regs=n_regs)
    check(be, s, "pixel_render_check_hi_EXPECTED.txt")
    check(be, s, "pixel_render_check_hi_wide_EXPECTED.txt", wide=True)


@pytest.mark.parametrize("n_regs", [255, 3])
def test_render_hi_transformed_and_bounded(be, n_regs):  # pixel_render.rs:152-236
    s = be.Shape.from_vm(model_path("hi.vm"), n_regs=n_regs)
    check(be, s, "pixel_render_check_hi_transformed_EXPECTED.txt", world_to_model=T_S)
    # View2::from_center_and_scale((0.5,0.5), 0.5).world_to_model() = T * S (fidget-gui/src/lib.rs:92-104)
    check(be, s, "pixel_render_check_hi_bounded_EXPECTED.txt", world_to_model=T_S)


@pytest.mark.parametrize("n_regs", [255, 3])
def test_render_quarter(be, n_regs):  # pixel_render.rs:238-277
    s = be.Shape.from_vm(model_path("quarter.vm"), n_regs=n_regs)
    check(be, s, "pixel_render_check_quarter_EXPECTED.txt")


@pytest.mark.parametrize("n_regs", [255, 3])
def test_render_circle_var(be, n_regs):  # pixel_render.rs:279-370
    ctx = be.Context()
    x, y = ctx.x(), ctx.y()
    r = ctx.sqrt(ctx.add(ctx.square(x), ctx.square(y)))
    v = 0x1234567
    s = be.Shape(ctx, ctx.sub(r, ctx.var(v)), n_regs=n_regs)
    check(be, s, "pixel_render_check_circle_var_EXPECTED_075.txt", vars={v: 0.75})
    check(be, s, "pixel_render_check_circle_var_EXPECTED_05.txt", vars={v: 0.5})
    with pytest.raises(ValueError):  # MissingVar (shape/mod.rs:848-857)
        be.render2d(s, 32, 32)


def test_render_neg_infinity(be):  # pixel_render.rs:372-385
    ctx = be.Context()
    s = be.Shape(ctx, ctx.constant(-np.inf))
    img = be.render2d(s, 256, 256, pixel_perfect=True)[0]
    assert be.pixel_inside(img).all()


def test_render_config_z(be):  # fidget-raster/src/pixel.rs:538-570
    ctx = be.Context()
    s = be.Shape(ctx, ctx.z())
    assert be.pixel_inside(be.render2d(s, 64, 64, z=-1.0)[0]).all()
    assert not be.pixel_inside(be.render2d(s, 64, 64, z=0.0)[0]).any()
    assert not be.pixel_inside(be.render2d(s, 64, 64, z=1.0)[0]).any()


def _pt2(be, mat3, x, y):
    m = be.lift_2d(mat3)
    p = be.transform_point(m, x, y, 0.0)
    return (float(p[0]), float(p[1]))


def test_screen_size(be):  # render/region.rs:204-228
    m = be.screen_to_world([1000, 500])
    assert _pt2(be, m, 500.0, 249.0) == (0.0, 0.0)
    assert _pt2(be, m, 500.0, -1.0) == (0.0, 1.0)
    assert _pt2(be, m, 500.0, 499.0) == (0.0, -1.0)
    assert _pt2(be, m, 0.0, 249.0) == (-2.0, 0.0)
    assert _pt2(be, m, 1000.0, 249.0) == (2.0, 0.0)


def test_render_config_transforms(be):  # fidget-raster/src/pixel.rs:572-609
    m = be.screen_to_world([512, 512])
    assert _pt2(be, m, 0.0, -1.0) == (-1.0, 1.0)
    assert _pt2(be, m, 512.0, -1.0) == (1.0, 1.0)
    assert _pt2(be, m, 512.0, 511.0) == (1.0, -1.0)
    m = be.screen_to_world([575, 575])
    assert _pt2(be, m, 0.0, -1.0) == (-1.0, 1.0)
    assert _pt2(be, m, 575.0, -1.0) == (1.0, 1.0)
    assert _pt2(be, m, 575.0, 574.0) == (1.0, -1.0)


def test_camera_render_config(be):  # fidget/tests/pixel_render.rs:429-474
    w2m = np.array([[0.5, 0, 0.5], [0, 0.5, 0.5], [0, 0, 1]], np.float32)
    m = be.mat_mul(w2m, be.screen_to_world([512, 512]))
    assert _pt2(be, m, 0.0, -1.0) == (0.0, 1.0)
    assert _pt2(be, m, 512.0, -1.0) == (1.0, 1.0)
    assert _pt2(be, m, 512.0, 511.0) == (1.0, 0.0)
    w2m = np.array([[0.25, 0, 0.5], [0, 0.25, 0.5], [0, 0, 1]], np.float32)
    m = be.mat_mul(w2m, be.screen_to_world([512, 512]))
    assert _pt2(be, m, 0.0, -1.0) == (0.25, 0.75)
    assert _pt2(be, m, 512.0, -1.0) == (0.75, 0.75)
    assert _pt2(be, m, 512.0, 511.0) == (0.75, 0.25)


def test_fill_encoding(be):  # fidget-raster/src/pixel.rs:159-241
    # an empty 256^2 render of `x + 100` is one Fill{depth:0, inside:false} per root tile
    ctx = be.Context()
    s = be.Shape(ctx, ctx.add(ctx.x(), 100.0))
    img = be.render2d(s, 256, 256)[0]
    bits = img.view(np.uint32)
    assert (bits == (0x7FC00000 | (0xF6 << 9) | (0 << 1) | 0)).all()
    s = be.Shape(ctx, ctx.sub(ctx.x(), 100.0))
    img = be.render2d(s, 256, 256)[0]
    assert (img.view(np.uint32) == (0x7FC00000 | (0xF6 << 9) | 1)).all()
    assert (be.pixel_fill_depth(img) == 0).all()


def test_tile_queues_3d(be):  # fidget-raster/src/voxel.rs:560-570
    ctx = be.Context()
    s = be.Shape(ctx, ctx.x())
    img = be.render3d(s, 128)[0]
    assert img.size == 128 * 128
    # x < 0 on the left half: saturated columns (depth = D, normal [0,0,1]); right half empty
    assert (img["depth"][:, :64] == 128).all() and (img["depth"][:, 65:] == 0).all()
    assert (img["normal"][:, :64] == np.array([0, 0, 1], np.float32)).all()


@pytest.mark.parametrize("scale", [1.0, 0.5])
@pytest.mark.parametrize("r", [0.5, 0.75])
def test_sphere_var(be, scale, r):  # fidget/tests/voxel_render.rs:13-75
    ctx = be.Context()
    x, y, z = ctx.x(), ctx.y(), ctx.z()
    v = 0x77
    sphere = ctx.sub(ctx.sqrt(ctx.add(ctx.add(ctx.square(x), ctx.square(y)), ctx.square(z))), ctx.var(v))
    s = be.Shape(ctx, sphere)
    size = 32
    w2m = np.diag([scale, scale, scale, 1.0]).astype(np.float32)  # View3::world_to_model, yaw = pitch = 0
    img = be.render3d(s, size, world_to_model=w2m, vars={v: r})[0]
    eps = 2.0 / size / scale * 2.0
    m = be.screen_to_world([size, size, size])
    for i, p in enumerate(img["depth"].reshape(-1)):
        p = int(p)
        if p == size:
            continue
        px, py = float(i % size), float(i // size)
        pos = be.transform_point(m, px, py, float(p)) * np.float32(scale)
        if p == 0:
            assert np.hypot(pos[0], pos[1]) + eps > r
        else:
            assert abs(r - float(np.sqrt((pos.astype(np.float64) ** 2).sum()))) < eps
