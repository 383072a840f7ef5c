"""fidget_raster::effects (fidget-raster/src/effects.rs) — post-processing of rendered images.

The reference has no test or golden image for this file ("parity unpinned" at the ulp level): the CPU legs below pin the
oracle's restatement on small hand-computed cases (window selection, tie rules, the saturating casts, the fill-pixel
colour tables); the GPU legs compare the HIP kernels with the oracle on rendered images, bit for bit (u8 within 1 for the
one colour map that calls exp / cos)."""
import numpy as np
import pytest

from conftest import model_path


def geom(h, w):
    import oracle as O
    return np.zeros((h, w), O.GEOMETRY_PIXEL)


# ---- oracle pinned by hand ---------------------------------------------------------------------------------------
def test_denoise_keeps_front_facing_and_empty(oracle_mod):
    O = oracle_mod
    img = geom(4, 4)
    img["depth"][1, 1] = 7
    img["normal"][1, 1] = (0.25, -0.5, 0.75)        # z > 0: kept (effects.rs:259-261)
    out = O.denoise_normals(img)
    assert out["depth"].tolist() == img["depth"].tolist()
    assert out["normal"][1, 1].tolist() == [0.25, -0.5, 0.75]
    assert (out["normal"][img["depth"] == 0] == 0).all()          # depth 0 -> [0; 3] (effects.rs:27-29)


def test_denoise_replaces_back_facing_by_best_window_mean(oracle_mod):
    O = oracle_mod
    img = geom(5, 5)
    img["depth"][:] = 3
    img["normal"][:] = (0.0, 0.0, 1.0)
    img["normal"][2, 2] = (0.0, 0.0, -1.0)          # back facing pixel in the middle
    img["normal"][0:2, 0:2] = (1.0, 0.0, 0.5)       # a different neighbourhood up-left
    out = O.denoise_normals(img)
    n = out["normal"][2, 2]
    # the four 3x3 windows touching (2,2); the winner is the one whose mean has the largest summed dot product with its
    # members: window (0,0) = pixels [2..4]x[2..4]: 8 x (0,0,1) front facing -> mean (0,0,1), score = 8*1 + (-1) = 7;
    # the windows reaching the up-left patch score less (mixed normals)
    assert n.tolist() == [0.0, 0.0, 1.0]
    # a pixel with no front-facing neighbour keeps its normal (unwrap_or((0.0, n)), effects.rs:324)
    img2 = geom(3, 3)
    img2["depth"][:] = 1
    img2["normal"][:] = (0.5, 0.5, -0.25)
    assert O.denoise_normals(img2)["normal"][1, 1].tolist() == [0.5, 0.5, -0.25]


def test_blur_picks_lowest_variance_window_and_keeps_nan(oracle_mod):
    O = oracle_mod
    s = np.full((5, 5), 0.5, np.float32)
    s[0:3, 0:3] = [[0.0, 1.0, 0.0], [1.0, 0.0, 1.0], [0.0, 1.0, 0.0]]   # noisy corner
    s[4, 4] = np.nan
    out = O.blur_ssao(s)
    assert np.isnan(out[4, 4])                                        # NaN stays NaN (effects.rs:106-108)
    # at (2,2): window (0,0) = [2..4]^2 holds 0.0 at (2,2), 0.5 elsewhere, NaN skipped: mean = 3.5 / 8
    assert out[2, 2] == np.float32(3.5) / np.float32(8.0)
    assert out[4, 0] == 0.5


def test_shading_values(oracle_mod):
    O = oracle_mod
    img = geom(2, 2)
    img["depth"][0, 0] = 1
    img["normal"][0, 0] = (0.0, 0.0, 2.0)
    out = O.apply_shading(img, 2)
    assert out.shape == (2, 2, 3) and (out[1, 1] == 0).all() and (out[0, 0] == out[0, 0, 0]).all()
    # by hand (f32): n = (0,0,1); p = (-1,-1,0); lights (5,-5,10,.5) (-5,0,10,.15) (0,-5,10,.15); ambient 0.2
    f = np.float32
    acc = f(0.2)
    for lx, ly, lz, w in ((5, -5, 10, 0.5), (-5, 0, 10, 0.15), (0, -5, 10, 0.15)):
        d = np.array([f(lx) - f(-1), f(ly) - f(-1), f(lz) - f(0)], np.float32)
        nrm = np.sqrt(f(0) + ((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]))
        acc = f(acc + max(f(d[2] / nrm), f(0)) * f(w))
    assert out[0, 0, 0] == int(min(max(acc, 0), 1) * f(255))
    # SSAO factor: accum *= s * 0.6 + 0.4
    s = np.full((2, 2), 0.5, np.float32)
    out2 = O.apply_shading(img, 2, s)
    assert out2[0, 0, 0] == int(f(min(max(f(acc * f(f(0.5) * f(0.6) + f(0.4))), 0), 1)) * f(255))


def test_ssao_flat_plane_is_unoccluded_and_empty_is_nan(oracle_mod):
    O = oracle_mod
    img = geom(16, 16)
    img["depth"][:] = 8
    img["normal"][:] = (0, 0, 1)
    img["depth"][0, 0] = 0
    rng = np.random.default_rng(3)
    k = rng.uniform(-1, 1, (3, 16)).astype(np.float32)
    k[2] = np.abs(k[2])
    nz = rng.uniform(-1, 1, (2, 8)).astype(np.float32)
    s = O.compute_ssao(img, 16, k, nz)
    assert np.isnan(s[0, 0])
    # interior pixels of a flat plane facing the camera: every hemisphere sample is in front of the surface
    assert (s[6:10, 6:10] == 1.0).all()
    assert ((s[~np.isnan(s)] >= 0) & (s[~np.isnan(s)] <= 1)).all()


def fill(depth, inside):
    return np.array([0x7FC00000 | (depth << 1) | int(inside) | (0xF6 << 9)], np.uint32).view(np.float32)[0]


def test_colour_maps_of_fill_and_distance_pixels(oracle_mod):
    O = oracle_mod
    img = np.array([[-1.0, 1.0, 0.0, np.nan], [fill(0, True), fill(0, False), fill(2, True), fill(5, False)]], np.float32)
    a = O.to_rgba_bitmap(img)
    assert a[0].tolist() == [[255] * 4, [0, 0, 0, 255], [0, 0, 0, 255], [0, 0, 0, 255]]       # 0.0 and NaN are outside (pixel.rs:184-193)
    assert a[1].tolist() == [[255] * 4, [0, 0, 0, 255], [255] * 4, [0, 0, 0, 255]]
    assert O.to_rgba_bitmap(img, True)[0, 1].tolist() == [0, 0, 0, 0]
    d = O.to_debug_bitmap(img)
    assert d[1].tolist() == [[255, 0, 0, 255], [50, 0, 0, 255], [0, 0, 255, 255], [50, 50, 0, 255]]   # effects.rs:485-494
    assert d[0, 0].tolist() == [255] * 4 and d[0, 1].tolist() == [0, 0, 0, 255]
    c = O.to_rgba_distance(img)
    assert c[0, 3].tolist() == [255, 0, 0, 255]                                                # NaN distance: pure red
    assert c[1, 0].tolist() == [184, 235, 255, 255] and c[1, 1].tolist() == [217, 144, 72, 255]
    assert c[0, 2].tolist() == [255, 255, 255, 255]                                            # f = 0: both smoothsteps are 0 -> white


# ---- device against the oracle -----------------------------------------------------------------------------------
def kernels(seed=7, nk=64, nn=256):
    """effects::ssao_kernel / ssao_noise (effects.rs:395-448) with a seeded generator instead of rand::rng()"""
    rng = np.random.default_rng(seed)
    k = np.zeros((3, nk), np.float32)
    for i in range(nk):
        while True:
            row = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0, 1)], np.float32)
            nrm = np.float32(np.linalg.norm(row))
            if np.finfo(np.float32).eps < nrm < 1.0:
                scale = np.float32((np.float32(i) / np.float32(nk - 1)) ** 2 * 0.9 + 0.1)
                k[:, i] = row * scale / nrm
                break
    nz = np.zeros((2, nn), np.float32)
    for i in range(nn):
        while True:
            row = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1)], np.float32)
            nrm = np.float32(np.linalg.norm(row))
            if np.finfo(np.float32).eps < nrm < 1.0:
                nz[:, i] = row / nrm
                break
    return k, nz


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return ((a == b) | (np.isnan(a) & np.isnan(b))).all() if a.dtype.kind == "f" else (a == b).all()


@pytest.mark.gpu
@pytest.mark.parametrize("model,size", [("bear.vm", 128), ("prospero.vm", 256), ("colonnade.vm", 200)])
def test_3d_effects_match_oracle(model, size, oracle_mod):
    import fidget_amd as F
    O = oracle_mod
    img = O.render3d(O.Shape.from_vm(model_path(model)), size)[0]
    k, nz = kernels()
    d_o, d_f = O.denoise_normals(img), F.denoise_normals(img)
    assert (d_o["depth"] == d_f["depth"]).all() and same(d_o["normal"], d_f["normal"])
    s_o, s_f = O.compute_ssao(d_o, size, k, nz), F.compute_ssao(d_o, size, k, nz)
    assert same(s_o, s_f), f"{(~((s_o == s_f) | (np.isnan(s_o) & np.isnan(s_f)))).sum()} SSAO pixels differ"
    b_o, b_f = O.blur_ssao(s_o), F.blur_ssao(s_o)
    assert same(b_o, b_f)
    assert same(O.apply_shading(d_o, size, b_o), F.apply_shading(d_o, size, b_o))
    assert same(O.apply_shading(d_o, size), F.apply_shading(d_o, size))


@pytest.mark.gpu
def test_2d_colour_maps_match_oracle(oracle_mod):
    import fidget_amd as F
    O = oracle_mod
    for model, n in (("hi.vm", 128), ("prospero.vm", 512)):
        img = O.render2d(O.Shape.from_vm(model_path(model)), n)[0]
        assert same(O.to_rgba_bitmap(img), F.to_rgba_bitmap(img))
        assert same(O.to_rgba_bitmap(img, True), F.to_rgba_bitmap(img, True))
        assert same(O.to_debug_bitmap(img), F.to_debug_bitmap(img))
        a, b = O.to_rgba_distance(img).astype(int), F.to_rgba_distance(img).astype(int)
        assert np.abs(a - b).max() <= 1      # exp / cos: glibc f32 vs one rounding from f64
        # pixel_perfect renders have no fill pixels: every pixel takes the distance path
        img = O.render2d(O.Shape.from_vm(model_path(model)), 64, pixel_perfect=True)[0]
        assert np.abs(O.to_rgba_distance(img).astype(int) - F.to_rgba_distance(img).astype(int)).max() <= 1
