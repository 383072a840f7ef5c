"""fidget-mesh (Manifold Dual Contouring): the oracle's restatement (oracle/src/mesh.hpp) pinned by the reference's own
tests (fidget-mesh/src/octree.rs:1092-1560, qef.rs:128-169, fidget/tests/octree.rs:9-30), then the device leaf sampler
against the oracle.  Vertex positions come out of a QEF solve whose SVD is not restated bit for bit: the tolerances are
the reference's."""
import itertools

import numpy as np
import pytest

from conftest import model_path


def sphere(c, center, r):
    x, y, z = c.x(), c.y(), c.z()
    d = c.add(c.add(c.square(c.sub(x, center[0])), c.square(c.sub(y, center[1]))), c.square(c.sub(z, center[2])))
    return c.sub(c.sqrt(d), r)


def cube(c, bx, by, bz):
    x, y, z = c.x(), c.y(), c.z()
    xb = c.max(c.sub(bx[0], x), c.sub(x, bx[1]))
    yb = c.max(c.sub(by[0], y), c.sub(y, by[1]))
    zb = c.max(c.sub(bz[0], z), c.sub(z, bz[1]))
    return c.max(c.max(xb, yb), zb)


def check_for_vertex_dupes(verts):            # octree.rs:1561-1570
    v = verts.view(np.uint32).reshape(-1, 3)
    assert len(np.unique(v, axis=0)) == len(v), "duplicate vertices"


def check_for_edge_matching(tris):            # octree.rs:1572-1594
    edges = {}
    for t in tris.tolist():
        assert t[0] != t[1] and t[1] != t[2] and t[0] != t[2], "triangle with duplicate edges"
        for e in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
            edges[e] = edges.get(e, 0) + 1
    for (a, b), n in edges.items():
        assert n == 1, f"duplicate edge ({a}, {b})"
        assert (b, a) in edges, "unpaired edges"


def test_tables_match_build_rs_properties(oracle_mod):
    """build.rs: every inside -> outside corner pair along an axis appears once; vertices partition them by region"""
    O = oracle_mod
    for mask in range(256):
        v2e, e2v = O.mdc_table(mask)
        want = {(s, s ^ ax) for s in range(8) for ax in (1, 2, 4) if (mask >> s) & 1 and not (mask >> (s ^ ax)) & 1}
        got = [e for vs in v2e for e in vs]
        assert set(got) == want and len(got) == len(want)
        assert len(v2e) <= 4
        n = 0
        for vi, vs in enumerate(v2e):
            for s, e in vs:
                t = s ^ e
                u = 1 if t == 4 else t << 1
                v = 1 if u == 4 else u << 1
                edge = {1: 0, 2: 1, 4: 2}[t] * 4 + (1 if s & u else 0) + (2 if s & v else 0)
                assert e2v[edge] == (vi, len(v2e) + n)
                n += 1
    assert O.mdc_table(0) == ([], [None] * 12) and O.mdc_table(255)[0] == []
    # the cube corner: one vertex with three edges (Nielson's case 1)
    assert O.mdc_table(1)[0] == [[(0, 1), (0, 2), (0, 4)]]
    # two opposite corners: two separate vertices
    assert len(O.mdc_table(0b10000001)[0]) == 2


def test_mesh_basic(oracle_mod):              # octree.rs:1141-1176
    O = oracle_mod
    c = O.Context()
    s = O.Shape(c, sphere(c, (0, 0, 0), 0.2))
    o = O.Octree(s, 0)
    assert len(o.cells) == 0 and o.root[0] == "Empty" and len(o.verts) == 0
    t, v = o.walk_dual()
    assert len(t) == 0 and len(v) == 0
    o = O.Octree(s, 1)
    assert len(o.cells) == 1 and o.root == ("Branch", 0, 0)
    assert len(o.verts) == 6 * 4 + 8
    for kind, mask, index in o.cells[0].tolist():
        assert O.CELL_KINDS[kind] == "Leaf" and bin(mask).count("1") == 1 and index % 4 == 0
    t, v = o.walk_dual()
    assert len(v) > 1 and len(t) > 0


def test_sphere_verts(oracle_mod):            # octree.rs:1178-1215
    O = oracle_mod
    c = O.Context()
    o = O.Octree(O.Shape(c, sphere(c, (0, 0, 0), 0.2)), 1)
    _, verts = o.walk_dual()
    edge_count = 0
    for v in verts:
        nz = int((v != 0).sum())
        assert nz in (1, 3)
        if nz == 1:
            assert abs(np.linalg.norm(v) - 0.2) < 2.0 / 65535
            edge_count += 1
        else:
            assert np.linalg.norm(np.abs(v) - 0.2) < 2.0 / 65535, v
    assert edge_count == 6


def test_sphere_manifold(oracle_mod):         # octree.rs:1217-1232
    O = oracle_mod
    c = O.Context()
    o = O.Octree(O.Shape(c, sphere(c, (0, 0, 0), 0.85)), 5)
    t, v = o.walk_dual()
    check_for_vertex_dupes(v)
    check_for_edge_matching(t)


def test_cube_verts(oracle_mod):              # octree.rs:1234-1276
    O = oracle_mod
    c = O.Context()
    o = O.Octree(O.Shape(c, cube(c, (-0.1, 0.6), (-0.2, 0.75), (-0.3, 0.4))), 1)
    _, verts = o.walk_dual()
    eps = 2.0 / 65535
    assert len(verts)
    lim = [(-0.1, 0.6), (-0.2, 0.75), (-0.3, 0.4)]
    for v in verts:
        on = [bool(v[k] != 0) for k in range(3)]
        near = [abs(v[k] - lim[k][0]) < eps or abs(v[k] - lim[k][1]) < eps for k in range(3)]
        assert sum(on) in (1, 3)
        if sum(on) == 1:
            assert any(on[k] and near[k] for k in range(3)), v
        else:
            assert all(near), v


def test_cube_edge(oracle_mod):               # octree.rs:1092-1107
    O = oracle_mod
    c = O.Context()
    o = O.Octree(O.Shape(c, cube(c, (-2, 2), (-2, 0.3), (-2, 0.6))), 0)
    assert len(o.verts) == 5
    assert np.linalg.norm(o.verts[0] - np.array([0.0, 0.3, 0.6])) < 1e-3


def test_plane_center(oracle_mod):            # octree.rs:1278-1312
    O = oracle_mod
    for dx, dy, off in itertools.product([0.0, 0.25, -0.25, 2.0, -2.0], [0.0, 0.25, -0.25, 2.0, -2.0], [0.0, -0.2, 0.2]):
        c = O.Context()
        f = c.add(c.add(c.add(c.mul(c.x(), dx), c.mul(c.y(), dy)), c.z()), off)
        s = O.Shape(c, f)
        o = O.Octree(s, 0)
        assert len(o.cells) == 0
        pos = o.verts[0]
        mass = o.verts[1:].mean(axis=0)
        assert np.linalg.norm(pos - mass) < 1e-3, (dx, dy, off, pos, mass)
        for v in o.verts:
            assert abs(s.eval_point(float(v[0]), float(v[1]), float(v[2]))[0]) < 1e-3


def test_cone_vert(oracle_mod):               # octree.rs:1314-1343 (cone: 1109-1139)
    O = oracle_mod
    for tip in (np.array([0.2, 0.3, 0.4]), np.array([1.2, 1.3, 1.4])):
        corner = np.array([-1.0, -1.0, -1.0])
        d = (tip - corner).astype(np.float32)
        length = float(np.linalg.norm(d))
        d = d / np.float32(length)
        c = O.Context()
        p = [c.x(), c.y(), c.z()]
        offs = [c.sub(p[k], float(corner[k])) for k in range(3)]
        a = c.add(c.add(c.mul(offs[0], float(d[0])), c.mul(offs[1], float(d[1]))), c.mul(offs[2], float(d[2])))
        apos = [c.add(float(corner[k]), c.mul(float(d[k]), a)) for k in range(3)]
        o2 = [c.sub(p[k], apos[k]) for k in range(3)]
        b = c.sqrt(c.add(c.add(c.square(o2[0]), c.square(o2[1])), c.square(o2[2])))
        f = c.sub(b, c.mul(0.1, c.sub(1.0, c.div(a, length))))
        s = O.Shape(c, f)
        assert abs(s.eval_point(*[float(t) for t in tip])[0]) < 1e-6
        assert s.eval_point(-1.0, -1.0, -1.0)[0] < 0
        o = O.Octree(s, 0)
        assert len(o.cells) == 0 and len(o.verts) == 4
        assert np.linalg.norm(o.verts[0] - tip) < 1e-3, (o.verts[0], tip)


def test_qef_rank2_and_near_planar(oracle_mod):     # qef.rs:133-168
    O = oracle_mod
    _, err = O.qef_solve([[-0.5, -0.75, -0.75], [-0.75, -1.0, -0.6], [-0.5, -1.0, -0.6]],
                         [[0.24, 0.12, 0.0, 0.0], [0.0, 0.0, 0.31, 0.0], [0.0, 0.0, 0.31, 0.0]])
    assert err == np.float32(1e-6)
    pos, err = O.qef_solve([[-0.5, -0.25, 0.4999981], [-0.5, -0.25, 0.5], [-0.5, -0.25, 0.5]],
                           [[-0.66666776, -0.33333388, 0.66666526, -1.2516975e-6], [-0.6666667, -0.33333334, 0.6666667, 0.0],
                            [-0.6666667, -0.33333334, 0.6666667, 0.0]])
    assert err == np.float32(1e-6)
    assert np.linalg.norm(pos - np.array([-0.5, -0.25, 0.5])) < 1e-3


@pytest.mark.parametrize("mask", list(range(256)))
def test_mesh_manifold(mask, oracle_mod):     # octree.rs:1345-1391: every one of the 256 corner masks
    O = oracle_mod
    c = O.Context()
    parts = [sphere(c, (0.5 if j & 1 else 0.0, 0.5 if j & 2 else 0.0, 0.5 if j & 4 else 0.0), 0.1) for j in range(8) if mask & (1 << j)]
    if not parts:
        return
    node = parts.pop()
    for p in parts:
        node = c.min(node, p)
    o = O.Octree(O.Shape(c, node), 2)
    t, v = o.walk_dual()
    if mask not in (0, 255):
        assert len(v) and len(t)
    check_for_vertex_dupes(v)
    check_for_edge_matching(t)


def test_colonnade_manifold_and_bounds(oracle_mod):   # octree.rs:1476-1500 (manifold, depth 5), 1502-1530 (bounds, depth 8)
    O = oracle_mod
    s = O.Shape.from_vm(model_path("colonnade.vm"))
    t, v = O.Octree(s, 5).walk_dual()
    check_for_edge_matching(t)
    _, v = O.Octree(s, 8).walk_dual()
    assert (v[:, 0] < 1).all() and (v[:, 0] > -1).all() and (v[:, 1] < 1).all() and (v[:, 1] > -1).all() and (v[:, 2] < 1).all() and (v[:, 2] > -0.5).all()


def test_bear_bounds(oracle_mod):             # octree.rs:1532-1559
    O = oracle_mod
    _, v = O.Octree(O.Shape.from_vm(model_path("bear.vm")), 5).walk_dual()
    assert (v[:, 0] < 1).all() and (v[:, 0] > -0.75).all() and (v[:, 1] < 1).all() and (v[:, 1] > -0.75).all()
    assert (v[:, 2] < 0.75).all() and (v[:, 2] > -0.75).all()


def test_octree_camera(oracle_mod):           # fidget/tests/octree.rs:9-30: View3::from_center_and_scale(center, 0.5)
    O = oracle_mod
    c = O.Context()
    s = O.Shape(c, sphere(c, (1.0, 1.0, 1.0), 0.25))
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] *= 0.5
    m[:3, 3] = 1.0
    _, v = O.Octree(s, 4, world_to_model=m).walk_dual()
    n = np.linalg.norm(v - 1.0, axis=1)
    assert len(v) and (n > 0.2).all() and (n < 0.3).all()


def test_gyroid_sphere_model_and_leaf_samples(oracle_mod):
    """models/gyroid-sphere.vm restates models/gyroid-sphere.rhai (BASELINE config 5); its leaf samples obey the invariants of
    octree.rs:590-862: intersections lie on cell edges between an inside and an outside corner, |value| small there"""
    O = oracle_mod
    s = O.Shape.from_vm(model_path("gyroid-sphere.vm"))
    x, y, z = 0.31, -0.22, 0.4
    g = np.sin(30 * x) * np.cos(30 * y) + np.sin(30 * y) * np.cos(30 * z) + np.sin(30 * z) * np.cos(30 * x)
    want = max(np.sqrt((30 * x) ** 2 + (30 * y) ** 2 + (30 * z) ** 2) - 25, abs(g) - 0.2)
    assert abs(s.eval_point(x, y, z)[0] - want) < 1e-4
    o = O.Octree(s, 5)
    sm = o.samples
    assert len(sm["info"]) > 100
    for i in range(0, len(sm["info"]), 37):
        mask, ne, nv = sm["info"][i]
        v2e, _ = O.mdc_table(int(mask))
        assert ne == sum(len(v) for v in v2e) and nv == len(v2e)
        for e in range(ne):
            on_edge = sorted(int(p in (0, 65535)) for p in sm["inter"][i, e])
            assert on_edge == [0, 1, 1] or on_edge == [1, 1, 1]
            assert abs(s.eval_point(*[float(t) for t in sm["pos"][i, e]])[0]) < 0.05


# ---- device leaf sampler against the oracle ---------------------------------------------------------------------------
def _key(bounds):
    return tuple(np.asarray(bounds, np.float32).view(np.uint32).tolist())


def _compare(F, O, fshape, oshape, depth, w2m=None):
    leaves, counts = F.mesh_sample(fshape, depth, world_to_model=w2m)
    o = O.Octree(oshape, depth, world_to_model=w2m)
    # every cell the reference's recursion interval-evaluates, and no other
    assert counts["cells"] == o.interval_evals, (counts, o.interval_evals)
    sm = o.samples
    want = {_key(sm["bounds"][i]): i for i in range(len(sm["info"]))}
    sampled = leaves[(leaves["mask"] != 0) & (leaves["mask"] != 255)]
    assert len(sampled) == len(want), (len(sampled), len(want))
    for lf in sampled:
        i = want[_key(lf["bounds"])]
        mask, ne, nv = (int(v) for v in sm["info"][i])
        assert (int(lf["mask"]), int(lf["n_edges"]), int(lf["n_verts"])) == (mask, ne, nv)
        assert (lf["inter"][:ne] == sm["inter"][i, :ne]).all(), "edge-search intersections differ"
        assert (lf["pos"][:ne].view(np.uint32) == sm["pos"][i, :ne].view(np.uint32)).all()
        g, w = lf["grad"][:ne], sm["grad"][i, :ne]
        assert ((g == w) | (np.isnan(g) & np.isnan(w))).all(), "gradients differ"
        v, wv = lf["vert"][:nv], sm["vert"][i, :nv]
        assert ((v == wv) | (np.isnan(v) & np.isnan(wv))).all(), f"QEF vertices differ: {v} vs {wv}"
    return counts, len(sampled)


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [0, 1, 3, 5])
def test_device_leaf_samples_sphere_and_cube(depth, oracle_mod):
    import fidget_amd as F
    O = oracle_mod
    for build in (lambda c: sphere(c, (0.1, -0.05, 0.2), 0.6), lambda c: cube(c, (-0.1, 0.6), (-0.2, 0.75), (-0.3, 0.4))):
        cf, co = F.Context(), O.Context()
        _compare(F, O, F.Shape(cf, build(cf)), O.Shape(co, build(co)), depth)


@pytest.mark.gpu
@pytest.mark.parametrize("model,depth", [("colonnade.vm", 6), ("prospero.vm", 5), ("tanglecube.vm", 6)])
def test_device_leaf_samples_models(model, depth, oracle_mod):
    import fidget_amd as F
    O = oracle_mod
    counts, n = _compare(F, O, F.Shape.from_vm(model_path(model)), O.Shape.from_vm(model_path(model)), depth)
    assert n > 50


@pytest.mark.gpu
def test_device_leaf_samples_camera(oracle_mod):
    import fidget_amd as F
    O = oracle_mod
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] *= 0.5
    m[:3, 3] = 1.0
    cf, co = F.Context(), O.Context()
    _compare(F, O, F.Shape(cf, sphere(cf, (1.0, 1.0, 1.0), 0.25)), O.Shape(co, sphere(co, (1.0, 1.0, 1.0), 0.25)), 4, w2m=m)


@pytest.mark.gpu
@pytest.mark.parametrize("model,depth", [("gyroid-sphere.vm", 6), ("bear.vm", 5)])
def test_device_leaf_samples_transcendental(model, depth, oracle_mod):
    """BASELINE configuration 5's model (sin / cos) and bear.vm (exp / ln too): the leaf records - intersections, positions, gradients,
    QEF vertices - bit for bit like every other model's: the device runs the host libm's routines (trans_libm.hpp; until round 4 a
    bound of 8 ulp on the gradients and a count of differing cells stood here)"""
    import fidget_amd as F
    O = oracle_mod
    counts, n = _compare(F, O, F.Shape.from_vm(model_path(model)), O.Shape.from_vm(model_path(model)), depth)
    assert n > 100


# ---- the whole mesher (device sampling + host assembly and dual walk) against the oracle -------------------------------------
def _same_mesh(F, O, fshape, oshape, depth, w2m=None):
    tris, verts, counts = F.mesh(fshape, depth, world_to_model=w2m)
    t, v = O.Octree(oshape, depth, world_to_model=w2m).walk_dual()
    t, v = np.asarray(t, np.uint64).reshape(-1, 3), np.asarray(v, np.float32).reshape(-1, 3)
    assert tris.shape == t.shape and verts.shape == v.shape, (tris.shape, t.shape, verts.shape, v.shape)
    assert (tris == t).all(), "triangles differ"
    assert ((verts == v) | (np.isnan(verts) & np.isnan(v))).all(), "vertices differ"
    return tris, verts


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [0, 1, 2, 4, 5])
def test_device_mesh_sphere_and_cube(depth, oracle_mod):
    """octree.rs:1141-1276's shapes: triangles and vertices identical to Octree::build(..).walk_dual(), collapsed cells included"""
    import fidget_amd as F
    O = oracle_mod
    for build in (lambda c: sphere(c, (0.1, -0.05, 0.2), 0.6), lambda c: cube(c, (-0.1, 0.6), (-0.2, 0.75), (-0.3, 0.4)),
                  lambda c: sphere(c, (0.0, 0.0, 0.0), 0.2)):
        cf, co = F.Context(), O.Context()
        tris, verts = _same_mesh(F, O, F.Shape(cf, build(cf)), O.Shape(co, build(co)), depth)
        if depth >= 2:
            assert len(tris) > 0
            check_for_vertex_dupes(verts)
            check_for_edge_matching(tris)


@pytest.mark.gpu
@pytest.mark.parametrize("model,depth", [("colonnade.vm", 6), ("prospero.vm", 6), ("tanglecube.vm", 6)])
def test_device_mesh_models(model, depth, oracle_mod):
    import fidget_amd as F
    O = oracle_mod
    tris, verts = _same_mesh(F, O, F.Shape.from_vm(model_path(model)), O.Shape.from_vm(model_path(model)), depth)
    assert len(tris) > 100
    if model == "colonnade.vm":
        check_for_edge_matching(tris)


@pytest.mark.gpu
def test_device_mesh_colonnade_at_the_reference_s_depth(oracle_mod):     # octree.rs:1502-1530 builds colonnade at depth 8
    """the device mesh at the depth the reference's own bound test uses: triangles and vertices identical to the oracle's, and the
    reference's bounds hold for it"""
    import fidget_amd as F
    O = oracle_mod
    tris, verts = _same_mesh(F, O, F.Shape.from_vm(model_path("colonnade.vm")), O.Shape.from_vm(model_path("colonnade.vm")), 8)
    assert len(tris) > 50000
    v = verts
    assert (v[:, 0] < 1).all() and (v[:, 0] > -1).all() and (v[:, 1] < 1).all() and (v[:, 1] > -1).all() and (v[:, 2] < 1).all() and (v[:, 2] > -0.5).all()
    check_for_edge_matching(tris)


@pytest.mark.gpu
@pytest.mark.parametrize("model,depth", [("gyroid-sphere.vm", 7), ("colonnade.vm", 6), ("bear.vm", 5)])
def test_leaf_passes_equal_the_per_cell_kernel(model, depth, monkeypatch):
    """The leaf sampling as passes over a chunk's cells (k_mesh_corners / k_mesh_edges / k_mesh_grads: every lane a point of its own) writes
    the records of k_mesh_leaf (one wavefront per cell), bit for bit - per lane the arithmetic is the same."""
    import fidget_amd as F
    shape = F.Shape.from_vm(model_path(model))
    monkeypatch.delenv("FHIP_MESH_LEAF_PASSES", raising=False)
    a, ca = F.mesh_sample(shape, depth)
    monkeypatch.setenv("FHIP_MESH_LEAF_PASSES", "0")          # (read by every build: the per-cell kernel)
    b, cb = F.mesh_sample(shape, depth)
    monkeypatch.delenv("FHIP_MESH_LEAF_PASSES")
    assert ca == cb and len(a) == len(b) > 1000
    a, b = a[np.argsort(a["path"], kind="stable")], b[np.argsort(b["path"], kind="stable")]       # (a level's cells take their slots from an atomic counter: any order)
    assert len(np.unique(a["path"])) == len(a)
    for f in ("bounds", "path", "mask", "n_edges", "n_verts"):
        assert (a[f] == b[f]).all(), f
    ne, nv = a["n_edges"], a["n_verts"]
    e_ok = np.arange(12)[None, :] < ne[:, None]
    v_ok = np.arange(4)[None, :] < nv[:, None]
    assert (a["inter"][e_ok] == b["inter"][e_ok]).all()
    for f, ok in (("pos", e_ok), ("grad", e_ok), ("vert", v_ok)):
        x, y = a[f][ok].view(np.uint32), b[f][ok].view(np.uint32)
        assert (x == y).all(), f
    assert (a["qef_err"][v_ok].view(np.uint32) == b["qef_err"][v_ok].view(np.uint32)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("model,depth", [("gyroid-sphere.vm", 7), ("colonnade.vm", 6), ("bear.vm", 5), ("prospero.vm", 5)])
def test_bulk_edge_passes_equal_the_hip_edge_search(model, depth, monkeypatch):
    """The four rounds of the edge search as passes around the assembly bulk interpreter (k_mesh_edge_* + fh_float_eval_*[_t]) against
    k_mesh_edges (the generic interpreter, FHIP_MESH_BULK_EDGES=0): the same records, bit for bit - the values are the same f32 values."""
    import fidget_amd as F
    shape = F.Shape.from_vm(model_path(model))
    monkeypatch.delenv("FHIP_MESH_BULK_EDGES", raising=False)
    a, ca = F.mesh_sample(shape, depth)
    monkeypatch.setenv("FHIP_MESH_BULK_EDGES", "0")
    b, cb = F.mesh_sample(shape, depth)
    monkeypatch.delenv("FHIP_MESH_BULK_EDGES")
    assert ca == cb and len(a) == len(b) > 500
    a, b = a[np.argsort(a["path"], kind="stable")], b[np.argsort(b["path"], kind="stable")]
    ne, nv = a["n_edges"], a["n_verts"]
    assert (ne == b["n_edges"]).all() and (a["mask"] == b["mask"]).all()
    e_ok = np.arange(12)[None, :] < ne[:, None]
    v_ok = np.arange(4)[None, :] < nv[:, None]
    assert (a["inter"][e_ok] == b["inter"][e_ok]).all()
    for f, ok in (("pos", e_ok), ("grad", e_ok), ("vert", v_ok)):
        assert (a[f][ok].view(np.uint32) == b[f][ok].view(np.uint32)).all(), f


@pytest.mark.gpu
@pytest.mark.parametrize("model,depth", [("gyroid-sphere.vm", 8), ("colonnade.vm", 7), ("bear.vm", 6), ("prospero.vm", 7)])
def test_device_assembly_equals_the_host_assembly(model, depth):
    """fhip_mesh_build assembles the octree in HBM (k_oct_*: check_done / collapse / places, mesh_collapse.hpp) and hands the host the
    finished octree; with the option off, the host's threads assemble it from copies of the levels and the leaf records, as
    fhip_mesh_merge does.  Same per-cell functions, same octree: the meshes are equal bit for bit, at sizes the oracle would take
    minutes for (a million cells), collapsed cells on several levels included (colonnade's flat faces)."""
    import fidget_amd as F
    shape = F.Shape.from_vm(model_path(model))
    assert shape.hip.option("mesh_device_assembly") == 1
    tris, verts, counts = F.mesh(shape, depth)
    with shape.hip.options(mesh_device_assembly=0):
        t2, v2, c2 = F.mesh(shape, depth)
    assert counts == c2 and len(tris) > 10000
    assert tris.shape == t2.shape and (tris == t2).all()
    assert verts.shape == v2.shape and (verts.view(np.uint32) == v2.view(np.uint32)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("model,depth", [("prospero.vm", 7), ("colonnade.vm", 8), ("bear.vm", 7), ("prospero.vm", 3), ("colonnade.vm", 4)])
def test_tape_simplification_down_the_octree_leaves_the_mesh_unchanged(model, depth):
    """Tapes of 256 ops and more are simplified once on the way down (octree.rs:546-553; capi_mesh.hpp: at level min(4, depth - 2) every
    ambiguous cell gets the root tape simplified under its own choices, and the cells and leaf samples below it are evaluated with that);
    with the option at 0 everything is evaluated with the root tape, as until round 3.  A min / max decided over a cell is decided the
    same way everywhere inside it, so the two builds give the same cells, the same leaf records and the same mesh - bit for bit."""
    import fidget_amd as F
    shape = F.Shape.from_vm(model_path(model))
    assert shape.hip.option("mesh_simplify_min_ops") == 256 and len(shape) >= 256
    tris, verts, counts = F.mesh(shape, depth)
    with shape.hip.options(mesh_simplify_min_ops=0):
        t2, v2, c2 = F.mesh(shape, depth)
    assert counts == c2
    assert tris.shape == t2.shape and (tris == t2).all()
    assert verts.shape == v2.shape and (verts.view(np.uint32) == v2.view(np.uint32)).all()
    a, ca = F.mesh_sample(shape, min(depth, 6))
    with shape.hip.options(mesh_simplify_min_ops=0):
        b, cb = F.mesh_sample(shape, min(depth, 6))
    assert ca == cb and len(a) == len(b)
    a, b = a[np.argsort(a["path"], kind="stable")], b[np.argsort(b["path"], kind="stable")]
    for f in ("path", "bounds", "mask", "n_edges", "n_verts"):
        assert np.ascontiguousarray(a[f]).tobytes() == np.ascontiguousarray(b[f]).tobytes(), f
    ne, nv = a["n_edges"].astype(np.int64), a["n_verts"].astype(np.int64)
    used_e, used_v = np.arange(12)[None, :] < ne[:, None], np.arange(a["vert"].shape[1])[None, :] < nv[:, None]       # (the slots a record uses; the rest is whatever the buffer held)
    for f in ("inter", "pos", "grad"):
        assert np.ascontiguousarray(a[f][used_e]).tobytes() == np.ascontiguousarray(b[f][used_e]).tobytes(), f
    assert np.ascontiguousarray(a["vert"][used_v]).tobytes() == np.ascontiguousarray(b["vert"][used_v]).tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("model,depth", [("gyroid-sphere.vm", 9), ("colonnade.vm", 8), ("bear.vm", 7), ("prospero.vm", 7), ("tanglecube.vm", 6)])
def test_device_dual_walk_equals_the_host_walk(model, depth):
    """fhip_mesh_build walks the dual on the device (mesh_walk.hpp: the recursion as level arrays in call order, MeshBuilder's numbering
    by atomic minima and prefix sums); with the option off, the host's threads walk the octree copied out of HBM.  Same triangles, same
    vertices, in the same order - at sizes far beyond the oracle's reach (gyroid-sphere at depth 9: 12.8 M triangles)."""
    import fidget_amd as F
    shape = F.Shape.from_vm(model_path(model))
    assert shape.hip.option("mesh_device_walk") == 1
    tris, verts, counts = F.mesh(shape, depth)
    with shape.hip.options(mesh_device_walk=0):
        t2, v2, c2 = F.mesh(shape, depth)
    assert counts == c2 and len(tris) > 1000
    assert tris.shape == t2.shape and (tris == t2).all()
    assert verts.shape == v2.shape and (verts.view(np.uint32) == v2.view(np.uint32)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("model,depth", [("gyroid-sphere.vm", 7), ("bear.vm", 6)])
def test_device_mesh_transcendental_models_identical(model, depth, oracle_mod):
    """BASELINE configuration 5's model (and bear.vm): triangles and vertices identical to the oracle's, element for element - sin / cos /
    exp / ln are the host libm's on the device (until round 4 this went through `match_meshes` with 6-7 of 830 k vertices unmatched)."""
    import fidget_amd as F
    O = oracle_mod
    tris, verts = _same_mesh(F, O, F.Shape.from_vm(model_path(model)), O.Shape.from_vm(model_path(model)), depth)
    assert len(tris) > 10000
    check_for_edge_matching(tris)


@pytest.mark.gpu
def test_device_mesh_camera(oracle_mod):          # fidget/tests/octree.rs:9-30
    import fidget_amd as F
    O = oracle_mod
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] *= 0.5
    m[:3, 3] = 1.0
    cf, co = F.Context(), O.Context()
    _, verts = _same_mesh(F, O, F.Shape(cf, sphere(cf, (1.0, 1.0, 1.0), 0.25)), O.Shape(co, sphere(co, (1.0, 1.0, 1.0), 0.25)), 4, w2m=m)
    assert (np.abs(np.linalg.norm(verts - 1.0, axis=1) - 0.25) < 0.01).all()


@pytest.mark.gpu
def test_device_mesh_gyroid_sphere_manifold(oracle_mod):
    """BASELINE configuration 5's model through the reference's own properties (octree.rs:1561-1594), and identical to the oracle's"""
    import fidget_amd as F
    O = oracle_mod
    tris, verts = _same_mesh(F, O, F.Shape.from_vm(model_path("gyroid-sphere.vm")), O.Shape.from_vm(model_path("gyroid-sphere.vm")), 6)
    check_for_edge_matching(tris)
    assert np.isfinite(verts).all() and (np.abs(verts) <= 1.0).all()


@pytest.mark.parametrize("model,depth,threads", [("gyroid-sphere.vm", 5, 4), ("bear.vm", 5, 3), ("colonnade.vm", 5, 8), ("prospero.vm", 4, 2)])
def test_library_dual_walk_sequential_and_parallel(model, depth, threads, oracle_mod, monkeypatch):
    """The host side of fhip_mesh_build walks the dual on the host's threads (independent sub-walks, vertices numbered by a
    final sequential pass): on the oracle's octree it gives the triangles and vertices of the sequential recursion, which are
    the oracle's own walk_dual, element for element - no device needed (fhip_debug_walk_dual)."""
    import fidget_amd as F
    O = oracle_mod
    oc = O.Octree(O.Shape.from_vm(model_path(model)), depth)
    ref_t, ref_v = oc.walk_dual()
    kinds = {"Invalid": 0, "Empty": 1, "Full": 2, "Branch": 3, "Leaf": 4}
    root = np.array([kinds[oc.root[0]], oc.root[1], oc.root[2]], np.uint32)
    # 2: the host's walk as fhip_mesh_build ran it - cells through a view, the mesh's vertices gathered; 3: the passes fhip_mesh_build runs on
    # the device (mesh_walk.hpp: the recursion as level arrays, numbering by minima and prefix sums), here as loops on the host
    for parallel, th in ((0, threads), (1, threads), (2, threads), (2, 1), (3, 1)):
        monkeypatch.setenv("FHIP_MESH_THREADS", str(th))
        t, v = F.debug_walk_dual(oc.cells, root, oc.verts, parallel)
        assert len(ref_t) > 1000
        assert (t == ref_t).all() and t.shape == ref_t.shape
        assert (v.view(np.uint32) == ref_v.view(np.uint32)).all()


@pytest.mark.parametrize("depth", [16, 18])
def test_dual_walk_passes_beyond_depth_15(depth, oracle_mod):
    """the device walk's cell references held the depth in 4 bits while fhip_mesh_build accepts depths up to 20 - at
    depth 16 the reference named the wrong cell.  Five bits now: a tiny sphere meshed at depth 16 / 18 (few cells, all of them deep)
    through the very passes the device runs (fhip_debug_walk_dual mode 3, plain loops on the host) gives the oracle's walk_dual."""
    import fidget_amd as F
    O = oracle_mod
    c = O.Context()
    oc = O.Octree(O.Shape(c, sphere(c, (0.30001, -0.2, 0.1), 3.0e-4 if depth == 16 else 0.9e-4)), depth)
    ref_t, ref_v = oc.walk_dual()
    kinds = {"Invalid": 0, "Empty": 1, "Full": 2, "Branch": 3, "Leaf": 4}
    root = np.array([kinds[oc.root[0]], oc.root[1], oc.root[2]], np.uint32)
    assert len(ref_t) > 100
    for mode in (0, 3):
        t, v = F.debug_walk_dual(oc.cells, root, oc.verts, mode)
        assert t.shape == ref_t.shape and (t == ref_t).all()
        assert (v.view(np.uint32) == ref_v.view(np.uint32)).all()


@pytest.mark.parametrize("model,depth,threads", [("gyroid-sphere.vm", 5, 1), ("gyroid-sphere.vm", 5, 3), ("gyroid-sphere.vm", 5, 64), ("colonnade.vm", 6, 7),
                                                 ("bear.vm", 4, 5), ("prospero.vm", 4, 2)])
def test_oracle_multithreaded_constructor_gives_the_single_threaded_mesh(model, depth, threads, oracle_mod):
    """Octree::build_inner_mt restated (oracle/src/mesh.hpp build_mt; octree.rs:94-210): sub-cells split off breadth-first until there are
    10 x threads of them, one octree each from a fresh handle on the root tape, spliced with shifted indices, check_done over the split
    cells in reverse.  Another layout of cells and vertices - the same tree: walk_dual gives the single-threaded build's mesh element for
    element, and every reachable cell is the same kind.  (This is what lets the oracle finish BASELINE configuration 5 at depth 10.)"""
    O = oracle_mod
    s = O.Shape.from_vm(model_path(model))
    a = O.Octree(s, depth)
    b = O.Octree(s, depth, threads=threads, keep_samples=False)
    ta, va = a.walk_dual()
    tb, vb = b.walk_dual()
    assert len(ta) > 500 and ta.shape == tb.shape and (ta == tb).all()
    assert (va.view(np.uint32) == vb.view(np.uint32)).all()
    assert a.root[0] == b.root[0]
    c = O.Octree(s, depth, threads=threads)          # (with the sampling records: the same leaves, task by task)
    assert len(c.samples["info"]) == len(a.samples["info"])
    key = lambda smp: np.lexsort(smp["bounds"].T[::-1])
    ka, kc = key(a.samples), key(c.samples)
    for f in ("bounds", "info"):
        assert (c.samples[f][kc] == a.samples[f][ka]).all(), f
    e_ok = np.arange(12)[None, :] < a.samples["info"][ka][:, 1][:, None]            # (entries beyond n_edges are not written)
    assert (c.samples["inter"][kc][e_ok] == a.samples["inter"][ka][e_ok]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("model,depth", [("gyroid-sphere.vm", 9), ("prospero.vm", 8), ("colonnade.vm", 9), ("bear.vm", 8)])
def test_device_mesh_equals_the_oracle_at_depth_8_and_9(model, depth, oracle_mod):
    """Round 4 compared depths 8-9 only with the library's own host walk.  With the oracle's multithreaded constructor these finish in
    seconds on the box's cores: triangles and vertices of fhip_mesh_build identical to the ORACLE's, element for element (bench.py does
    the same at depth 10, BASELINE configuration 5's size)."""
    import fidget_amd as F
    O = oracle_mod
    tris, verts, counts = F.mesh(F.Shape.from_vm(model_path(model)), depth)
    t, v = O.Octree(O.Shape.from_vm(model_path(model)), depth, threads=O.max_threads(), keep_samples=False).walk_dual()
    assert len(tris) > 100000
    assert tris.shape == t.shape and (tris == t).all()
    assert verts.shape == v.shape and (verts.view(np.uint32) == v.view(np.uint32)).all()


# ---- the QEF solve against something that is not itself ------------------------------------------------------------------------
# Product and oracle share the solve's arithmetic (cyclic Jacobi in f64 for nalgebra's f32 SVD, qef.rs:67-126): their equality pins
# the pipeline, not the solve.  tests/qef_independent.py solves every vertex's QEF again with LAPACK's f64 SVD under qef.rs's rank
# rule; the bound is the rounding of an f32 position in units of a leaf cell (coordinates up to 1, cells of 2 / 2^depth), and NO
# vertex may sit a rank decision away (that would be a fraction of a cell).
def _qef_bound(depth):
    return 4 * 2.0 ** -24 / (2.0 / 2 ** depth)


@pytest.mark.parametrize("model,depth", [("gyroid-sphere.vm", 6), ("colonnade.vm", 6), ("bear.vm", 5), ("prospero.vm", 5)])
def test_oracle_vertices_minimise_their_qef_by_an_independent_solve(model, depth, oracle_mod):
    import qef_independent as Q
    from fidget_amd import MESH_LEAF
    O = oracle_mod
    sm = O.Octree(O.Shape.from_vm(model_path(model)), depth).samples
    recs = np.zeros(len(sm["info"]), MESH_LEAF)
    recs["bounds"] = sm["bounds"]; recs["mask"] = sm["info"][:, 0]; recs["n_edges"] = sm["info"][:, 1]; recs["n_verts"] = sm["info"][:, 2]
    recs["pos"] = sm["pos"]; recs["grad"] = sm["grad"]; recs["vert"] = sm["vert"]
    q = Q.check(recs, Q.per_vertex_counts(O.mdc_table))
    assert q["vertices"] > 1000 and q["max_deviation_cell_fraction"] < _qef_bound(depth), q


@pytest.mark.gpu
@pytest.mark.parametrize("model,depth", [("gyroid-sphere.vm", 8), ("colonnade.vm", 8), ("bear.vm", 7), ("prospero.vm", 7)])
def test_device_vertices_minimise_their_qef_by_an_independent_solve(model, depth, oracle_mod):
    import fidget_amd as F
    import qef_independent as Q
    recs, _ = F.mesh_sample(F.Shape.from_vm(model_path(model)), depth)
    q = Q.check(recs, Q.per_vertex_counts(oracle_mod.mdc_table))
    print(model, depth, q)
    assert q["vertices"] > 10000 and q["max_deviation_cell_fraction"] < _qef_bound(depth) and q["over_1e-4_of_a_cell"] == 0, q
