"""The device's octree assembly (fidget_amd/csrc/mesh_collapse.hpp: check_done / collapsible / merged Hermite data per cell, level by
level, then the places of vertices and blocks) built for the host and run on the ORACLE's cell classes and leaf samples: the octree -
root, blocks of cells, vertices - must be the oracle's (fidget_mesh::Octree::build, octree.rs), element for element.  No GPU:
the kernels of mesh.hip call the same functions, one thread per item (tests/test_mesh.py compares their result on the device)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import model_path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_build", "mesh_assembly_host.cpp")
CSRC = os.path.join(ROOT, "fidget_amd", "csrc")


@pytest.fixture(scope="module")
def asm_lib():
    out = os.path.join(ROOT, "tests", "host_build", "_build")
    os.makedirs(out, exist_ok=True)
    san = os.environ.get("FIDGET_SANITIZE") == "1"      # tests/test_sanitizers.py: the same sources under ASan + UBSan
    so = os.path.join(out, "libmesh_assembly_host_san.so" if san else "libmesh_assembly_host.so")
    deps = [SRC] + [os.path.join(CSRC, f) for f in ("mesh_collapse.hpp", "mesh_qef.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"] if san else ["-O2"]
        subprocess.check_call(["g++", "-std=c++17"] + flags + ["-ffp-contract=off", "-fPIC", "-shared", "-I", CSRC, SRC, "-o", so])
    lib = C.CDLL(so)
    lib.fh_asm_run.restype = C.c_void_p
    lib.fh_asm_run.argtypes = [C.c_uint32, C.c_uint32] + [C.c_void_p] * 6 + [C.c_uint32, C.c_void_p, C.c_void_p]
    lib.fh_asm_counts.argtypes = [C.c_void_p, C.c_void_p]
    lib.fh_asm_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fh_asm_free.argtypes = [C.c_void_p]
    lib.fh_asm_sizes.restype = C.c_uint32
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


MDC = np.dtype([("n_edges", np.uint8, 256), ("n_verts", np.uint8, 256), ("per_vert", np.uint8, (256, 4)), ("edge", np.uint8, (256, 12, 2))])


def mdc_table(O):
    T = np.zeros(1, MDC)
    for m in range(256):
        v2e, _ = O.mdc_table(m)
        T["n_verts"][0, m] = len(v2e)
        n = 0
        for vi, es in enumerate(v2e):
            T["per_vert"][0, m, vi] = len(es)
            for (a, b) in es:
                T["edge"][0, m, n] = (a, b)
                n += 1
        T["n_edges"][0, m] = n
    return T


def levels_from_the_oracle(O, shape, depth, rng):
    """What k_mesh_cells / k_mesh_leaf leave in HBM, from the oracle: per level the cells' classes and slots (slots in an arbitrary order,
    as the device's atomic counter hands them out), the ambiguous cells' bounds, and the leaf records."""
    from fidget_amd import MESH_LEAF
    o = O.Octree(shape, depth)
    sm = o.samples
    by_bounds = {tuple(sm["bounds"][i].view(np.uint32).tolist()): i for i in range(len(sm["info"]))}
    T = mdc_table(O)
    levels = []
    cells = [np.array([-1, 1, -1, 1, -1, 1], np.float32)]
    recs = None
    n_eval = 0
    for d in range(depth + 1):
        cls = np.zeros(len(cells), np.uint8)
        for i, b in enumerate(cells):
            (lo, hi), _ = shape.eval_interval((float(b[0]), float(b[1])), (float(b[2]), float(b[3])), (float(b[4]), float(b[5])))
            cls[i] = 2 if hi < 0 else (1 if lo > 0 else 3)
        n_eval += len(cells)
        amb = np.flatnonzero(cls == 3)
        order = rng.permutation(len(amb))
        slot = np.full(len(cells), 0xFFFFFFFF, np.uint32)
        slot[amb] = order
        amb_bounds = np.zeros((len(amb), 6), np.float32)
        amb_bounds[order] = np.array([cells[i] for i in amb], np.float32).reshape(-1, 6)
        levels.append((cls, slot, amb_bounds))
        if d == depth:
            recs = np.zeros(len(amb), MESH_LEAF)
            for s in range(len(amb)):
                b = amb_bounds[s]
                r = recs[s]
                r["bounds"] = b
                i = by_bounds.get(tuple(b.view(np.uint32).tolist()))
                if i is None:       # an ambiguous cell whose corners are all inside or all outside
                    m = 0
                    for c in range(8):
                        v, _ = shape.eval_point(float(b[c & 1]), float(b[2 + ((c >> 1) & 1)]), float(b[4 + ((c >> 2) & 1)]))
                        m |= (1 << c) if v < 0 else 0
                    assert m in (0, 255), m
                    r["mask"] = m
                    continue
                mask, ne, nv = (int(v) for v in sm["info"][i])
                r["mask"], r["n_edges"], r["n_verts"] = mask, ne, nv
                r["inter"][:ne] = sm["inter"][i, :ne]; r["pos"][:ne] = sm["pos"][i, :ne]; r["grad"][:ne] = sm["grad"][i, :ne]; r["vert"][:nv] = sm["vert"][i, :nv]
                ii = 0
                for vi in range(nv):     # the vertices' QEF errors (octree.rs:805-848), as k_mesh_leaf stores them
                    pts, grs, forced = [], [], False
                    for _ in range(T["per_vert"][0, mask, vi]):
                        k = min(ii, 11)
                        if np.isnan(sm["grad"][i, k]).any():
                            forced = True
                            break
                        pts.append(sm["pos"][i, k]); grs.append(sm["grad"][i, k]); ii += 1
                    r["qef_err"][vi] = -2.0 if forced else O.qef_solve(np.array(pts), np.array(grs))[1]
            break
        if len(amb) == 0:
            break
        nxt = [None] * (8 * len(amb))
        for s in range(len(amb)):
            b = amb_bounds[s]
            mid = [np.float32((b[2 * k] + b[2 * k + 1]) / np.float32(2.0)) for k in range(3)]
            for c in range(8):
                cb = np.zeros(6, np.float32)
                for k in range(3):
                    cb[2 * k], cb[2 * k + 1] = (mid[k], b[2 * k + 1]) if c & (1 << k) else (b[2 * k], mid[k])
                nxt[8 * s + c] = cb
        cells = nxt
    assert n_eval == o.interval_evals
    return o, levels, recs, T


def assemble(lib, depth, levels, recs, T, mat=None):
    n_cells = np.array([len(l[0]) for l in levels], np.uint32)
    n_amb = np.array([len(l[2]) for l in levels], np.uint32)
    cls = np.concatenate([l[0] for l in levels]); slot = np.concatenate([l[1] for l in levels])
    bounds = np.ascontiguousarray(np.concatenate([l[2].reshape(-1, 6) for l in levels]), np.float32)
    n_rec = 0 if recs is None else len(recs)
    rec = np.zeros(1, np.uint8) if n_rec == 0 else recs
    h = lib.fh_asm_run(depth, len(levels), _p(n_cells), _p(cls), _p(slot), _p(n_amb), _p(bounds), _p(rec), n_rec, _p(T), _p(mat))
    assert h
    c = np.zeros(6, np.uint32)
    lib.fh_asm_counts(h, _p(c))
    cells = np.zeros((int(c[3]), 8, 3), np.uint32); verts = np.zeros((int(c[4]), 3), np.float32)
    lib.fh_asm_copy(h, _p(cells), _p(verts))
    lib.fh_asm_free(h)
    return tuple(int(v) for v in c[:3]), cells, verts, int(c[5])


def test_record_sizes(asm_lib):
    from fidget_amd import MESH_LEAF
    assert asm_lib.fh_asm_sizes(0) == MESH_LEAF.itemsize == 528 and asm_lib.fh_asm_sizes(1) == MDC.itemsize


def _shapes(O):
    def sphere():
        c = O.Context()
        x, y, z = c.x(), c.y(), c.z()
        return O.Shape(c, c.sub(c.sqrt(c.add(c.add(c.square(x), c.square(y)), c.square(z))), c.constant(0.6)))

    def cube():      # flat faces: whole subtrees collapse
        c = O.Context()
        f = lambda a: c.sub(c.abs(a), c.constant(0.45))
        return O.Shape(c, c.max(c.max(f(c.x()), f(c.y())), f(c.z())))

    def slab():      # a plane off the cell boundaries: everything collapses, up to the root's children
        c = O.Context()
        return O.Shape(c, c.sub(c.add(c.mul(c.x(), c.constant(0.3)), c.z()), c.constant(0.13)))
    return {"sphere": sphere, "cube": cube, "slab": slab}


CASES = [("sphere", 0), ("sphere", 1), ("sphere", 4), ("cube", 3), ("cube", 5), ("slab", 4), ("colonnade.vm", 5), ("prospero.vm", 4), ("gyroid-sphere.vm", 5),
         ("bear.vm", 4)]


@pytest.mark.parametrize("name,depth", CASES)
def test_assembly_passes_give_the_oracles_octree(name, depth, oracle_mod, asm_lib):
    O = oracle_mod
    shape = O.Shape.from_vm(model_path(name)) if name.endswith(".vm") else _shapes(O)[name]()
    o, levels, recs, T = levels_from_the_oracle(O, shape, depth, np.random.default_rng(depth * 7 + len(name)))
    root, cells, verts, collapsed = assemble(asm_lib, depth, levels, recs, T)
    kinds = {"Invalid": 0, "Empty": 1, "Full": 2, "Branch": 3, "Leaf": 4}
    assert root == (kinds[o.root[0]], o.root[1], o.root[2]), (root, o.root)
    assert cells.shape == o.cells.shape and (cells == o.cells).all()
    assert verts.shape == o.verts.shape and (verts.view(np.uint32) == o.verts.view(np.uint32)).all()
    if name in ("cube", "slab") and depth >= 3:
        assert collapsed > 0          # (the case is here for the collapse)


def test_vertices_go_back_to_model_space(oracle_mod, asm_lib):
    """octree.rs:58-65 on every vertex the passes write (here with an arbitrary matrix over an octree built without one: the
    classification side of a camera is the device tests' business)"""
    O = oracle_mod
    shape = _shapes(O)["cube"]()
    o, levels, recs, T = levels_from_the_oracle(O, shape, 4, np.random.default_rng(3))
    mat = np.array([[0.9, 0.1, 0, 0.05], [-0.1, 0.8, 0.2, 0], [0, -0.2, 1.1, -0.1], [0.01, 0, 0.02, 1.0]], np.float32)
    _, cells, verts, _ = assemble(asm_lib, 4, levels, recs, T, mat=np.ascontiguousarray(mat.reshape(-1)))
    want = np.zeros_like(o.verts)
    m = mat.reshape(-1)
    for i, (x, y, z) in enumerate(o.verts):
        n = np.float32(np.float32(np.float32(m[12] * x) + np.float32(m[13] * y)) + np.float32(m[14] * z)) + m[15]
        r = [np.float32(np.float32(np.float32(m[4 * k] * x) + np.float32(m[4 * k + 1] * y)) + np.float32(m[4 * k + 2] * z)) + m[4 * k + 3] for k in range(3)]
        want[i] = [np.float32(v / n) for v in r] if n != 0 else r
    assert (verts.view(np.uint32) == want.view(np.uint32)).all()
    assert (cells == o.cells).all()
