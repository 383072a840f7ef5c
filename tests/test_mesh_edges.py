"""The edge search of the mesh path as per-edge steps (fidget_amd/csrc/mesh_edges.hpp: what k_mesh_edge_begin / _points / _narrow / _end run,
one lane per edge or sample, around the assembly bulk interpreter) built for the host and driven with the ORACLE's f32 evaluator: the
intersections - u16 cell coordinates and positions - are the oracle's leaf samples (fidget-mesh octree.rs:662-803), bit for bit.  No GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import model_path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_build", "mesh_edges_host.cpp")
CSRC = os.path.join(ROOT, "fidget_amd", "csrc")


@pytest.fixture(scope="module")
def edges_lib():
    out = os.path.join(ROOT, "tests", "host_build", "_build")
    os.makedirs(out, exist_ok=True)
    san = os.environ.get("FIDGET_SANITIZE") == "1"      # tests/test_sanitizers.py: the same sources under ASan + UBSan
    so = os.path.join(out, "libmesh_edges_host_san.so" if san else "libmesh_edges_host.so")
    deps = [SRC] + [os.path.join(CSRC, f) for f in ("mesh_edges.hpp", "mesh_qef.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"] if san else ["-O2"]
        subprocess.check_call(["g++", "-std=c++17"] + flags + ["-ffp-contract=off", "-fPIC", "-shared", "-I", CSRC, SRC, "-o", so])
    lib = C.CDLL(so)
    lib.fh_edge_begin.argtypes = [C.c_int, C.c_int, C.c_void_p]
    lib.fh_edge_points.argtypes = [C.c_void_p] * 3
    lib.fh_edge_narrow.argtypes = [C.c_void_p, C.c_uint32]
    lib.fh_edge_end.argtypes = [C.c_void_p] * 4
    lib.fh_edge_bracket_bytes.restype = C.c_uint32
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("model,depth", [("colonnade.vm", 4), ("prospero.vm", 3), ("tanglecube.vm", 4)])
def test_edge_steps_give_the_oracles_intersections(model, depth, oracle_mod, edges_lib):
    O = oracle_mod
    assert edges_lib.fh_edge_bracket_bytes() == 12
    shape = O.Shape.from_vm(model_path(model))
    sm = O.Octree(shape, depth).samples
    n_checked = 0
    for i in range(0, len(sm["info"]), max(1, len(sm["info"]) // 150)):
        mask, ne, nv = (int(v) for v in sm["info"][i])
        bounds = np.ascontiguousarray(sm["bounds"][i], np.float32)
        edges = [e for vs in O.mdc_table(mask)[0] for e in vs]
        assert len(edges) == ne
        for k, (st, en) in enumerate(edges):
            br = np.zeros(6, np.uint16)
            edges_lib.fh_edge_begin(st, en, _p(br))
            for _ in range(4):
                xyz = np.zeros((16, 3), np.float32)
                edges_lib.fh_edge_points(_p(br), _p(bounds), _p(xyz))
                v = np.asarray(shape.eval_float_slice(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy()), np.float32).reshape(-1)[:16]
                m16 = int(sum(1 << j for j in range(16) if v[j] >= 0))
                edges_lib.fh_edge_narrow(_p(br), m16)
            q = np.zeros(3, np.uint16); pos = np.zeros(3, np.float32)
            edges_lib.fh_edge_end(_p(br), _p(bounds), _p(q), _p(pos))
            assert (q == sm["inter"][i, k]).all(), (i, k, q, sm["inter"][i, k])
            assert (pos.view(np.uint32) == sm["pos"][i, k].view(np.uint32)).all()
            n_checked += 1
    assert n_checked > 200
