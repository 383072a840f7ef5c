"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes wrapper over ``oracle/_build/liboracle.so``, the CPU restatement (C++17,
``oracle/src``) of the reference's ``VmShape`` path: graph -> SSA tape ->
register tape -> interval / point / bulk-f32 / bulk-grad interpreters ->
``VmData::simplify`` -> ``fidget_raster::{pixel,voxel}::render``.

The reference is Rust and cannot be compiled in this environment (no rustc /
cargo, crates not vendored), so there is no ``oracle/_ref``; the restatement is
pinned against the reference's own known-answer tests and golden images in
``tests/test_kat_*.py``, ``tests/test_compiler_kat.py``, ``tests/test_reference_units.py``,
``tests/test_render_golden.py`` and ``tests/test_mesh.py`` (see each test's file:line citation).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package.  ``fidget_amd`` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FIDGET_SANITIZE=1 (tests/test_sanitizers.py): the same sources under AddressSanitizer + UndefinedBehaviorSanitizer, as a library
# of its own; the process must have been started with the sanitizer runtimes preloaded
_SANITIZE = os.environ.get("FIDGET_SANITIZE") == "1"
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle_san.so" if _SANITIZE else "liboracle.so")

UNARY = ["neg", "abs", "recip", "sqrt", "square", "floor", "ceil", "round", "sin", "cos", "tan",
         "asin", "acos", "atan", "exp", "ln", "not", "rand"]
BINARY = ["add", "sub", "mul", "div", "atan2", "min", "max", "compare", "mod", "and", "or", "mix"]

# Opc numbering in oracle/src/compiler.hpp
OPC = ["Output", "Input", "CopyReg", "CopyImm", "Neg", "Abs", "Recip", "Sqrt", "Square", "Floor",
       "Ceil", "Round", "Sin", "Cos", "Tan", "Asin", "Acos", "Atan", "Exp", "Ln", "Not", "Rand",
       "Add", "Sub", "Mul", "Div", "Atan2", "Compare", "Mix", "Mod", "Min", "Max", "And", "Or",
       "Load", "Store"]
FORM = ["", "Reg", "RegReg", "RegImm", "ImmReg"]

UNKNOWN, LEFT, RIGHT, BOTH = 0, 1, 2, 3

SIMPLIFY_REFERENCE, SIMPLIFY_NEVER, SIMPLIFY_ALWAYS = 0, 1, 2

VM_TILES_3D = [128, 64, 32, 16, 8]   # fidget-core/src/vm/mod.rs:251-253
VM_TILES_2D = [128, 32, 8]           # fidget-core/src/vm/mod.rs:255-257


def build(force=False):
    """Compile the oracle with gcc (no GPU needed)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, "src", f)) > os.path.getmtime(_LIB_PATH)
        for f in os.listdir(os.path.join(_HERE, "src"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["san"] if _SANITIZE else []) + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, u32, i32, f32 = C.c_void_p, C.c_uint32, C.c_int, C.c_float
        sig = {
            "orc_ctx_new": (vp, []),
            "orc_ctx_free": (None, [vp]),
            "orc_ctx_len": (u32, [vp]),
            "orc_ctx_x": (u32, [vp]), "orc_ctx_y": (u32, [vp]), "orc_ctx_z": (u32, [vp]),
            "orc_ctx_var": (u32, [vp, C.c_uint64]),
            "orc_ctx_constant": (u32, [vp, f32]),
            "orc_ctx_unary": (u32, [vp, i32, u32]),
            "orc_ctx_binary": (u32, [vp, i32, u32, u32]),
            "orc_ctx_less_than": (u32, [vp, u32, u32]),
            "orc_ctx_less_than_or_equal": (u32, [vp, u32, u32]),
            "orc_ctx_if_nonzero_else": (u32, [vp, u32, u32, u32]),
            "orc_ctx_from_text": (u32, [vp, C.c_char_p]),
            "orc_ctx_eval_xyz": (f32, [vp, u32, f32, f32, f32]),
            "orc_shape_new": (vp, [vp, vp, i32, u32]),
            "orc_shape_free": (None, [vp]),
            "orc_shape_len": (u32, [vp]), "orc_shape_ssa_len": (u32, [vp]),
            "orc_shape_choice_count": (u32, [vp]), "orc_shape_output_count": (u32, [vp]),
            "orc_shape_slot_count": (u32, [vp]), "orc_shape_var_count": (u32, [vp]),
            "orc_shape_axis_index": (i32, [vp, i32]),
            "orc_shape_var_index": (i32, [vp, C.c_uint64]),
            "orc_shape_dump": (u32, [vp, i32, vp, vp, u32]),
            "orc_shape_simplify": (vp, [vp, vp, u32, u32]),
            "orc_shape_bytecode": (u32, [vp, vp, u32, vp, vp]),
            "orc_eval_interval": (i32, [vp, vp, u32, vp, vp]),
            "orc_eval_point": (i32, [vp, vp, u32, vp, vp]),
            "orc_eval_float_slice": (i32, [vp, vp, u32, u32, vp]),
            "orc_eval_grad_slice": (i32, [vp, vp, u32, u32, vp]),
            "orc_invalid_intervals": (C.c_uint64, []),
            "orc_reset_invalid_intervals": (None, []),
            "orc_screen_to_world": (None, [vp, i32, vp]),
            "orc_mat_mul": (None, [vp, vp, i32, vp]),
            "orc_transform_point": (None, [vp, f32, f32, f32, vp]),
            "orc_transform_interval": (None, [vp, vp, vp]),
            "orc_stats_names": (C.c_char_p, []),
            "orc_stats_count": (u32, []),
            "orc_render2d": (i32, [vp, vp, u32, u32, f32, i32, vp, u32, i32, i32, vp, vp, vp, vp, vp, u32]),
            "orc_render3d": (i32, [vp, vp, u32, u32, u32, vp, u32, i32, i32, vp, vp, vp, vp, vp, u32]),
            "orc_max_threads": (i32, []),
            "orc_math_unary": (None, [i32, u32, u32, C.c_uint64, vp]),
            "orc_mesh_build": (vp, [vp, vp, u32, i32, vp, vp, u32]), "orc_mesh_free": (None, [vp]),
            "orc_mesh_build_mt": (vp, [vp, vp, u32, i32, vp, vp, u32, i32, i32]),
            "orc_mesh_counts": (None, [vp, vp]), "orc_mesh_verts": (None, [vp, vp]), "orc_mesh_cells": (None, [vp, vp]),
            "orc_mesh_samples": (None, [vp, vp, vp, vp, vp, vp, vp]), "orc_mesh_walk_dual": (None, [vp, vp]),
            "orc_mesh_dual_copy": (None, [vp, vp, vp]), "orc_mesh_table": (None, [i32, vp, vp]),
            "orc_qef_solve": (None, [vp, vp, i32, vp, vp]),
            "orc_fx_denoise_normals": (None, [vp, i32, i32, vp]),
            "orc_fx_compute_ssao": (None, [vp, i32, i32, i32, vp, i32, vp, i32, vp]),
            "orc_fx_blur_ssao": (None, [vp, i32, i32, vp]),
            "orc_fx_apply_shading": (None, [vp, i32, i32, i32, vp, vp]),
            "orc_fx_to_rgba_bitmap": (None, [vp, C.c_uint64, i32, vp]),
            "orc_fx_to_debug_bitmap": (None, [vp, C.c_uint64, vp]),
            "orc_fx_to_rgba_distance": (None, [vp, C.c_uint64, vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


BAD_NODE = 0xFFFFFFFF


class BadNode(Exception):
    pass


class Context:
    """Mirror of ``fidget_core::Context`` (context/mod.rs)."""

    def __init__(self):
        self._h = lib().orc_ctx_new()

    def __del__(self):
        try:
            lib().orc_ctx_free(self._h)
        except Exception:
            pass

    def __len__(self):
        return lib().orc_ctx_len(self._h)

    def _node(self, v):
        if isinstance(v, (float, int)) and not isinstance(v, Node):
            return self.constant(float(v))
        return v

    def x(self): return Node(lib().orc_ctx_x(self._h))
    def y(self): return Node(lib().orc_ctx_y(self._h))
    def z(self): return Node(lib().orc_ctx_z(self._h))
    def var(self, index): return Node(lib().orc_ctx_var(self._h, int(index)))
    def constant(self, f): return Node(lib().orc_ctx_constant(self._h, float(f)))

    def _un(self, name, a):
        r = lib().orc_ctx_unary(self._h, UNARY.index(name), int(self._node(a)))
        if r == BAD_NODE:
            raise BadNode()
        return Node(r)

    def _bin(self, name, a, b):
        r = lib().orc_ctx_binary(self._h, BINARY.index(name), int(self._node(a)), int(self._node(b)))
        if r == BAD_NODE:
            raise BadNode()
        return Node(r)

    def less_than(self, a, b):
        return Node(lib().orc_ctx_less_than(self._h, int(self._node(a)), int(self._node(b))))

    def less_than_or_equal(self, a, b):
        return Node(lib().orc_ctx_less_than_or_equal(self._h, int(self._node(a)), int(self._node(b))))

    def if_nonzero_else(self, c, a, b):
        return Node(lib().orc_ctx_if_nonzero_else(self._h, int(self._node(c)), int(self._node(a)), int(self._node(b))))

    def eval_xyz(self, node, x, y, z):
        return lib().orc_ctx_eval_xyz(self._h, int(node), x, y, z)

    @staticmethod
    def from_text(text):
        ctx = Context()
        if isinstance(text, str):
            text = text.encode()
        r = lib().orc_ctx_from_text(ctx._h, text)
        if r == BAD_NODE:
            raise ValueError("parse error")
        return ctx, Node(r)


class Node(int):
    pass


def _mk_un(name):
    def f(self, a):
        return self._un(name, a)
    return f


def _mk_bin(name):
    def f(self, a, b):
        return self._bin(name, a, b)
    return f


for _n in UNARY:
    setattr(Context, {"not": "not_"}.get(_n, _n), _mk_un(_n))
for _n in BINARY:
    setattr(Context, {"and": "and_", "or": "or_", "mod": "modulo"}.get(_n, _n), _mk_bin(_n))


class Shape:
    """``VmData<N>`` + the four VM evaluators (vm/data.rs, vm/mod.rs)."""

    def __init__(self, ctx=None, node=None, n_regs=255, _h=None, roots=None):
        self.n_regs = n_regs
        if _h is not None:
            self._h = _h
            return
        if roots is None:
            roots = [node]
        r = np.array([int(n) for n in roots], dtype=np.uint32)
        self._h = lib().orc_shape_new(ctx._h, _p(r), len(r), n_regs)
        if not self._h:
            raise BadNode()

    def __del__(self):
        try:
            lib().orc_shape_free(self._h)
        except Exception:
            pass

    @staticmethod
    def from_vm(path_or_text, n_regs=255):
        text = open(path_or_text).read() if os.path.exists(path_or_text) else path_or_text
        ctx, root = Context.from_text(text)
        return Shape(ctx, root, n_regs)

    # sizes ---------------------------------------------------------------
    def size(self): return lib().orc_shape_len(self._h)
    __len__ = size
    def ssa_len(self): return lib().orc_shape_ssa_len(self._h)
    def choice_count(self): return lib().orc_shape_choice_count(self._h)
    def output_count(self): return lib().orc_shape_output_count(self._h)
    def slot_count(self): return lib().orc_shape_slot_count(self._h)
    def var_count(self): return lib().orc_shape_var_count(self._h)
    def axis_index(self, axis): return lib().orc_shape_axis_index(self._h, axis)
    def var_index(self, index): return lib().orc_shape_var_index(self._h, int(index))

    def dump(self, which):
        """List of (op, form, out, a, b, idx, imm_bits); which=0 SSA (root first), 1 asm (eval order)."""
        n = lib().orc_shape_dump(self._h, which, None, None, 0)
        rec = np.zeros((n, 6), dtype=np.uint32)
        imm = np.zeros(n, dtype=np.uint32)
        lib().orc_shape_dump(self._h, which, _p(rec), _p(imm), n)
        return [(OPC[r[0]], FORM[r[1]], int(r[2]), int(r[3]), int(r[4]), int(r[5]), int(i)) for r, i in zip(rec, imm)]

    def ssa_ops(self): return self.dump(0)
    def asm_ops(self): return self.dump(1)

    def simplify(self, choices, n_regs=None):
        c = np.asarray(choices, dtype=np.uint8)
        n_regs = self.n_regs if n_regs is None else n_regs
        h = lib().orc_shape_simplify(self._h, _p(c), len(c), n_regs)
        if not h:
            raise ValueError("BadChoiceSlice")
        return Shape(_h=h, n_regs=n_regs)

    def bytecode(self):
        rc, mc = C.c_uint32(0), C.c_uint32(0)
        n = lib().orc_shape_bytecode(self._h, None, 0, C.byref(rc), C.byref(mc))
        if n == 0:
            raise ValueError("ReservedRegister")
        w = np.zeros(n, dtype=np.uint32)
        lib().orc_shape_bytecode(self._h, _p(w), n, C.byref(rc), C.byref(mc))
        return w, rc.value, mc.value

    # evaluators ------------------------------------------------------------
    def _xyz_vars(self, x, y, z, extra=None):
        n = max(self.var_count(), 1)
        vs = [None] * n
        for axis, v in enumerate((x, y, z)):
            i = self.axis_index(axis)
            if i >= 0:
                vs[i] = v
        for k, v in (extra or {}).items():
            i = self.var_index(k)
            if i >= 0:
                vs[i] = v
        return vs

    def eval_interval_raw(self, vars_):
        """vars_: list of (lo, hi).  Returns (list of (lo,hi), choices or None)."""
        v = np.array(vars_, dtype=np.float32).reshape(-1, 2)
        out = np.zeros((max(self.output_count(), 1), 2), dtype=np.float32)
        ch = np.zeros(max(self.choice_count(), 1), dtype=np.uint8)
        r = lib().orc_eval_interval(self._h, _p(v), len(v), _p(out), _p(ch))
        if r < 0:
            raise ValueError("BadVarSlice")
        return out[: self.output_count()], (ch[: self.choice_count()].copy() if r == 1 else None)

    def eval_interval(self, x, y, z, extra=None):
        vs = self._xyz_vars(x, y, z, extra)
        vs = [(0.0, 0.0) if v is None else ((v, v) if np.isscalar(v) else tuple(v)) for v in vs]
        out, tr = self.eval_interval_raw(vs)
        return (float(out[0][0]), float(out[0][1])), tr

    def eval_interval_batch(self, batch):
        """batch: list of per-variable (lo, hi) lists (None = [0,0]).  Returns [((lo,hi), trace|None)]."""
        out = []
        for vs in batch:
            vs = [(0.0, 0.0) if v is None else v for v in vs]
            o, tr = self.eval_interval_raw(vs)
            out.append(((float(o[0][0]), float(o[0][1])), tr))
        return out

    def eval_point_raw(self, vars_):
        v = np.array(vars_, dtype=np.float32)
        out = np.zeros(max(self.output_count(), 1), dtype=np.float32)
        ch = np.zeros(max(self.choice_count(), 1), dtype=np.uint8)
        r = lib().orc_eval_point(self._h, _p(v), len(v), _p(out), _p(ch))
        if r < 0:
            raise ValueError("BadVarSlice")
        return out[: self.output_count()], (ch[: self.choice_count()].copy() if r == 1 else None)

    def eval_point(self, x, y, z, extra=None):
        vs = [0.0 if v is None else v for v in self._xyz_vars(x, y, z, extra)]
        out, tr = self.eval_point_raw(vs)
        return float(out[0]), tr

    def eval_float_slice_raw(self, arrays):
        arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in arrays]
        n = len(arrs[0]) if arrs else 0
        if any(len(a) != n for a in arrs):
            raise ValueError("MismatchedSlices")
        ptrs = (C.c_void_p * max(len(arrs), 1))(*[a.ctypes.data for a in arrs])
        out = np.zeros((max(self.output_count(), 1), n), dtype=np.float32)
        r = lib().orc_eval_float_slice(self._h, ptrs, len(arrs), n, _p(out))
        if r < 0:
            raise ValueError("BadVarSlice")
        return out[: self.output_count()]

    def eval_float_slice(self, x, y, z, extra=None):
        x, y, z = (np.asarray(a, dtype=np.float32) for a in (x, y, z))
        n = len(x)
        vs = self._xyz_vars(x, y, z, extra)
        vs = [np.zeros(n, np.float32) if v is None else (np.full(n, v, np.float32) if np.isscalar(v) else v) for v in vs]
        return self.eval_float_slice_raw(vs)[0]

    def eval_grad_slice_raw(self, arrays):
        arrs = [np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 4) for a in arrays]
        n = len(arrs[0]) if arrs else 0
        ptrs = (C.c_void_p * max(len(arrs), 1))(*[a.ctypes.data for a in arrs])
        out = np.zeros((max(self.output_count(), 1), n, 4), dtype=np.float32)
        r = lib().orc_eval_grad_slice(self._h, ptrs, len(arrs), n, _p(out))
        if r < 0:
            raise ValueError("BadVarSlice")
        return out[: self.output_count()]

    def eval_grad_slice(self, x, y, z, extra=None):
        """x, y, z: float arrays; seeded d/dx, d/dy, d/dz = identity (grad_slice.rs tests)."""
        x, y, z = (np.asarray(a, dtype=np.float32) for a in (x, y, z))
        n = len(x)

        def seed(v, k):
            g = np.zeros((n, 4), np.float32)
            g[:, 0] = v
            g[:, 1 + k] = 1.0
            return g
        vs = self._xyz_vars(seed(x, 0), seed(y, 1), seed(z, 2), extra)
        out = []
        for v in vs:
            if v is None:
                out.append(np.zeros((n, 4), np.float32))
            elif np.isscalar(v):
                g = np.zeros((n, 4), np.float32)
                g[:, 0] = v
                out.append(g)
            else:
                out.append(v)
        return self.eval_grad_slice_raw(out)[0]


def stats_names():
    return lib().orc_stats_names().decode().split(",")


def screen_to_world(size):
    n = len(size)
    s = np.array(size, dtype=np.uint32)
    out = np.zeros((n + 1, n + 1), dtype=np.float32)
    lib().orc_screen_to_world(_p(s), n, _p(out))
    return out


def mat_mul(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    out = np.zeros_like(a)
    lib().orc_mat_mul(_p(a), _p(b), a.shape[0], _p(out))
    return out


def transform_point(mat4, x, y, z):
    m = np.ascontiguousarray(mat4, np.float32)
    out = np.zeros(3, np.float32)
    lib().orc_transform_point(_p(m), x, y, z, _p(out))
    return out


def lift_2d(m3):
    """pixel.rs:281-285"""
    m3 = np.asarray(m3, np.float32)
    m = np.zeros((4, 4), np.float32)
    m[0, :2] = m3[0, :2]; m[0, 3] = m3[0, 2]
    m[1, :2] = m3[1, :2]; m[1, 3] = m3[1, 2]
    m[2, 2] = 1.0
    m[3, :2] = m3[2, :2]; m[3, 3] = m3[2, 2]
    return m


def _vars(vars_):
    vars_ = vars_ or {}
    k = np.array(list(vars_.keys()), dtype=np.uint64)
    v = np.array(list(vars_.values()), dtype=np.float32)
    return k, v


def render2d(shape, width, height=None, z=0.0, pixel_perfect=False, world_to_model=None,
             tile_sizes=None, mode=SIMPLIFY_REFERENCE, threads=0, vars=None):
    """``fidget_raster::pixel::render``.  Returns (float32 image [h,w] of raw pixels, stats dict, seconds)."""
    height = width if height is None else height
    ts = np.array(tile_sizes or VM_TILES_2D, dtype=np.uint32)
    out = np.zeros((height, width), dtype=np.float32)
    stats = np.zeros(lib().orc_stats_count(), dtype=np.uint64)
    secs = C.c_double(0)
    w2m = None if world_to_model is None else np.ascontiguousarray(world_to_model, np.float32)
    vk, vv = _vars(vars)
    r = lib().orc_render2d(shape._h, _p(w2m), width, height, z, int(pixel_perfect), _p(ts), len(ts), mode, threads,
                           _p(out), _p(stats), C.byref(secs), _p(vk), _p(vv), len(vk))
    if r != 0:
        raise ValueError("MissingVar")
    return out, dict(zip(stats_names(), (int(s) for s in stats))), secs.value


GEOMETRY_PIXEL = np.dtype([("normal", np.float32, 3), ("depth", np.uint32)])


def render3d(shape, width, height=None, depth=None, world_to_model=None, tile_sizes=None,
             mode=SIMPLIFY_REFERENCE, threads=0, vars=None):
    """``fidget_raster::voxel::render``.  Returns (GeometryPixel image [h,w], stats, seconds)."""
    height = width if height is None else height
    depth = width if depth is None else depth
    ts = np.array(tile_sizes or VM_TILES_3D, dtype=np.uint32)
    out = np.zeros((height, width), dtype=GEOMETRY_PIXEL)
    stats = np.zeros(lib().orc_stats_count(), dtype=np.uint64)
    secs = C.c_double(0)
    w2m = None if world_to_model is None else np.ascontiguousarray(world_to_model, np.float32)
    vk, vv = _vars(vars)
    r = lib().orc_render3d(shape._h, _p(w2m), width, height, depth, _p(ts), len(ts), mode, threads, _p(out),
                           _p(stats), C.byref(secs), _p(vk), _p(vv), len(vk))
    if r != 0:
        raise ValueError("MissingVar")
    return out, dict(zip(stats_names(), (int(s) for s in stats))), secs.value


def pixel_inside(img):
    """RawDistancePixel::inside (pixel.rs:187-193) over a float32 image."""
    bits = img.view(np.uint32)
    is_nan = np.isnan(img)
    key_mask, key = 0xFF << 9, 0xF6 << 9
    is_fill = is_nan & ((bits & key_mask) == key)
    return np.where(is_fill, (bits & 1) == 1, img < 0.0)


def pixel_fill_depth(img):
    """Fill depth (recursion level) or -1 for distance samples."""
    bits = img.view(np.uint32)
    is_nan = np.isnan(img)
    key_mask, key = 0xFF << 9, 0xF6 << 9
    is_fill = is_nan & ((bits & key_mask) == key)
    return np.where(is_fill, ((bits >> 1) & 0xFF).astype(np.int32), -1)


def max_threads():
    return lib().orc_max_threads()


# ---- fidget_raster::effects (effects.rs) -------------------------------------------------------
def denoise_normals(image):
    """effects.rs:17-36: image = GEOMETRY_PIXEL array [h, w]"""
    image = np.ascontiguousarray(image, GEOMETRY_PIXEL)
    out = np.zeros_like(image)
    lib().orc_fx_denoise_normals(_p(image), image.shape[1], image.shape[0], _p(out))
    return out


def compute_ssao(image, depth, kernel, noise):
    """effects.rs:73-95 with the kernel (3 x n) and noise (2 x m) matrices given (the reference draws them from rand::rng())"""
    image = np.ascontiguousarray(image, GEOMETRY_PIXEL)
    k = np.ascontiguousarray(np.asarray(kernel, np.float32).T)   # column j = sample j -> [n][3]
    nz = np.ascontiguousarray(np.asarray(noise, np.float32).T)
    out = np.zeros(image.shape, np.float32)
    lib().orc_fx_compute_ssao(_p(image), image.shape[1], image.shape[0], depth, _p(k), len(k), _p(nz), len(nz), _p(out))
    return out


def blur_ssao(ssao):
    ssao = np.ascontiguousarray(ssao, np.float32)
    out = np.zeros_like(ssao)
    lib().orc_fx_blur_ssao(_p(ssao), ssao.shape[1], ssao.shape[0], _p(out))
    return out


def apply_shading(image, depth, ssao=None):
    """effects.rs:42-67 (ssao = blurred occlusion map or None); returns uint8 [h, w, 3]"""
    image = np.ascontiguousarray(image, GEOMETRY_PIXEL)
    s = None if ssao is None else np.ascontiguousarray(ssao, np.float32)
    out = np.zeros(image.shape + (3,), np.uint8)
    lib().orc_fx_apply_shading(_p(image), image.shape[1], image.shape[0], depth, _p(s), _p(out))
    return out


def _rgba(fn, image, *args):
    image = np.ascontiguousarray(image, np.float32)
    out = np.zeros(image.shape + (4,), np.uint8)
    fn(_p(image), image.size, *args, _p(out))
    return out


def to_rgba_bitmap(image, transparent=False):
    return _rgba(lib().orc_fx_to_rgba_bitmap, image, int(transparent))


def to_debug_bitmap(image):
    return _rgba(lib().orc_fx_to_debug_bitmap, image)


def to_rgba_distance(image):
    return _rgba(lib().orc_fx_to_rgba_distance, image)


MATH_OPS = ["sin", "cos", "tan", "asin", "acos", "atan", "exp", "ln", "atan2"]


def math_unary(op, first, stride, count):
    """glibc f32 libm over the floats with bit patterns first + i * stride (the reference's transcendental opcodes)"""
    out = np.zeros(count, np.float32)
    lib().orc_math_unary(MATH_OPS.index(op), first & 0xFFFFFFFF, stride, count, _p(out))
    return out


# ---- fidget-mesh ------------------------------------------------------------------------------------------------
CELL_KINDS = ["Invalid", "Empty", "Full", "Branch", "Leaf"]


class Octree:
    """fidget_mesh::Octree::build (octree.rs:48-68): the single-threaded path, or with `threads` the multithreaded constructor
    (Settings::threads, build_inner_mt octree.rs:94-210: sub-cells split off breadth-first until there are 10 x threads of them, one
    octree each, spliced and fixed up - another cell / vertex layout, the same tree and the same walk_dual).  Attributes: root (kind,
    mask, index), cells [n, 8, 3] (kind, mask, index), verts [n, 3]; samples (keep_samples): the leaf sampling data (bounds, mask,
    intersections as u16 positions and f32 points, gradients (dx, dy, dz, v), cell vertices) in evaluation order."""

    def __init__(self, shape, depth, world_to_model=None, mode=SIMPLIFY_REFERENCE, vars=None, threads=None, keep_samples=True):
        w2m = None if world_to_model is None else np.ascontiguousarray(world_to_model, np.float32)
        vk, vv = _vars(vars)
        if threads:
            self._h = lib().orc_mesh_build_mt(shape._h, _p(w2m), depth, mode, _p(vk), _p(vv), len(vk), int(threads), 1 if keep_samples else 0)
        else:
            self._h = lib().orc_mesh_build_mt(shape._h, _p(w2m), depth, mode, _p(vk), _p(vv), len(vk), 0, 1 if keep_samples else 0)
        if not self._h:
            raise ValueError("MissingVar")
        c = np.zeros(8, np.uint64)
        lib().orc_mesh_counts(self._h, _p(c))
        nc, nv, ns = int(c[0]), int(c[1]), int(c[2])
        self.interval_evals = int(c[3])
        self.root = (CELL_KINDS[int(c[4])], int(c[5]), int(c[6]))
        self.verts = np.zeros((nv, 3), np.float32)
        lib().orc_mesh_verts(self._h, _p(self.verts))
        self.cells = np.zeros((nc, 8, 3), np.uint32)
        lib().orc_mesh_cells(self._h, _p(self.cells))
        self.samples = {"bounds": np.zeros((ns, 6), np.float32), "info": np.zeros((ns, 3), np.uint32), "inter": np.zeros((ns, 12, 3), np.uint16),
                        "pos": np.zeros((ns, 12, 3), np.float32), "grad": np.zeros((ns, 12, 4), np.float32), "vert": np.zeros((ns, 4, 3), np.float32)}
        s = self.samples
        lib().orc_mesh_samples(self._h, _p(s["bounds"]), _p(s["info"]), _p(s["inter"]), _p(s["pos"]), _p(s["grad"]), _p(s["vert"]))

    def __del__(self):
        try:
            lib().orc_mesh_free(self._h)
        except Exception:
            pass

    def walk_dual(self):
        """Octree::walk_dual (dc.rs): (triangles [n, 3] vertex indices, vertices [m, 3])"""
        c = np.zeros(2, np.uint64)
        lib().orc_mesh_walk_dual(self._h, _p(c))
        tris = np.zeros((int(c[0]), 3), np.uint64)
        verts = np.zeros((int(c[1]), 3), np.float32)
        lib().orc_mesh_dual_copy(self._h, _p(tris), _p(verts))
        return tris, verts


def mdc_table(mask):
    """(CELL_TO_VERT_TO_EDGES[mask] as a list of vertices = lists of (start, end), CELL_TO_EDGE_TO_VERT[mask] as [12] of (vert, edge) or None)"""
    v = np.zeros(128, np.int32)
    e = np.zeros((12, 2), np.int32)
    lib().orc_mesh_table(mask, _p(v), _p(e))
    out, k = [], 1
    for _ in range(v[0]):
        n = v[k]; k += 1
        out.append([(int(v[k + 2 * i]), int(v[k + 2 * i + 1])) for i in range(n)])
        k += 2 * n
    return out, [None if a < 0 else (int(a), int(b)) for a, b in e]


def qef_solve(points, grads):
    p = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
    g = np.ascontiguousarray(grads, np.float32).reshape(-1, 4)
    pos = np.zeros(3, np.float32)
    err = np.zeros(1, np.float32)
    lib().orc_qef_solve(_p(p), _p(g), len(p), _p(pos), _p(err))
    return pos, float(err[0])
