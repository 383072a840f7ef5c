"""fidget_amd — Python binding of libfidget_hip.so (the MI355X / gfx950 backend).

Host-side mirror of the reference's user-facing surface for the evaluation hot path:

    reference (Rust)                                this module
    ---------------------------------------------   ------------------------------
    fidget_core::Context (+ from_text)              Context
    Shape<F> / F: Function + MathFunction           Shape (one device tape; simplify())
    TracingEvaluator / BulkEvaluator impls          Shape.eval_interval / eval_point /
                                                    eval_float_slice / eval_grad_slice
    fidget_raster::pixel::render / voxel::render    render2d / render3d

Everything that computes goes through the C ABI in include/fidget_hip.h (ctypes);
there is NO CPU fallback: importing works without a GPU (so the build and symbol
checks run anywhere), but creating a context without a HIP device raises.
"""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("FHIP_LIB") or os.path.join(_CSRC, "libfidget_hip.so")     # (FHIP_LIB: a variant build, tools/build_lib_variant.py - A/B runs)
_SOURCES = ["capi.hip", "capi_core.hpp", "capi_context.hpp", "capi_tapes.hpp", "capi_eval.hpp", "capi_render.hpp", "capi_effects.hpp", "capi_mesh.hpp", "capi_debug.hpp", "kernels.hip", "prune2.hip", "effects.hip", "mesh.hip", "mesh_qef.hpp", "mesh_collapse.hpp", "mesh_edges.hpp", "mesh_walk.hpp", "host_mesh.hpp", "dev_ops.hpp", "host_graph.hpp", "host_regtape.hpp", "render_state.h", "tape_format.h",
            "gen_interp.py", "gen_tiles.py", "gen_tilesv.py", "gen_normals.py", "gen_prune.py", "gen_ubench.py", "gen_trans.py", "trans_funcs.hip", "trans_libm.hpp", "offsets.cpp", "../../include/fidget_hip.h",
            "../../include/fidget_hip_debug.h"]

UNARY = ["neg", "abs", "recip", "sqrt", "square", "floor", "ceil", "round", "sin", "cos", "tan",
         "asin", "acos", "atan", "exp", "ln", "not", "rand"]          # context/op.rs:11-30
BINARY = ["add", "sub", "mul", "div", "atan2", "min", "max", "compare", "mod", "and", "or", "mix"]  # op.rs:35-48

FH_OPS = (["Output", "Input", "CopyReg", "CopyImm"] + [u.capitalize() for u in UNARY]
          + [b.capitalize() + "RR" for b in ["add", "sub", "mul", "div", "atan2", "compare", "mix", "mod", "min", "max", "and", "or"]]
          + [b.capitalize() + "RI" for b in ["add", "sub", "mul", "div", "atan2", "compare", "mix", "mod", "min", "max", "and", "or"]]
          + [b.capitalize() + "IR" for b in ["sub", "div", "atan2", "compare", "mix", "mod"]])

STATUS = {1: "BadVarSlice", 2: "MismatchedSlices", 3: "BadChoiceSlice", 4: "MissingVar", 5: "BadTape",
          6: "Unsupported", 7: "HipError", 8: "Cancelled", 9: "ParseError", 10: "Overflow"}

VM_TILES_3D = [128, 64, 32, 16, 8]
VM_TILES_2D = [128, 32, 8]
GEOMETRY_PIXEL = np.dtype([("normal", np.float32, 3), ("depth", np.uint32)])
# RenderHints of the HIP shape (capi.hip): 2D 128 / 16 as fidget-jit, 3D 128 / 32 / 8; the VM shape's for comparison
HIP_TILES_2D, HIP_TILES_3D = [128, 16], [128, 32, 8]
VM_TILES_2D, VM_TILES_3D = [128, 32, 8], [128, 64, 32, 16, 8]

EXPORTS = [
    "fhip_ctx_create", "fhip_ctx_destroy", "fhip_ctx_trim", "fhip_ctx_reserve_arena", "fhip_libm_probe", "fhip_last_error", "fhip_ctx_sync", "fhip_cancel", "fhip_cancel_reset", "fhip_cancel_watch", "fhip_ctx_set_option", "fhip_ctx_get_option",
    "fhip_tape_from_bytecode", "fhip_tape_free", "fhip_tape_len", "fhip_tape_reg_tape", "fhip_tape_choice_count", "fhip_tape_reg_count",
    "fhip_tape_var_count", "fhip_tape_output_count", "fhip_tape_ops", "fhip_simplify", "fhip_interval_eval",
    "fhip_point_eval", "fhip_float_eval", "fhip_grad_eval", "fhip_render2d", "fhip_render3d", "fhip_render3d_shard", "fhip_render3d_block", "fhip_merge_depth", "fhip_denoise_normals", "fhip_compute_ssao", "fhip_blur_ssao", "fhip_apply_shading", "fhip_to_rgba", "fhip_mesh_sample", "fhip_mesh_build", "fhip_mesh_vertices", "fhip_mesh_triangles", "fhip_mesh_vertices_ptr", "fhip_mesh_triangles_ptr", "fhip_mesh_free", "fhip_mesh_counts", "fhip_mesh_leaves", "fhip_mesh_sample_part", "fhip_mesh_part_bytes", "fhip_mesh_part_export", "fhip_mesh_merge",
    "fhip_profile_enable", "fhip_profile_read", "fhip_profile_read_kernels", "fhip_render_counters", "fhip_graph_new", "fhip_graph_free",
    "fhip_graph_len", "fhip_graph_var", "fhip_graph_constant", "fhip_graph_unary", "fhip_graph_binary",
    "fhip_graph_from_text", "fhip_tape_from_graph", "fhip_tape_axis_slot", "fhip_tape_var_slot",
    "fhip_screen_to_world", "fhip_debug_groups", "fhip_debug_ubench", "fhip_debug_math_sweep", "fhip_debug_stats", "fhip_debug_leaf_stats", "fhip_debug_tape_links", "fhip_debug_tape_chain", "fhip_debug_bench", "fhip_debug_leaves", "fhip_debug_arena", "fhip_debug_probe", "fhip_debug_trans_probe", "fhip_debug_lane_frames", "fhip_debug_rare_frames", "fhip_debug_lane_tune", "fhip_debug_walk_dual", "fhip_tape_group_count", "fhip_tape_group_op",
    "fhip_tape_group", "fhip_tape_term_plan", "fhip_tape_term_group", "fhip_tape_term_tree", "fhip_tape_term_choice_src",
]


class FidgetHipError(RuntimeError):
    def __init__(self, status, msg=""):
        self.status = status
        super().__init__(f"{STATUS.get(status, status)}: {msg}")


def build(force=False, verbose=False):
    """Compile libfidget_hip.so for gfx950 (cross-compiles without a GPU): the assembly
    interpreters (gen_interp.py -> .s -> code object, embedded in the library) and the HIP
    kernels + C ABI (capi.hip)."""
    srcs = [os.path.join(_CSRC, s) for s in _SOURCES]
    if os.environ.get("FHIP_LIB"):
        return LIB_PATH         # (a variant built elsewhere: never rebuilt from here)
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(s) <= os.path.getmtime(LIB_PATH) for s in srcs):
        return LIB_PATH
    gen = os.path.join(_CSRC, "_gen")
    os.makedirs(gen, exist_ok=True)
    llvm = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin")
    co = os.path.join(gen, "interp_gfx950.co")

    def run(cmd, **kw):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, **kw)

    run(["g++", "-std=c++17", "-I", _CSRC, os.path.join(_CSRC, "offsets.cpp"), "-o", os.path.join(gen, "offsets")])
    with open(os.path.join(gen, "offsets.json"), "w") as f:
        subprocess.check_call([os.path.join(gen, "offsets")], stdout=f)
    # the transcendental routines the assembly interpreters call: trans_funcs.hip through LLVM IR, where the functions are limited to
    # 8 scalar registers + vcc / the return address ("amdgpu-num-sgpr": clang takes that attribute on kernels only; the handlers
    # that call them have ten free) -> llc -> assembly, embedded by gen_trans.py
    ll = os.path.join(gen, "trans_funcs.ll")
    run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "-emit-llvm", "--cuda-device-only", "-I", _CSRC,
         "-o", ll, os.path.join(_CSRC, "trans_funcs.hip")])
    ir = open(ll).read()
    groups = set()
    for fn in ("fh_t_sin", "fh_t_atan2", "fh_t_mod", "fh_t_sin4", "fh_t_ln4"):
        m = re.search(rf"^define [^\n]*@{fn}\([^\n]*\) (#\d+)", ir, re.M)
        assert m, f"trans_funcs.ll: {fn} not found"
        groups.add(m.group(1))
    for g in groups:
        ir, n = re.subn(rf"^attributes {g} = {{ ", f'attributes {g} = {{ "amdgpu-num-sgpr"="18" ', ir, flags=re.M)
        assert n == 1
    with open(ll, "w") as f:
        f.write(ir)
    run([os.path.join(llvm, "llc"), "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-O3", "-o", os.path.join(gen, "trans_funcs.s"), ll])
    run([sys.executable, os.path.join(_CSRC, "gen_interp.py"), os.path.join(gen, "offsets.json"),
         os.path.join(gen, "interp_gfx950.s"), os.path.join(gen, "trans_funcs.s")])
    run([os.path.join(llvm, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c",
         os.path.join(gen, "interp_gfx950.s"), "-o", os.path.join(gen, "interp_gfx950.o")])
    run([os.path.join(llvm, "ld.lld"), "-shared", os.path.join(gen, "interp_gfx950.o"), "-o", co])
    run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         f'-DFH_INTERP_CO="{co}"', "-o", LIB_PATH, os.path.join(_CSRC, "capi.hip")])
    return LIB_PATH


class _Cfg2D(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("world_to_model", C.c_void_p), ("z", C.c_float),
                ("pixel_perfect", C.c_int), ("tile_sizes", C.c_void_p), ("n_tile_sizes", C.c_uint32),
                ("var_keys", C.c_void_p), ("var_values", C.c_void_p), ("n_vars", C.c_uint32),
                ("axis_slots", C.c_void_p)]


class _Cfg3D(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("depth", C.c_uint32), ("world_to_model", C.c_void_p),
                ("tile_sizes", C.c_void_p), ("n_tile_sizes", C.c_uint32), ("var_keys", C.c_void_p),
                ("var_values", C.c_void_p), ("n_vars", C.c_uint32), ("axis_slots", C.c_void_p)]


_lib = None


def lib():
    """Load the HIP extension; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(LIB_PATH)
        vp, u32, i32, f32, u64 = C.c_void_p, C.c_uint32, C.c_int, C.c_float, C.c_uint64
        sig = {
            "fhip_ctx_create": (i32, [i32, vp, C.POINTER(vp)]), "fhip_ctx_destroy": (None, [vp]),
            "fhip_libm_probe": (i32, [C.c_char_p, C.c_size_t]), "fhip_ctx_trim": (i32, [vp]), "fhip_ctx_reserve_arena": (i32, [vp, C.c_size_t]),
            "fhip_last_error": (C.c_char_p, [vp]), "fhip_ctx_sync": (i32, [vp]),
            "fhip_cancel": (None, [vp]), "fhip_cancel_reset": (None, [vp]), "fhip_cancel_watch": (None, [vp, vp]),
            "fhip_ctx_set_option": (i32, [vp, C.c_char_p, i32]), "fhip_ctx_get_option": (i32, [vp, C.c_char_p, C.POINTER(i32)]),
            "fhip_tape_from_bytecode": (i32, [vp, vp, C.c_size_t, C.POINTER(vp)]), "fhip_tape_free": (None, [vp]),
            "fhip_tape_len": (u32, [vp]), "fhip_tape_reg_tape": (i32, [vp, u32, vp, u32, vp, u32, vp]), "fhip_tape_choice_count": (u32, [vp]), "fhip_tape_reg_count": (u32, [vp]),
            "fhip_tape_var_count": (u32, [vp]), "fhip_tape_output_count": (u32, [vp]),
            "fhip_tape_ops": (u32, [vp, vp, u32]),
            "fhip_simplify": (i32, [vp, vp, vp, u32, C.POINTER(vp)]),
            "fhip_interval_eval": (i32, [vp, vp, vp, u32, u32, vp, vp, vp]),
            "fhip_point_eval": (i32, [vp, vp, vp, u32, u32, vp, vp, vp]),
            "fhip_float_eval": (i32, [vp, vp, vp, vp, u32, vp]),
            "fhip_grad_eval": (i32, [vp, vp, vp, vp, u32, vp]),
            "fhip_render2d": (i32, [vp, vp, C.POINTER(_Cfg2D), vp, i32]),
            "fhip_render3d": (i32, [vp, vp, C.POINTER(_Cfg3D), vp, i32]),
            "fhip_render3d_shard": (i32, [vp, vp, C.POINTER(_Cfg3D), vp, i32, u32, u32]),
            "fhip_render3d_block": (i32, [vp, vp, C.POINTER(_Cfg3D), vp, i32, u32, vp]),
            "fhip_merge_depth": (i32, [vp, vp, vp, u64, u32]),
            "fhip_denoise_normals": (i32, [vp, vp, u32, u32, vp, i32]),
            "fhip_compute_ssao": (i32, [vp, vp, u32, u32, u32, vp, u32, vp, u32, vp, i32]),
            "fhip_blur_ssao": (i32, [vp, vp, u32, u32, vp, i32]),
            "fhip_apply_shading": (i32, [vp, vp, u32, u32, u32, vp, vp, i32]),
            "fhip_to_rgba": (i32, [vp, vp, u32, u32, i32, vp, i32]),
            "fhip_mesh_sample": (i32, [vp, vp, u32, vp, vp, vp, vp, u32, C.POINTER(vp)]), "fhip_mesh_free": (None, [vp]),
            "fhip_mesh_counts": (None, [vp, vp]), "fhip_mesh_leaves": (None, [vp, vp]),
            "fhip_mesh_build": (i32, [vp, vp, u32, vp, vp, vp, vp, u32, C.POINTER(vp)]),
            "fhip_mesh_vertices": (None, [vp, vp]), "fhip_mesh_triangles": (None, [vp, vp]),
            "fhip_mesh_vertices_ptr": (vp, [vp]), "fhip_mesh_triangles_ptr": (vp, [vp]),
            "fhip_mesh_sample_part": (i32, [vp, vp, u32, vp, vp, vp, vp, u32, u32, u32, C.POINTER(vp)]),
            "fhip_mesh_part_bytes": (C.c_uint64, [vp]), "fhip_mesh_part_export": (None, [vp, vp]),
            "fhip_mesh_merge": (i32, [vp, vp, vp, u32, vp, C.POINTER(vp)]),
            "fhip_profile_enable": (None, [vp, i32]), "fhip_profile_read": (i32, [vp, vp, vp]), "fhip_profile_read_kernels": (i32, [vp, vp, vp]),
            "fhip_render_counters": (i32, [vp, vp]),
            "fhip_debug_stats": (i32, [vp, vp]), "fhip_debug_leaf_stats": (i32, [vp, vp]), "fhip_debug_tape_links": (u32, [vp, vp, u32]), "fhip_debug_tape_chain": (u32, [vp, vp, u32]),
            "fhip_tape_group_count": (u32, [vp]), "fhip_tape_group_op": (i32, [vp]),
            "fhip_tape_group": (i32, [vp, vp, u32, vp]), "fhip_tape_term_plan": (u32, [vp, vp]), "fhip_tape_term_group": (i32, [vp, vp, u32, vp]),
            "fhip_tape_term_tree": (u32, [vp, vp, u32]), "fhip_tape_term_choice_src": (u32, [vp, vp, u32]),
            "fhip_debug_leaves": (u32, [vp, vp, u32]), "fhip_debug_arena": (u32, [vp, u32, u32, vp]), "fhip_debug_probe": (i32, [vp, vp]),
            "fhip_debug_bench": (i32, [vp, vp, u32, u32, i32, vp]),
            "fhip_debug_walk_dual": (None, [vp, u64, vp, vp, u64, i32, vp, vp, vp]),
            "fhip_debug_groups": (u32, [vp, i32, u32, vp, u32, vp]),
            "fhip_debug_ubench": (i32, [vp, u32, u32, u32, vp]),
            "fhip_debug_math_sweep": (i32, [vp, i32, u32, u32, u64, vp, vp]),
            "fhip_debug_trans_probe": (i32, [vp, u32, u32, u32, u64, vp]),
            "fhip_debug_lane_frames": (u64, [vp]),
            "fhip_debug_rare_frames": (u64, [vp]),
            "fhip_debug_lane_tune": (i32, [vp, vp, vp]),
            "fhip_graph_new": (vp, []), "fhip_graph_free": (None, [vp]), "fhip_graph_len": (u32, [vp]),
            "fhip_graph_var": (u32, [vp, i32, u64]), "fhip_graph_constant": (u32, [vp, f32]),
            "fhip_graph_unary": (u32, [vp, i32, u32]), "fhip_graph_binary": (u32, [vp, i32, u32, u32]),
            "fhip_graph_from_text": (u32, [vp, C.c_char_p]),
            "fhip_tape_from_graph": (i32, [vp, vp, vp, u32, C.POINTER(vp)]),
            "fhip_tape_axis_slot": (i32, [vp, i32]), "fhip_tape_var_slot": (i32, [vp, u64]),
            "fhip_screen_to_world": (None, [vp, i32, vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class HipContext:
    """fhip_ctx: one device + stream.  `stream` may be a raw hipStream_t (e.g. torch's)."""

    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        st = lib().fhip_ctx_create(device, C.c_void_p(stream or 0), C.byref(h))
        if st != 0:
            raise FidgetHipError(st, "no usable HIP device: fidget_amd has no CPU fallback")
        self._h = h

    def __del__(self):
        try:
            lib().fhip_ctx_destroy(self._h)
        except Exception:
            pass

    def check(self, st):
        if st != 0:
            raise FidgetHipError(st, (lib().fhip_last_error(self._h) or b"").decode())

    def sync(self):
        self.check(lib().fhip_ctx_sync(self._h))

    def trim(self):
        """fhip_ctx_trim: memory kept between calls for speed alone (mesh leaf records, frame lanes) goes back to the device"""
        self.check(lib().fhip_ctx_trim(self._h))

    def reserve_arena(self, megabytes):
        """fhip_ctx_reserve_arena: every buffer set's tape arena at least this large from the next frame on"""
        self.check(lib().fhip_ctx_reserve_arena(self._h, int(megabytes)))

    def cancel(self):
        lib().fhip_cancel(self._h)

    def cancel_watch(self, flag):
        """fhip_cancel_watch: `flag` = a numpy uint8 array of one element the context reads beside its own flag (None: stop watching)"""
        self._watched = flag          # (kept alive while it is watched)
        lib().fhip_cancel_watch(self._h, None if flag is None else flag.ctypes.data_as(C.c_void_p))

    def cancel_reset(self):
        lib().fhip_cancel_reset(self._h)

    def lane_tune(self):
        """What the frame arrangement tuner knows about the kind of 3D frame queued last (fhip_debug_lane_tune): phase 0 .. 2 measuring
        (stage pipeline, lanes, stage pipeline again), 3 waiting, 4 decided, -1 none; ms per frame of the three windows; the decision."""
        ms, ln = (C.c_float * 3)(), C.c_int(0)
        ph = lib().fhip_debug_lane_tune(self._h, ms, C.byref(ln))
        return {"phase": int(ph), "stage_pipeline_ms": [float(ms[0]), float(ms[2])], "frame_lanes_ms": float(ms[1]), "kept": "frame lanes" if ln.value else "stage pipeline"}

    def lane_frames(self):
        """Frames of this context that went to a frame lane so far (fhip_debug_lane_frames)"""
        return int(lib().fhip_debug_lane_frames(self._h))

    def rare_frames(self):
        """3D frames of this context rendered in rare mode so far - the launches for tapes beyond the assembly kernels' register files folded
        into the slab's other launches (fhip_debug_rare_frames; capi_render.hpp)"""
        return int(lib().fhip_debug_rare_frames(self._h))

    def set_option(self, name, value=1):
        """A behaviour switch of this context (fhip_ctx_set_option: "no_column_inv", "frame_lanes", ...).  The environment
        (FHIP_<NAME>) is read once, when the context is created; this is the only way to change a switch afterwards."""
        self.check(lib().fhip_ctx_set_option(self._h, name.encode(), int(value)))

    def option(self, name):
        v = C.c_int(0)
        self.check(lib().fhip_ctx_get_option(self._h, name.encode(), C.byref(v)))
        return v.value

    def options(self, **kw):
        """Context manager: switches set for the duration of a block, then restored."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            old = {k: self.option(k) for k in kw}
            try:
                for k, v in kw.items():
                    self.set_option(k, v)
                yield self
            finally:
                for k, v in old.items():
                    self.set_option(k, v)
        return scope()

    def profile(self, on):
        lib().fhip_profile_enable(self._h, int(on))

    def profile_read(self):
        ms = np.zeros(4, np.float64)
        n = np.zeros(4, np.uint32)
        self.check(lib().fhip_profile_read(self._h, _p(ms), _p(n)))
        names = ["tiles", "points", "normals", "other"]
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(names)}

    def profile_read_kernels(self):
        """(ms, launches) of the last profiled frame per assembly kernel, every launch timed on its own."""
        ms = np.zeros(8, np.float64)
        n = np.zeros(8, np.uint32)
        self.check(lib().fhip_profile_read_kernels(self._h, _p(ms), _p(n)))
        # (fh_prune1: the root level's prune, i.e. k_prune2 with the scalar sweep behind it when option prune2 is on)
        names = ["fh_columns", "fh_float_eval_16x4", "fh_float_eval_32x2", "fh_tiles", "fh_prune1", "fh_tiles_v32", "fh_tiles_v64", "unused"]
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(names)}

    def counters(self):
        c = np.zeros(8, np.uint64)
        self.check(lib().fhip_render_counters(self._h, _p(c)))
        return {"arena_ops": int(c[0]), "arena_overflow": int(c[1]), "leaves_last_slab": int(c[2]),
                "queue_overflow": int(c[3]), "groups_last_slab": [int(v) for v in c[4:6]], "hip_tile_stage_frames": int(c[6]), "substituted_tile_lists": int(c[7])}

    def last_leaves(self, cap=1 << 20):
        """Leaf records of the last slab of the last 3D frame: structured array (off, len, regs, choices, x, y, z)."""
        dt = np.dtype([("off", np.uint32), ("len", np.uint32), ("regs", np.uint16), ("choices", np.uint16),
                       ("x", np.uint32), ("y", np.uint32), ("z", np.uint32)])
        buf = np.zeros(cap, dt)
        n = lib().fhip_debug_leaves(self._h, _p(buf), cap)
        return buf[:n]

    def groups(self, kind, index, cap=1 << 20):
        """Work-queue entries the last 3D frame left behind (kind 0: tile level `index`; 1: parked z-slab `index`):
        (structured array (off, len, regs, choices, x, y, z, ...), (n_small_layout, n_other))."""
        dt = np.dtype([("off", np.uint32), ("len", np.uint32), ("regs", np.uint16), ("choices", np.uint16),
                       ("x", np.uint32), ("y", np.uint32), ("z", np.uint32), ("first", np.uint32), ("n", np.uint32), ("stride", np.uint32)])
        assert dt.itemsize == 36  # sizeof(FhGroup)
        buf = np.zeros(cap, dt)
        cnt = np.zeros(2, np.uint32)
        n = lib().fhip_debug_groups(self._h, kind, index, _p(buf), cap, _p(cnt))
        return buf[:n], (int(cnt[0]), int(cnt[1]))

    def arena_ops(self, off, n):
        """`n` device-format ops of the tape arena from op `off` (diagnostics; see last_leaves)."""
        buf = np.zeros(n, np.uint64)
        got = lib().fhip_debug_arena(self._h, int(off), int(n), _p(buf))
        return buf[:got]

    def leaf_stats(self):
        """Leaf-stage counters of the last profiled 3D frame (render_state.h leaf_stat)."""
        c = np.zeros(8, np.uint64)
        self.check(lib().fhip_debug_leaf_stats(self._h, _p(c)))
        return {"leaves": int(c[0]), "tape_ops": int(c[1]), "tape_words_read": int(c[2]), "lane_ops": int(c[3]),
                "prune2_phase_clocks_max": [int(v) for v in c[4:8]]}

    def wave_stats(self):
        """Per kernel kind: mean / max busy microseconds of the waves that found work, their
        number and the units of work they pulled (frame totals)."""
        c = np.zeros(64, np.uint64)
        self.check(lib().fhip_debug_stats(self._h, _p(c)))
        names = ["tiles_l0", "tiles_l1", "tiles_l2", "tiles_l3", "tiles_l4", "columns_c0", "columns_c12", "tiles_2d"]
        self.tile_v = {f"l{l}": {"slots": int(c[8 + l]), "fwd_clocks": int(c[16 + l]), "prune_clocks": int(c[24 + l]), "max_slot_clocks": int(c[l])}
                       for l in range(8) if c[8 + l]}
        self.tile_phases = {f"l{l}": {"fwd_us": int(c[32 + l]) / 100.0, "prune_us": int(c[40 + l]) / 100.0,
                                      "ops": int(c[48 + l]), "ops_written": int(c[56 + l]), "fwd_shader_clocks": int(c[16 + l])} for l in range(8) if c[48 + l]}
        return {k: {"busy_us_sum": int(c[4 * i]) / 100.0, "busy_us_max": int(c[4 * i + 1]) / 100.0,
                    "waves": int(c[4 * i + 2]), "units": int(c[4 * i + 3])} for i, k in enumerate(names) if c[4 * i + 2]}


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = HipContext(int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("FHIP_USE_LOCAL_RANK") else 0)
    return _default_ctx


BAD_NODE = 0xFFFFFFFF


class BadNode(Exception):
    pass


class Node(int):
    pass


class Context:
    """Host mirror of fidget_core::Context (constructor identities, dedup, .vm text)."""

    def __init__(self):
        self._h = C.c_void_p(lib().fhip_graph_new())

    def __del__(self):
        try:
            lib().fhip_graph_free(self._h)
        except Exception:
            pass

    def __len__(self):
        return lib().fhip_graph_len(self._h)

    def _node(self, v):
        if isinstance(v, (float, int)) and not isinstance(v, Node):
            return self.constant(float(v))
        return v

    def x(self): return Node(lib().fhip_graph_var(self._h, 0, 0))
    def y(self): return Node(lib().fhip_graph_var(self._h, 1, 0))
    def z(self): return Node(lib().fhip_graph_var(self._h, 2, 0))
    def var(self, index): return Node(lib().fhip_graph_var(self._h, 3, int(index)))
    def constant(self, f): return Node(lib().fhip_graph_constant(self._h, float(f)))

    def _un(self, name, a):
        r = lib().fhip_graph_unary(self._h, UNARY.index(name), int(self._node(a)))
        if r == BAD_NODE:
            raise BadNode()
        return Node(r)

    def _bin(self, name, a, b):
        r = lib().fhip_graph_binary(self._h, BINARY.index(name), int(self._node(a)), int(self._node(b)))
        if r == BAD_NODE:
            raise BadNode()
        return Node(r)

    # derived constructors (context/mod.rs:701-780)
    def less_than(self, lhs, rhs):
        return self.max(self._bin("compare", rhs, lhs), 0.0)

    def less_than_or_equal(self, lhs, rhs):
        return self.min(self.add(self._bin("compare", rhs, lhs), 1.0), 1.0)

    def if_nonzero_else(self, cond, a, b):
        cond, a, b = self._node(cond), self._node(a), self._node(b)
        lhs = self.and_(cond, a)
        rhs = self.and_(self.not_(cond), b)
        return self.or_(lhs, rhs)

    @staticmethod
    def from_text(text):
        ctx = Context()
        if isinstance(text, str):
            text = text.encode()
        r = lib().fhip_graph_from_text(ctx._h, text)
        if r == BAD_NODE:
            raise ValueError("parse error")
        return ctx, Node(r)


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
    """One function compiled to a device tape (the role of `Shape<HipFunction>`)."""

    def __init__(self, ctx=None, node=None, n_regs=255, _h=None, roots=None, hip=None, _vars=None):
        # n_regs is accepted for signature parity with GenericVmFunction<N>; device tapes
        # always use dense renumbering over up to 256 registers (no spills).
        self._hip = hip  # created lazily: tape construction and simplify are host-only
        self.n_regs = n_regs
        if _h is not None:
            self._h = _h
            self._vars = _vars
            return
        if roots is None:
            roots = [node]
        r = np.array([int(n) for n in roots], dtype=np.uint32)
        h = C.c_void_p()
        st = lib().fhip_tape_from_graph(None, ctx._h, _p(r), len(r), C.byref(h))
        if st != 0:
            raise FidgetHipError(st, "tape construction failed")
        self._h = h
        self._vars = None

    @property
    def hip(self):
        if self._hip is None:
            self._hip = default_context()
        return self._hip

    def __del__(self):
        try:
            lib().fhip_tape_free(self._h)
        except Exception:
            pass

    def groups(self):
        """Tape parallelism: (combining op name, [Shape of each independent sub-tape]); ('', []) if the
        root of the function is not a min / max of many parts."""
        n = lib().fhip_tape_group_count(self._h)
        out = []
        for g in range(n):
            h = C.c_void_p()
            st = lib().fhip_tape_group(None, self._h, g, C.byref(h))
            if st != 0:
                raise FidgetHipError(st, "tape group")
            out.append(Shape(_h=h, hip=self._hip, _vars=self._vars))
        op = lib().fhip_tape_group_op(self._h)
        return ({30: "min", 31: "max"}.get(op, ""), out)

    def term_parts(self):
        """(group Shapes, tree ops as an (n, 3) u32 array, choice sources as a u32 array) of term_plan()."""
        n = self.term_plan()
        gs = []
        for g in range(n["groups"]):
            h = C.c_void_p()
            st = lib().fhip_tape_term_group(None, self._h, g, C.byref(h))
            if st != 0:
                raise FidgetHipError(st, "term group")
            gs.append(Shape(_h=h, hip=self._hip, _vars=self._vars))
        tree = np.zeros((max(n["tree_ops"], 1), 3), np.uint32)
        lib().fhip_tape_term_tree(self._h, _p(tree), n["tree_ops"])
        src = np.zeros(max(n["choices"], 1), np.uint32)
        lib().fhip_tape_term_choice_src(self._h, _p(src), n["choices"])
        return gs, tree[:n["tree_ops"]], src[:n["choices"]]

    def term_plan(self):
        """The renderer's root-level split: dict(groups, terms, tree_ops, tree_regs, choices); groups = 0 if none."""
        info = (C.c_uint32 * 4)()
        n = lib().fhip_tape_term_plan(self._h, info)
        return dict(groups=n, terms=info[0], tree_ops=info[1], tree_regs=info[2], choices=info[3])

    @staticmethod
    def from_vm(path_or_text, n_regs=255, hip=None):
        text = open(path_or_text).read() if os.path.exists(path_or_text) else path_or_text
        ctx, root = Context.from_text(text)
        return Shape(ctx, root, n_regs, hip=hip)

    @staticmethod
    def from_bytecode(words, axis_slots=(0, 1, 2), hip=None):
        """The reference wire format (fidget_bytecode::Bytecode::data()); axis_slots = VarMap slots of X, Y, Z."""
        w = np.ascontiguousarray(words, dtype=np.uint32)
        h = C.c_void_p()
        st = lib().fhip_tape_from_bytecode(None, _p(w), len(w), C.byref(h))
        if st != 0:
            raise FidgetHipError(st, "bad bytecode")
        return Shape(_h=h, hip=hip, _vars=tuple(axis_slots))

    # sizes ---------------------------------------------------------------
    def device_len(self): return lib().fhip_tape_len(self._h)

    def size(self):
        """Function::size (eval/mod.rs:171) = VmData<N>::len(): the device tape's length - one op per SSA op - or, for a Shape made with
        fewer than 255 registers, the reference's RegTape under that limit with its loads and stores (fhip_tape_reg_tape)."""
        if self.n_regs == 255 and lib().fhip_tape_reg_count(self._h) <= 255:
            return self.device_len()
        return int(self.reg_tape()[1][0])     # (also for a tape that needs more than 255 registers: VmData<255>::len() counts its loads / stores)
    __len__ = size
    def ssa_len(self): return self.device_len()  # device tapes carry no load/store: one op per SSA op

    def reg_tape(self, n_regs=None, want_words=False):
        """RegTape::new::<N> + Bytecode::new of this tape (host side): ([(op, form, out, a, b, idx, imm bits)] in evaluation order - the
        oracle's asm_ops() records -, (len, slot_count, reg_count, mem_count)[, bytecode words])."""
        n_regs = self.n_regs if n_regs is None else n_regs
        info = np.zeros(4, np.uint32)
        st = lib().fhip_tape_reg_tape(self._h, n_regs, None, 0, None, 0, _p(info))
        if st not in (0, 6):
            raise FidgetHipError(st, "register allocation")
        n = int(info[0])
        rec = np.zeros((n, 4), np.uint32)
        words = np.zeros(2 * n + 4, np.uint32)
        st = lib().fhip_tape_reg_tape(self._h, n_regs, _p(rec), n, _p(words), len(words), _p(info))
        if want_words and st == 6:
            raise ValueError("ReservedRegister")
        NAMES = ["Output", "Input", "CopyReg", "CopyImm", "Neg", "Abs", "Recip", "Sqrt", "Square", "Floor", "Ceil", "Round", "Sin", "Cos", "Tan",
                 "Asin", "Acos", "Atan", "Exp", "Ln", "Not", "Rand"]
        BIN = ["Add", "Sub", "Mul", "Div", "Atan2", "Compare", "Mix", "Mod", "Min", "Max", "And", "Or"]
        ops = []
        for op, o, a, w in rec.tolist():
            if op == 52: ops.append(("Load", "", o, 0, 0, w, 0))
            elif op == 53: ops.append(("Store", "", 0, a, 0, w, 0))
            elif op == 0: ops.append(("Output", "", 0, a, 0, w, 0))
            elif op == 1: ops.append(("Input", "", o, 0, 0, w, 0))
            elif op == 3: ops.append(("CopyImm", "", o, 0, 0, 0, w))
            elif op < 22: ops.append((NAMES[op], "Reg", o, a, 0, 0, 0))
            elif op < 34: ops.append((BIN[op - 22], "RegReg", o, a, w, 0, 0))
            elif op < 46: ops.append((BIN[op - 34], "RegImm", o, a, 0, 0, w))
            else: ops.append((["Sub", "Div", "Atan2", "Compare", "Mix", "Mod"][op - 46], "ImmReg", o, a, 0, 0, w))
        out = (ops, tuple(int(v) for v in info))
        return out + (words,) if want_words else out

    def asm_ops(self): return self.reg_tape()[0]

    def bytecode(self):
        """fidget_bytecode::Bytecode::new of the tape as VmData<n_regs>: (words, reg_count, mem_count)"""
        _, info, words = self.reg_tape(want_words=True)
        return words, info[2], info[3]
    def choice_count(self): return lib().fhip_tape_choice_count(self._h)
    def output_count(self): return lib().fhip_tape_output_count(self._h)
    def slot_count(self): return lib().fhip_tape_reg_count(self._h)
    def var_count(self): return lib().fhip_tape_var_count(self._h)

    def axis_index(self, axis):
        if self._vars is not None:
            return self._vars[axis]
        return lib().fhip_tape_axis_slot(self._h, axis)

    def var_index(self, index): return lib().fhip_tape_var_slot(self._h, int(index))

    def words(self):
        """Device tape as raw 8-byte ops (tape_format.h), evaluation order."""
        n = self.device_len()
        w = np.zeros(max(n, 1), dtype=np.uint64)
        lib().fhip_tape_ops(self._h, _p(w), n)
        return w[:n]

    def links(self):
        """Per-op links for the linked prune (fhip_debug_tape_links; host_graph.hpp compute_links): [n, 5] = (opcode, class, choice ordinal,
        producer of a, producer of b) - a producer is an op index, 0x8000 | ordinal for a choice op, 0xFFFF for none; None when the tape
        does not qualify."""
        n = self.device_len()
        w = np.zeros(max(n, 1), dtype=np.uint64)
        if lib().fhip_debug_tape_links(self._h, _p(w), n) != n:
            return None
        w = w[:n]
        return np.stack([w & 0xFF, (w >> 8) & 0xFF, (w >> 16) & 0xFFFF, (w >> 32) & 0xFFFF, (w >> 48) & 0xFFFF], axis=1).astype(np.int64)

    def chain(self):
        """The root chain for the linked prune's liveness pass (fhip_debug_tape_chain): [n, 2] = (choice ordinal, op index) per chain op in
        evaluation order; empty when the root tree is no chain"""
        w = np.zeros(65536, dtype=np.uint32)
        n = lib().fhip_debug_tape_chain(self._h, _p(w), len(w))
        return np.stack([w[:n] & 0xFFFF, w[:n] >> 16], axis=1).astype(np.int64)

    def ops(self):
        """Device tape as (name, out, a, b, imm_bits) tuples in evaluation order."""
        n = self.device_len()
        w = np.zeros(max(n, 1), dtype=np.uint64)
        lib().fhip_tape_ops(self._h, _p(w), n)
        out = []
        for v in w[:n]:
            v = int(v)
            out.append((FH_OPS[v & 0xFF], (v >> 8) & 0xFFF, (v >> 20) & 0xFFF, v >> 32, v >> 32))  # (name, out, a, b|imm, imm)
        return out

    def simplify(self, choices, n_regs=None):
        c = np.ascontiguousarray(choices, dtype=np.uint8)
        h = C.c_void_p()
        st = lib().fhip_simplify(None, self._h, _p(c), len(c), C.byref(h))
        if st != 0:
            raise ValueError(STATUS.get(st, str(st)))
        return Shape(_h=h, hip=self._hip, _vars=self._vars if self._vars is not None else tuple(self.axis_index(a) for a in range(3)),
                     n_regs=self.n_regs if n_regs is None else n_regs)._with_named(self)

    def _with_named(self, parent):
        self._named_parent = parent  # children keep the parent's Var::V slots
        return self

    def _named_slot(self, index):
        p = self
        while getattr(p, "_named_parent", None) is not None:
            p = p._named_parent
        return lib().fhip_tape_var_slot(p._h, int(index))

    # evaluators ------------------------------------------------------------
    def _xyz_vars(self, x, y, z, extra=None):
        n = max(self.var_count(), 1)
        vs = [None] * n
        for axis, v in enumerate((x, y, z)):
            i = self.axis_index(axis)
            if 0 <= i < n:
                vs[i] = v
        for k, v in (extra or {}).items():
            i = self._named_slot(k)
            if 0 <= i < n:
                vs[i] = v
        return vs

    def _tracing(self, fn, v, comp):
        n, nv = v.shape[0], v.shape[1]
        no, nc = self.output_count(), self.choice_count()
        out = np.zeros((n, max(no, 1)) + ((2,) if comp == 2 else ()), dtype=np.float32)
        ch = np.zeros((n, max(nc, 1)), dtype=np.uint8)
        simp = np.zeros(n, dtype=np.uint8)
        st = fn(self.hip._h, self._h, _p(v), nv, n, _p(out), _p(ch), _p(simp))
        if st in (1, 2, 3):
            raise ValueError(STATUS[st])
        self.hip.check(st)
        return out[:, :no], ch[:, :nc], simp

    def eval_interval_batch(self, batch):
        """batch: list of per-variable (lo, hi) lists (None = [0,0]).  Returns [((lo,hi), trace|None)]."""
        v = np.array([[(0.0, 0.0) if a is None else a for a in vs] for vs in batch], dtype=np.float32)
        if v.ndim != 3:
            v = v.reshape(len(batch), -1, 2)
        out, ch, simp = self._tracing(lib().fhip_interval_eval, np.ascontiguousarray(v), 2)
        return [((float(out[i, 0, 0]), float(out[i, 0, 1])), (ch[i].copy() if simp[i] else None)) for i in range(len(batch))]

    def eval_interval_raw(self, vars_):
        v = np.array(vars_, dtype=np.float32).reshape(1, -1, 2)
        out, ch, simp = self._tracing(lib().fhip_interval_eval, v, 2)
        return out[0], (ch[0].copy() if simp[0] else None)

    def eval_interval(self, x, y, z, extra=None):
        vs = self._xyz_vars(x, y, z, extra)
        vs = [(0.0, 0.0) if v is None else ((v, v) if np.isscalar(v) else tuple(v)) for v in vs]
        out, tr = self.eval_interval_raw(vs)
        return (float(out[0][0]), float(out[0][1])), tr

    def eval_point_raw(self, vars_):
        v = np.array(vars_, dtype=np.float32).reshape(1, -1)
        out, ch, simp = self._tracing(lib().fhip_point_eval, v, 1)
        return out[0], (ch[0].copy() if simp[0] else None)

    def eval_point(self, x, y, z, extra=None):
        vs = [0.0 if v is None else v for v in self._xyz_vars(x, y, z, extra)]
        out, tr = self.eval_point_raw(vs)
        return float(out[0]), tr

    def _bulk(self, fn, arrs, comp):
        n = len(arrs[0]) // comp if arrs else 0
        lens = np.array([len(a) // comp for a in arrs], dtype=np.uint32)
        ptrs = (C.c_void_p * max(len(arrs), 1))(*[a.ctypes.data for a in arrs])
        no = self.output_count()
        outs = [np.zeros(n * comp, dtype=np.float32) for _ in range(no)]
        optrs = (C.c_void_p * max(no, 1))(*[o.ctypes.data for o in outs])
        st = fn(self.hip._h, self._h, ptrs, _p(lens), len(arrs), optrs)
        if st in (1, 2, 3):
            raise ValueError(STATUS[st])
        self.hip.check(st)
        return outs

    def eval_float_slice_raw(self, arrays):
        arrs = [np.ascontiguousarray(a, dtype=np.float32).reshape(-1) for a in arrays]
        outs = self._bulk(lib().fhip_float_eval, arrs, 1)
        n = len(arrs[0]) if arrs else 0
        return np.array(outs, dtype=np.float32).reshape(self.output_count(), n)

    def eval_float_slice(self, x, y, z, extra=None):
        x, y, z = (np.asarray(a, dtype=np.float32) for a in (x, y, z))
        n = len(x)
        vs = self._xyz_vars(x, y, z, extra)
        vs = [np.zeros(n, np.float32) if v is None else (np.full(n, v, np.float32) if np.isscalar(v) else v) for v in vs]
        return self.eval_float_slice_raw(vs)[0]

    def eval_grad_slice_raw(self, arrays):
        arrs = [np.ascontiguousarray(a, dtype=np.float32).reshape(-1) for a in arrays]
        outs = self._bulk(lib().fhip_grad_eval, arrs, 4)
        n = len(arrs[0]) // 4 if arrs else 0
        return np.array(outs, dtype=np.float32).reshape(self.output_count(), n, 4)

    def eval_grad_slice(self, x, y, z, extra=None):
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


# ---- geometry helpers (host f32, same operation order as the C++ driver) --------------
def screen_to_world(size):
    n = len(size)
    s = np.array(size, dtype=np.uint32)
    out = np.zeros((n + 1, n + 1), dtype=np.float32)
    lib().fhip_screen_to_world(_p(s), n, _p(out))
    return out


def mat_mul(a, b):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    d = a.shape[0]
    out = np.zeros((d, d), np.float32)
    for col in range(d):
        for row in range(d):
            acc = np.float32(a[row, 0] * b[0, col])
            for k in range(1, d):
                acc = np.float32(np.float32(a[row, k] * b[k, col]) + acc)
            out[row, col] = acc
    return out


def lift_2d(m3):
    m3 = np.asarray(m3, np.float32)
    m = np.zeros((4, 4), np.float32)
    m[0, :2] = m3[0, :2]; m[0, 3] = m3[0, 2]
    m[1, :2] = m3[1, :2]; m[1, 3] = m3[1, 2]
    m[2, 2] = 1.0
    m[3, :2] = m3[2, :2]; m[3, 3] = m3[2, 2]
    return m


def transform_point(mat4, x, y, z):
    m = np.asarray(mat4, np.float32)
    x, y, z = np.float32(x), np.float32(y), np.float32(z)
    r = [np.float32(np.float32(np.float32(m[i, 0] * x) + np.float32(m[i, 1] * y)) + np.float32(m[i, 2] * z)) + m[i, 3]
         for i in range(4)]
    r = [np.float32(v) for v in r]
    if r[3] != 0:
        return np.array([r[0] / r[3], r[1] / r[3], r[2] / r[3]], np.float32)
    return np.array(r[:3], np.float32)


def _var_arrays(shape, vars_):
    vars_ = vars_ or {}
    k = np.array(list(vars_.keys()), dtype=np.uint64)
    v = np.array(list(vars_.values()), dtype=np.float32)
    return k, v


def _dev_ptr(out):
    """Accept a torch CUDA tensor (device pointer) or None."""
    if out is None:
        return None
    return C.c_void_p(out.data_ptr())


def render2d(shape, width, height=None, z=0.0, pixel_perfect=False, world_to_model=None, tile_sizes=None,
             vars=None, out=None, mode=None, threads=None):
    """fidget_raster::pixel::render on the GPU.  Returns (float32 [h,w] RawDistancePixel image, stats, seconds);
    with `out` (a torch CUDA float32 tensor) the call is asynchronous and returns (out, None, None)."""
    height = width if height is None else height
    hip = shape.hip
    ts = np.array(tile_sizes, dtype=np.uint32) if tile_sizes else None
    w2m = None if world_to_model is None else np.ascontiguousarray(world_to_model, np.float32)
    vk, vv = _var_arrays(shape, vars)
    ax = None
    if shape._vars is not None:
        ax = np.array(shape._vars, dtype=np.int32)
        vk = np.array([shape._named_slot(k) for k in (vars or {})], dtype=np.uint64)
    cfg = _Cfg2D(width, height, _p(w2m), z, int(pixel_perfect), _p(ts), 0 if ts is None else len(ts), _p(vk), _p(vv),
                 len(vk), _p(ax))
    if out is not None:
        st = lib().fhip_render2d(hip._h, shape._h, C.byref(cfg), _dev_ptr(out), 1)
        if st == 4:
            raise ValueError("MissingVar")
        hip.check(st)
        return out, None, None
    img = np.zeros((height, width), dtype=np.float32)
    import time
    t0 = time.perf_counter()
    st = lib().fhip_render2d(hip._h, shape._h, C.byref(cfg), _p(img), 0)
    dt = time.perf_counter() - t0
    if st == 4:
        raise ValueError("MissingVar")
    hip.check(st)
    return img, {}, dt


def render3d(shape, width, height=None, depth=None, world_to_model=None, tile_sizes=None, vars=None, out=None,
             shard=0, n_shards=1, mode=None, threads=None, block=None, host_out=None):
    """fidget_raster::voxel::render on the GPU.  Returns (GeometryPixel [h,w] image, stats, seconds).
    Multi-GPU parts: (shard, n_shards) = root-tile columns round robin; block = (index, (nx, ny, nz)) = one block of an
    nx x ny x nz split of the volume (fhip_render3d_block)."""
    height = width if height is None else height
    depth = width if depth is None else depth
    hip = shape.hip
    if out is not None and world_to_model is None and not tile_sizes and not vars and shape._vars is None:
        # (the same frame again - what a caller that holds its RenderConfig does: the configuration struct is kept with the shape, the call
        # is the C call and nothing else; queued frames are bound by the thread that queues them)
        key = (width, height, depth, shard, n_shards, None if block is None else (int(block[0]), tuple(int(b) for b in block[1])))
        frames = shape.__dict__.setdefault("_frames", {})
        ent = frames.get(key)
        if ent is None:
            cfg = _Cfg3D(width, height, depth, None, None, 0, None, None, 0, None)
            split = None if block is None else np.array(block[1], dtype=np.uint32)
            ent = frames[key] = (cfg, C.byref(cfg), split, _p(split))
        if block is not None:
            st = lib().fhip_render3d_block(hip._h, shape._h, ent[1], C.c_void_p(out.data_ptr()), 1, key[5][0], ent[3])
        else:
            st = lib().fhip_render3d_shard(hip._h, shape._h, ent[1], C.c_void_p(out.data_ptr()), 1, shard, n_shards)
        if st:
            if st == 4:
                raise ValueError("MissingVar")
            hip.check(st)
        return out, None, None
    ts = np.array(tile_sizes, dtype=np.uint32) if tile_sizes else None
    w2m = None if world_to_model is None else np.ascontiguousarray(world_to_model, np.float32)
    vk, vv = _var_arrays(shape, vars)
    ax = None
    if shape._vars is not None:
        ax = np.array(shape._vars, dtype=np.int32)
        vk = np.array([shape._named_slot(k) for k in (vars or {})], dtype=np.uint64)
    cfg = _Cfg3D(width, height, depth, _p(w2m), _p(ts), 0 if ts is None else len(ts), _p(vk), _p(vv), len(vk), _p(ax))
    def call(ptr, dev):
        if block is not None:
            split = np.array(block[1], dtype=np.uint32)
            return lib().fhip_render3d_block(hip._h, shape._h, C.byref(cfg), ptr, dev, int(block[0]), _p(split))
        return lib().fhip_render3d_shard(hip._h, shape._h, C.byref(cfg), ptr, dev, shard, n_shards)
    if out is not None:
        st = call(_dev_ptr(out), 1)
        if st == 4:
            raise ValueError("MissingVar")
        hip.check(st)
        return out, None, None
    # (host_out: a caller's own [height, width] GEOMETRY_PIXEL array to land in - pinned memory, say)
    img = np.zeros((height, width), dtype=GEOMETRY_PIXEL) if host_out is None else host_out
    assert img.dtype == GEOMETRY_PIXEL and img.shape == (height, width) and img.flags.c_contiguous
    import time
    t0 = time.perf_counter()
    st = call(_p(img), 0)
    dt = time.perf_counter() - t0
    if st == 4:
        raise ValueError("MissingVar")
    hip.check(st)
    return img, {}, dt


def merge_depth(front, back, image_depth, hip=None):
    """fhip_merge_depth on torch CUDA tensors ([h, w, 4] int32 GeometryPixel words), in place on `front`."""
    hip = hip or default_context()
    hip.check(lib().fhip_merge_depth(hip._h, _dev_ptr(front), _dev_ptr(back), front.numel() // 4, int(image_depth)))
    return front


def pixel_inside(img):
    """RawDistancePixel::inside (fidget-raster/src/pixel.rs:187-193)."""
    img = np.asarray(img, np.float32)
    bits = img.view(np.uint32)
    is_fill = np.isnan(img) & ((bits & (0xFF << 9)) == (0xF6 << 9))
    return np.where(is_fill, (bits & 1) == 1, img < 0.0)


def pixel_fill_depth(img):
    img = np.asarray(img, np.float32)
    bits = img.view(np.uint32)
    is_fill = np.isnan(img) & ((bits & (0xFF << 9)) == (0xF6 << 9))
    return np.where(is_fill, ((bits >> 1) & 0xFF).astype(np.int32), -1)


# ---- fidget_raster::effects (fidget-raster/src/effects.rs), host arrays in / out -----------------------
def denoise_normals(image, hip=None):
    hip = hip or default_context()
    image = np.ascontiguousarray(image, GEOMETRY_PIXEL)
    out = np.zeros_like(image)
    hip.check(lib().fhip_denoise_normals(hip._h, _p(image), image.shape[1], image.shape[0], _p(out), 0))
    return out


def compute_ssao(image, depth, kernel, noise, hip=None):
    """kernel: 3 x n, noise: 2 x m (the matrices effects::ssao_kernel / ssao_noise return)"""
    hip = hip or default_context()
    image = np.ascontiguousarray(image, GEOMETRY_PIXEL)
    k = np.ascontiguousarray(np.asarray(kernel, np.float32).T)
    nz = np.ascontiguousarray(np.asarray(noise, np.float32).T)
    out = np.zeros(image.shape, np.float32)
    hip.check(lib().fhip_compute_ssao(hip._h, _p(image), image.shape[1], image.shape[0], depth, _p(k), len(k), _p(nz), len(nz), _p(out), 0))
    return out


def blur_ssao(ssao, hip=None):
    hip = hip or default_context()
    ssao = np.ascontiguousarray(ssao, np.float32)
    out = np.zeros_like(ssao)
    hip.check(lib().fhip_blur_ssao(hip._h, _p(ssao), ssao.shape[1], ssao.shape[0], _p(out), 0))
    return out


def apply_shading(image, depth, ssao=None, hip=None):
    hip = hip or default_context()
    image = np.ascontiguousarray(image, GEOMETRY_PIXEL)
    s = None if ssao is None else np.ascontiguousarray(ssao, np.float32)
    out = np.zeros(image.shape + (3,), np.uint8)
    hip.check(lib().fhip_apply_shading(hip._h, _p(image), image.shape[1], image.shape[0], depth, _p(s), _p(out), 0))
    return out


def _to_rgba(image, mode, hip=None):
    hip = hip or default_context()
    image = np.ascontiguousarray(image, np.float32)
    out = np.zeros(image.shape + (4,), np.uint8)
    hip.check(lib().fhip_to_rgba(hip._h, _p(image), image.shape[1], image.shape[0], mode, _p(out), 0))
    return out


def to_rgba_bitmap(image, transparent=False, hip=None):
    return _to_rgba(image, 1 if transparent else 0, hip)


def to_debug_bitmap(image, hip=None):
    return _to_rgba(image, 2, hip)


def to_rgba_distance(image, hip=None):
    return _to_rgba(image, 3, hip)


# ---- fidget_mesh: leaf sampling on the device -----------------------------------------------------------------------
MESH_LEAF = np.dtype([("bounds", np.float32, 6), ("path", np.uint64), ("mask", np.uint32), ("n_edges", np.uint32), ("n_verts", np.uint32),
                      ("pad", np.uint32), ("inter", np.uint16, (12, 3)), ("pad2", np.uint16, 4), ("pos", np.float32, (12, 3)),
                      ("grad", np.float32, (12, 4)), ("vert", np.float32, (4, 3)), ("qef_err", np.float32, 4)])


def debug_walk_dual(cells, root, verts, parallel):
    """Octree::walk_dual of the library's host side on a given octree (fhip_debug_walk_dual): (triangles, vertices)"""
    cells = np.ascontiguousarray(cells, np.uint32)
    root = np.ascontiguousarray(root, np.uint32)
    verts = np.ascontiguousarray(verts, np.float32)
    c = np.zeros(2, np.uint64)
    lib().fhip_debug_walk_dual(_p(cells), len(cells), _p(root), _p(verts), len(verts), int(parallel), _p(c), None, None)
    tris = np.zeros((int(c[0]), 3), np.uint64)
    out = np.zeros((int(c[1]), 3), np.float32)
    lib().fhip_debug_walk_dual(_p(cells), len(cells), _p(root), _p(verts), len(verts), int(parallel), _p(c), _p(tris), _p(out))
    return tris, out


def libm_probe():
    """fhip_libm_probe: (number of probe arguments on which this host's libm differs from the routines the device restates, the first
    difference by name or "")"""
    buf = C.create_string_buffer(200)
    n = lib().fhip_libm_probe(buf, 200)
    return int(n), buf.value.decode()


class _MeshHandle:
    """owner of an fhip_mesh whose arrays numpy views borrow"""
    def __init__(self, h):
        self.h = h
        self._free = lib().fhip_mesh_free      # (held here: at interpreter shutdown the module's globals may be gone before the last view)

    def __del__(self):
        if self.h and self._free is not None:
            self._free(self.h)
            self.h = None

    def view(self, ptr, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        if not n or not ptr:
            return np.zeros(shape, dtype)
        buf = (C.c_uint8 * n).from_address(ptr)
        buf._owner = self           # (numpy keeps `buf` as the view's base, `buf` keeps the handle)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)


def mesh(shape, depth, world_to_model=None, vars=None):
    """fidget_mesh::Octree::build(...).walk_dual(): (triangles [n, 3] uint64, vertices [m, 3] float32, counts)"""
    return mesh_sample(shape, depth, world_to_model, vars, _build=True)


def mesh_part(shape, depth, part, n_parts, world_to_model=None, vars=None, alloc=None):
    """The device side of a mesh build for part `part` of `n_parts` (the root's octants o with o * n_parts // 8 == part;
    fhip_mesh_sample_part), as the flat uint8 buffer fhip_mesh_part_export writes: what a rank sends to the merging rank.
    `alloc(nbytes)` -> writable uint8 array to export into (shared memory, say); default: a fresh numpy array."""
    hip = shape.hip
    w2m = None if world_to_model is None else np.ascontiguousarray(world_to_model, np.float32)
    vk, vv = _var_arrays(shape, vars)
    ax = None
    if shape._vars is not None:
        ax = np.array(shape._vars, dtype=np.int32)
        vk = np.array([shape._named_slot(k) for k in (vars or {})], dtype=np.uint64)
    h = C.c_void_p()
    st = lib().fhip_mesh_sample_part(hip._h, shape._h, depth, _p(w2m), _p(ax), _p(vk), _p(vv), len(vk), part, n_parts, C.byref(h))
    if st == 4:
        raise ValueError("MissingVar")
    hip.check(st)
    try:
        n = int(lib().fhip_mesh_part_bytes(h))
        buf = np.zeros(n, np.uint8) if alloc is None else alloc(n)
        assert buf.dtype == np.uint8 and buf.size == n and buf.flags.c_contiguous
        lib().fhip_mesh_part_export(h, buf.ctypes.data_as(C.c_void_p))
    finally:
        lib().fhip_mesh_free(h)
    return buf


def mesh_merge(parts, world_to_model=None, hip=None):
    """fhip_mesh_merge: the buffers of all parts (parts[k] = mesh_part(.., k, len(parts))) -> (triangles, vertices, counts),
    the mesh of `mesh()` on one GPU.  `hip`: a context for the error text only (no device work)."""
    parts = [np.ascontiguousarray(b, np.uint8) for b in parts]
    w2m = None if world_to_model is None else np.ascontiguousarray(world_to_model, np.float32)
    ptrs = (C.c_void_p * len(parts))(*[b.ctypes.data for b in parts])
    sizes = np.array([b.size for b in parts], np.uint64)
    h = C.c_void_p()
    st = lib().fhip_mesh_merge(hip._h if hip is not None else None, ptrs, _p(sizes), len(parts), _p(w2m), C.byref(h))
    if st:
        if hip is not None:
            hip.check(st)
        raise RuntimeError(f"fhip_mesh_merge: status {st}")
    try:
        c = np.zeros(8, np.uint64)
        lib().fhip_mesh_counts(h, _p(c))
        verts = np.zeros((int(c[6]), 3), np.float32)
        tris = np.zeros((int(c[7]), 3), np.uint64)
        if len(verts):
            lib().fhip_mesh_vertices(h, _p(verts))
        if len(tris):
            lib().fhip_mesh_triangles(h, _p(tris))
    finally:
        lib().fhip_mesh_free(h)
    return tris, verts, {"cells": int(c[0]), "full": int(c[1]), "empty": int(c[2]), "leaf_cells": int(c[3]), "levels": int(c[5])}


def mesh_sample(shape, depth, world_to_model=None, vars=None, _build=False):
    """The evaluation side of fidget_mesh::Octree::build on the device (fhip_mesh_sample): returns (leaf records as a
    MESH_LEAF array, counts dict)."""
    hip = shape.hip
    w2m = None if world_to_model is None else np.ascontiguousarray(world_to_model, np.float32)
    vk, vv = _var_arrays(shape, vars)
    ax = None
    if shape._vars is not None:
        ax = np.array(shape._vars, dtype=np.int32)
        vk = np.array([shape._named_slot(k) for k in (vars or {})], dtype=np.uint64)
    h = C.c_void_p()
    fn = lib().fhip_mesh_build if _build else lib().fhip_mesh_sample
    st = fn(hip._h, shape._h, depth, _p(w2m), _p(ax), _p(vk), _p(vv), len(vk), C.byref(h))
    if st == 4:
        raise ValueError("MissingVar")
    hip.check(st)
    try:
        c = np.zeros(8, np.uint64)
        lib().fhip_mesh_counts(h, _p(c))
        assert int(c[4]) == MESH_LEAF.itemsize, (int(c[4]), MESH_LEAF.itemsize)
        counts = {"cells": int(c[0]), "full": int(c[1]), "empty": int(c[2]), "leaf_cells": int(c[3]), "levels": int(c[5])}
        if _build:
            # the arrays where the mesh holds them, no copy: numpy views whose base keeps the handle (freed with the last of them)
            owner = _MeshHandle(h)
            h = None
            verts = owner.view(lib().fhip_mesh_vertices_ptr(owner.h), (int(c[6]), 3), np.float32)
            tris = owner.view(lib().fhip_mesh_triangles_ptr(owner.h), (int(c[7]), 3), np.uint64)
            return tris, verts, counts
        leaves = np.zeros(int(c[3]), MESH_LEAF)
        if len(leaves):
            lib().fhip_mesh_leaves(h, _p(leaves))
    finally:
        lib().fhip_mesh_free(h)
    return leaves, {"cells": int(c[0]), "full": int(c[1]), "empty": int(c[2]), "leaf_cells": int(c[3]), "levels": int(c[5])}
