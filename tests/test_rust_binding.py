"""rust/fidget-hip/src/ffi.rs against include/fidget_hip.h.  The image has no Rust toolchain, so nothing compiles the crate here; what
can drift silently is the one file that restates the header.  This test parses both and demands, for every function ffi.rs declares:
the header declares it, with the same number of arguments, the same C types in the same order (pointer depth and constness included)
and the same return type; for the two config structs: the same fields, in the same order, of the same types.  It also holds the crate's
file list ("put the Rust crate in the tree as files")."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "rust", "fidget-hip")

BASE = {"uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "uint16_t": "u16", "int32_t": "i32", "int64_t": "i64", "int": "c_int", "size_t": "usize",
        "float": "f32", "double": "f64", "char": "c_char", "void": "c_void", "fhip_status": "fhip_status",
        "fhip_ctx": "fhip_ctx", "fhip_tape": "fhip_tape", "fhip_mesh": "fhip_mesh", "fhip_graph": "fhip_graph",
        "fhip_render2d_config": "fhip_render2d_config", "fhip_render3d_config": "fhip_render3d_config"}


def c_type(decl):
    """'const float* const* vars' / 'uint32_t info[4]' -> (rust spelling of the type, name)"""
    decl = decl.strip()
    arr = re.search(r"\[\w*\]\s*$", decl)
    if arr:
        decl = decl[:arr.start()]
    toks = re.findall(r"\w+|\*", decl)
    name = toks.pop() if (toks and toks[-1] != "*" and toks[-1] not in BASE and toks[-1] != "const" and len(toks) > 1) else None
    base, base_const, i = None, False, 0
    while i < len(toks) and toks[i] != "*":
        if toks[i] == "const":
            base_const = True
        elif toks[i] not in ("struct", "unsigned"):
            base = toks[i]
        i += 1
    t, pointee_const = BASE[base], base_const
    while i < len(toks):
        assert toks[i] == "*", decl
        t = ("*const " if pointee_const else "*mut ") + t
        pointee_const = False
        i += 1
        while i < len(toks) and toks[i] == "const":
            pointee_const = True
            i += 1
    if arr:
        t = ("*const " if pointee_const or base_const and "*" not in decl else "*mut ") + t if "*" not in decl else t
        if "*" not in decl:
            t = ("*const " if base_const else "*mut ") + BASE[base]
    return t, name


def header_functions():
    src = open(os.path.join(ROOT, "include", "fidget_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"typedef struct \w+ \{.*?\} \w+;", " ", src, flags=re.S)
    out = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(fhip_\w+)\s*\(([^;{}]*?)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        ret_t = None if ret == "void" else c_type(ret + " _r")[0]
        arg_t = [] if args in ("", "void") else [c_type(a)[0] for a in args.split(",")]
        out[name] = (arg_t, ret_t)
    return out


def header_struct(name):
    src = open(os.path.join(ROOT, "include", "fidget_hip.h")).read()
    body = re.search(r"typedef struct " + name + r" \{(.*?)\} " + name + ";", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", " ", body, flags=re.S)
    fields = []
    for stmt in body.split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        first, *more = [p.strip() for p in stmt.split(",")]
        t, n = c_type(first)
        fields.append((n, t))
        for extra in more:          # `uint32_t width, height, depth;`
            fields.append((extra, t))
    return fields


def rust_norm(t):
    return re.sub(r"\s+", " ", t.strip())


def rust_functions():
    src = open(os.path.join(CRATE, "src", "ffi.rs")).read()
    src = re.sub(r"//.*", "", src)
    block = re.search(r'extern "C" \{(.*)\}', src, re.S).group(1)
    out = {}
    for m in re.finditer(r"pub fn (\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", block, re.S):
        args = [a.strip() for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = ([rust_norm(a.split(":", 1)[1]) for a in args], rust_norm(m.group(3)) if m.group(3) else None)
    return out


def rust_struct(name):
    src = open(os.path.join(CRATE, "src", "ffi.rs")).read()
    src = re.sub(r"//.*", "", src)
    body = re.search(r"pub struct " + name + r" \{(.*?)\}", src, re.S).group(1)
    return [(m.group(1), rust_norm(m.group(2))) for m in re.finditer(r"pub (\w+)\s*:\s*([^,]+),", body)]


def test_c_type_parser():
    assert c_type("const float* const* vars") == ("*const *const f32", "vars")
    assert c_type("float* const* out") == ("*const *mut f32", "out")
    assert c_type("fhip_tape** out") == ("*mut *mut fhip_tape", "out")
    assert c_type("const fhip_tape* tape") == ("*const fhip_tape", "tape")
    assert c_type("uint32_t info[4]") == ("*mut u32", "info")
    assert c_type("const uint32_t split[3]") == ("*const u32", "split")
    assert c_type("const void* const* parts") == ("*const *const c_void", "parts")
    assert c_type("size_t n_words") == ("usize", "n_words")
    assert c_type("void* stream") == ("*mut c_void", "stream")


def test_every_function_of_the_binding_is_the_headers():
    hdr, rs = header_functions(), rust_functions()
    assert len(rs) >= 38 and len(hdr) >= len(rs)
    for name, (args, ret) in rs.items():
        assert name in hdr, f"ffi.rs declares {name}, include/fidget_hip.h does not"
        hargs, hret = hdr[name]
        assert len(args) == len(hargs), f"{name}: {len(args)} arguments in ffi.rs, {len(hargs)} in the header"
        for k, (a, b) in enumerate(zip(args, hargs)):
            assert a == b or {a, b} == {"c_int", "fhip_status"}, f"{name}: argument {k} is `{a}` in ffi.rs and `{b}` in the header"
        assert ret == hret, f"{name}: returns `{ret}` in ffi.rs and `{hret}` in the header"
    # the surface a binding needs: nothing of it may be missing from ffi.rs
    for name in ("fhip_ctx_create", "fhip_ctx_destroy", "fhip_tape_from_bytecode", "fhip_tape_free", "fhip_simplify", "fhip_interval_eval", "fhip_point_eval",
                 "fhip_float_eval", "fhip_grad_eval", "fhip_render2d", "fhip_render3d", "fhip_render3d_shard", "fhip_render3d_block", "fhip_merge_depth",
                 "fhip_mesh_build", "fhip_mesh_counts", "fhip_mesh_vertices", "fhip_mesh_triangles", "fhip_mesh_free", "fhip_cancel", "fhip_libm_probe"):
        assert name in rs, name


def test_config_structs_have_the_headers_fields_in_order():
    for name in ("fhip_render2d_config", "fhip_render3d_config"):
        h, r = header_struct(name), rust_struct(name)
        assert [n for n, _ in h] == [n for n, _ in r], (name, h, r)
        for (n, a), (_, b) in zip(h, r):
            assert a == b, f"{name}.{n}: `{b}` in ffi.rs, `{a}` in the header"


def test_the_crate_is_in_the_tree_and_calls_only_what_it_binds():
    for f in ("Cargo.toml", "build.rs", "README.md", "src/ffi.rs", "src/lib.rs", "src/render.rs", "src/mesh.rs", "tests/eval.rs", "tests/render.rs"):
        assert os.path.exists(os.path.join(CRATE, f)), f
    rs = rust_functions()
    for f in ("src/lib.rs", "src/render.rs", "src/mesh.rs"):
        for called in re.findall(r"ffi::(fhip_\w+)", open(os.path.join(CRATE, f)).read()):
            assert called in rs or called.endswith("_config") or called in ("fhip_ctx", "fhip_tape", "fhip_mesh", "fhip_status"), (f, called)
    ev = open(os.path.join(CRATE, "tests", "eval.rs")).read()
    for macro in ("interval_tests!", "float_slice_tests!", "grad_slice_tests!", "point_tests!"):     # fidget-jit/src/lib.rs:1381-1384
        assert f"fidget_core::{macro}(HipFunction)" in ev
    cargo = open(os.path.join(CRATE, "Cargo.toml")).read()
    assert 'features = ["eval-tests"]' in cargo and "fidget-bytecode" in cargo
