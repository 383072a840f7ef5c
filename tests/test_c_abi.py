"""The boundary from a native caller: tests/c_abi/render_hi.c, plain C11 against include/fidget_hip.h only (the closest stand-in
for the Rust FFI of INTEGRATION.md that this image can compile).  Without a GPU: the header is valid strict C11, every entry
point the program uses links and loads, and the config structs are laid out as the ctypes mirror in fidget_amd/__init__.py
types them by hand.  With one: the program renders hi.vm and finds the reference's golden image (pixel_render.rs:75-106)."""
import ctypes as C
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "render_hi.c")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    import fidget_amd as F
    F.build()
    out = str(tmp_path_factory.mktemp("c_abi") / "render_hi")
    libdir = os.path.dirname(F.LIB_PATH)
    # (--allow-shlib-undefined: the HIP runtime the library needs is found when it is loaded, as for any other caller)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), SRC, "-o", out,
                           "-L", libdir, "-lfidget_hip", f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined"])
    return out


def test_c_program_compiles_links_and_loads(exe):
    r = subprocess.run([exe, "--symbols"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


def test_config_struct_layouts_match_the_ctypes_mirror(exe):
    import fidget_amd as F
    lay = json.loads(subprocess.run([exe, "--layout"], capture_output=True, text=True, timeout=120, check=True).stdout)
    for cname, T in (("fhip_render2d_config", F._Cfg2D), ("fhip_render3d_config", F._Cfg3D)):
        assert lay[f"sizeof.{cname}"] == C.sizeof(T)
        names = [f[0] for f in T._fields_]
        in_c = [k.split(".")[1] for k in lay if k.startswith(cname + ".")]
        assert names == in_c, (names, in_c)            # same fields, same order
        for n in names:
            d = getattr(T, n)
            assert lay[f"{cname}.{n}"] == [d.offset, d.size], (cname, n)


@pytest.mark.gpu
def test_c_program_renders_hi_and_matches_the_golden_image(exe):
    r = subprocess.run([exe, os.path.join(ROOT, "models", "hi.vm"), os.path.join(ROOT, "tests", "golden", "pixel_render_check_hi_EXPECTED.txt")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 of 1024 pixels differ" in r.stdout


def test_libm_probe_names_this_hosts_libm():
    """fhip_libm_probe (no device): on the deployment image - glibc 2.35, x86-64 with FMA - the restated routines are the running libm's,
    0 of 288 arguments differ; the message buffer is left empty and a small buffer is respected."""
    import fidget_amd as F
    n, first = F.libm_probe()
    assert n == 0 and first == "", (n, first)
    import ctypes as C
    assert F.lib().fhip_libm_probe(None, 0) == 0
    buf = C.create_string_buffer(4)
    assert F.lib().fhip_libm_probe(buf, 4) == 0 and buf.value == b""
