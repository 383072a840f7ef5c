"""Host code under AddressSanitizer + UndefinedBehaviorSanitizer (the reference runs its unsafe code under sanitizers).

Three legs, none needs a GPU:
  * the library's host side - graph, SSA flattening, dense allocation, the RegisterAllocator with spills, bytecode producer and
    importer, root split, term plan, links - built with g++ from the headers capi.hip includes (tests/host_build/host_frontend_san.cpp),
    over every model and 20 000 hostile bytecode buffers;
  * the mesher's host-side passes (mesh_collapse.hpp, mesh_edges.hpp: compiled for host and device alike) through their own tests;
  * the oracle (test infrastructure, but its verdicts are only worth what its memory safety is) through a part of the CPU suite.
The last two run in a child Python with the sanitizer runtimes preloaded and FIDGET_SANITIZE=1, which makes oracle/ and the host
builds compile and load their `_san` variants."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]


def _runtime(name):
    p = subprocess.check_output(["g++", f"-print-file-name={name}"], text=True).strip()
    if not os.path.isabs(p):
        pytest.skip(f"{name} not installed")
    return p


def test_host_front_end_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_frontend_san")
    subprocess.check_call(["g++", "-std=c++17"] + SAN + ["-I", os.path.join(ROOT, "fidget_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host_build", "host_frontend_san.cpp"), "-o", exe])
    models = sorted(glob.glob(os.path.join(ROOT, "models", "*.vm")))
    r = subprocess.run([exe] + models, capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "hostile inputs refused or handled" in r.stdout and r.stdout.count(" ops, ") == len(models)
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def test_oracle_and_mesh_host_passes_under_asan_ubsan():
    env = dict(os.environ, FIDGET_SANITIZE="1", LD_PRELOAD=":".join([_runtime("libasan.so"), _runtime("libubsan.so")]),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="4")
    tests = ["tests/test_compiler_kat.py", "tests/test_reference_units.py", "tests/test_render_golden.py", "tests/test_mesh_edges.py",
             "tests/test_mesh_assembly.py", "tests/test_effects.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider", "-k", "not prospero and not bear"] + tests,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, tail
    assert " passed" in r.stdout
