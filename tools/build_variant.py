#!/usr/bin/env python3
"""Build a variant of the assembly kernels' code object (no GPU): gen_interp.py under the given environment (FH_EXP=..., FH_BLKL=...)
-> fidget_amd/csrc/_gen/variants/<name>.co, for tools/variants.py (FHIP_INTERP_CO).  usage: tools/build_variant.py <name> [K=V ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fidget_amd", "csrc")
gen = os.path.join(CSRC, "_gen")
name = sys.argv[1]
env = dict(os.environ)
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1)
    env[k] = v
vd = os.path.join(gen, "variants")
os.makedirs(vd, exist_ok=True)
llvm = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin")
s, o, co = (os.path.join(vd, name + e) for e in (".s", ".o", ".co"))
subprocess.check_call([sys.executable, os.path.join(CSRC, "gen_interp.py"), os.path.join(gen, "offsets.json"), s, os.path.join(gen, "trans_funcs.s")], env=env)
subprocess.check_call([os.path.join(llvm, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o])
subprocess.check_call([os.path.join(llvm, "ld.lld"), "-shared", o, "-o", co])
os.remove(o)
print(co)
