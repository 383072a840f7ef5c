#!/usr/bin/env python3
"""Build a variant of the whole library (no GPU) with extra compiler flags into fidget_amd/csrc/_gen/variants/lib_<name>.so, for A/B runs under
FHIP_LIB=<that file> (tools/variants.py name@FHIP_LIB=...).  The assembly kernels are the current build's (_gen/interp_gfx950.co).
usage: tools/build_lib_variant.py <name> [-DFLAG=..] ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fidget_amd", "csrc")
name, flags = sys.argv[1], sys.argv[2:]
vd = os.path.join(CSRC, "_gen", "variants")
os.makedirs(vd, exist_ok=True)
out = os.path.join(vd, f"lib_{name}.so")
co = os.path.join(CSRC, "_gen", "interp_gfx950.co")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", f'-DFH_INTERP_CO="{co}"'] + flags +
                      ["-o", out, os.path.join(CSRC, "capi.hip")])
print(out)
