"""Micro-benchmark of the point interpreter variants on a pruned prospero tape (diagnostics)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fidget_amd as F
L = F.lib()
L.fhip_debug_bench.restype = C.c_int
L.fhip_debug_bench.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_double)]
s = F.Shape.from_vm("models/prospero.vm")
hip = s.hip
# prune with a real trace from a small tile
_, tr = s.eval_interval((0.10, 0.11), (0.30, 0.31), (0, 0))
c = s.simplify(tr)
print("child len", c.size(), "regs", c.slot_count())
for tape, name in ((c, "child"), (s, "root")):
    for variant, vn, zb in ((0, "vgpr16x4", 4), (1, "vgpr32x2", 2), (3, "vgpr32x1", 1), (2, "lds", 1)):
        if variant != 2 and tape.slot_count() > (16 if variant == 0 else 32):
            continue
        for waves in (1, 256, 2048, 8192):
            reps = 2000 if tape is c else 20
            ms = C.c_double()
            st = L.fhip_debug_bench(hip._h, tape._h, waves, reps, variant, C.byref(ms))
            steps = reps * tape.size()
            ns_step = ms.value * 1e6 / steps
            print(f"{name:5s} {vn:9s} waves={waves:5d}: {ms.value:8.3f} ms  {ns_step:7.1f} ns/step/wave-chain  "
                  f"throughput {waves*steps*zb*64/ms.value/1e6:9.1f} M lane-ops/s ({waves*steps/ms.value/1e6:7.2f} G steps/s)")
