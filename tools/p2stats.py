"""GPU box: the linked prune's phases (k_prune2, shader clocks of the slowest child per phase: A choices, B1 liveness, B2 positions, B3 + B4
register scan and emission) for the root level of a frame.  usage: p2stats.py [size] [root32_max]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import fidget_amd as F
os.environ["FHIP_STATS"] = "1"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
if len(sys.argv) > 2:
    hip.set_option("root32_max", int(sys.argv[2]))
shape = F.Shape.from_vm(os.path.join(ROOT, "models", "prospero.vm"), hip=hip)
out = torch.zeros((n, n, 4), dtype=torch.int32, device="cuda")
for _ in range(2):
    F.render3d(shape, n, out=out); hip.sync()
print(n, "phase clocks max (A, B1, B2, B3+B4):", hip.leaf_stats()["prune2_phase_clocks_max"], flush=True)
hip.wave_stats()
print("tile levels:", hip.tile_phases, flush=True)
slab = (n + 511) // 512 - 1       # (the front slab: the only one a frame without z renders)
g, cnt = hip.groups(1, slab)
print(f"parked parents of slab {slab}:", cnt, "len min/median/max", int(g["len"].min()), int(np.median(g["len"])), int(g["len"].max()), "regs max", int(g["regs"].max()), "choices max", int(g["choices"].max()), flush=True)
