import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import fidget_amd as F
os.environ["FHIP_STATS"] = "1"
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
shape = F.Shape.from_vm("/root/repo/models/prospero.vm", hip=hip)
out = torch.zeros((1024, 1024, 4), dtype=torch.int32, device="cuda")
for lvl in (0, 1):
    hip.set_option("prune2_probe_level", lvl)
    for _ in range(2):
        F.render3d(shape, 1024, out=out); hip.sync()
    print("level", lvl, "phase clocks max (A, B1, B2, B3+B4):", hip.leaf_stats()["prune2_phase_clocks_max"])
