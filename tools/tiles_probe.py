#!/usr/bin/env python3
"""GPU box: per-level shader clocks of the VGPR tile kernels (flags bit 0 probes) for one frame, slabs serialised."""
import os, sys, json
os.environ["FHIP_PROBE"] = "1"
os.environ["FHIP_NO_PIPELINE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fidget_amd as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = sys.argv[2] if len(sys.argv) > 2 else "prospero.vm"
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
shape = F.Shape.from_vm(os.path.join(ROOT, "models", model), hip=hip)
out = torch.zeros((n, n, 4), dtype=torch.int32, device="cuda")
for _ in range(2):
    F.render3d(shape, n, out=out)
hip.sync()
hip.wave_stats()
res = {"tile_v": hip.tile_v, "tile_phases": hip.tile_phases}
for l, v in hip.tile_v.items():
    ops = hip.tile_phases.get(l, {}).get("ops", 0)
    if ops:
        v["fwd_clocks_per_op"] = v["fwd_clocks"] / ops
        v["prune_clocks_per_op"] = v["prune_clocks"] / ops
        v["mean_slot_clocks"] = (v["fwd_clocks"] + v["prune_clocks"]) / max(v["slots"], 1)
print(json.dumps(res, indent=1))
