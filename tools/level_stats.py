#!/usr/bin/env python3
"""GPU box: shape of the tile stage's work per level (parents, tape length / registers / choices) for a 3D frame."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import fidget_amd as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = sys.argv[2] if len(sys.argv) > 2 else "prospero.vm"
os.environ["FHIP_STATS"] = "1"
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
shape = F.Shape.from_vm(os.path.join(ROOT, "models", model), hip=hip)
out = torch.zeros((n, n, 4), dtype=torch.int32, device="cuda")
F.render3d(shape, n, out=out)
hip.sync()
res = {"model": model, "size": n, "root": {"len": shape.size(), "regs": shape.slot_count(), "choices": shape.choice_count()}}


def describe(g):
    if len(g) == 0:
        return {"n": 0}
    q = lambda a: [float(v) for v in np.percentile(a, [50, 90, 99, 100])]
    d = {"n": int(len(g)), "len_sum": int(g["len"].sum()), "len_p50_90_99_max": q(g["len"]), "regs_p50_90_99_max": q(g["regs"]),
         "choices_p50_90_99_max": q(g["choices"])}
    for r in (8, 16, 24, 32, 48, 64, 128):
        d[f"regs<={r}"] = float((g["regs"] <= r).mean())
    for c in (64, 128, 256, 512, 768):
        d[f"choices<={c}"] = float((g["choices"] <= c).mean())
    d["fits_32r_256c"] = float(((g["regs"] <= 32) & (g["choices"] <= 256)).mean())
    d["fits_32r_256c_opshare"] = float(g["len"][(g["regs"] <= 32) & (g["choices"] <= 256)].sum() / max(g["len"].sum(), 1))
    d["fits_64r_768c"] = float(((g["regs"] <= 64) & (g["choices"] <= 768)).mean())
    d["fits_64r_768c_opshare"] = float(g["len"][(g["regs"] <= 64) & (g["choices"] <= 768)].sum() / max(g["len"].sum(), 1))
    d["len<=64"] = float((g["len"] <= 64).mean())
    return d


g1, c1 = hip.groups(0, 1)
res["level1_parents"] = describe(g1); res["level1_counts"] = c1
allg = []
for k in range((n + 127) // 128):
    g, c = hip.groups(1, k)
    res[f"slab{k}_level2_parents"] = describe(g)
    allg.append(g)
res["level2_parents_all_slabs"] = describe(np.concatenate(allg))
lv = hip.last_leaves()
res["leaves_last_slab"] = describe(lv)
hip.wave_stats()
res["tile_phases"] = hip.tile_phases
print(json.dumps(res, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"level_stats_{model}_{n}.json"), "w"), indent=1)
