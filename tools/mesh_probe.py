import os, sys, time, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np
import fidget_amd as F, oracle as O
res = {}
for model, depth in (("colonnade.vm", 8), ("prospero.vm", 7), ("gyroid-sphere.vm", 7), ("bear.vm", 6)):
    p = os.path.join("models", model)
    t0 = time.time(); tris, verts, counts = F.mesh(F.Shape.from_vm(p), depth); tg = time.time() - t0
    t0 = time.time(); t, v = O.Octree(O.Shape.from_vm(p), depth).walk_dual(); to = time.time() - t0
    t, v = np.asarray(t, np.uint64).reshape(-1, 3), np.asarray(v, np.float32).reshape(-1, 3)
    r = {"gpu_s": tg, "oracle_s": to, "tris": [len(tris), len(t)], "verts": [len(verts), len(v)]}
    if tris.shape == t.shape and verts.shape == v.shape:
        r["tris_equal"] = bool((tris == t).all())
        d = np.abs(verts - v).max(axis=1)
        r["verts_bit_equal_fraction"] = float((verts.view(np.uint32) == v.view(np.uint32)).all(axis=1).mean())
        r["vert_max_abs_diff"] = float(d.max()); r["vert_p999"] = float(np.percentile(d, 99.9)); r["n_above_1e-5"] = int((d > 1e-5).sum()); r["n_above_1e-3"] = int((d > 1e-3).sum())
        if not r["tris_equal"]:
            r["tris_differ"] = int((tris != t).any(axis=1).sum())
    res[model + "@" + str(depth)] = r
    print(model, depth, r, flush=True)
os.makedirs("gpurun_out/r03e", exist_ok=True)
json.dump(res, open("gpurun_out/r03e/mesh_probe.json", "w"), indent=1)
