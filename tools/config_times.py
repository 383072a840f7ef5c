#!/usr/bin/env python3
"""Frame times of the other BASELINE.json configurations (parity cases, not bench lines), with the
oracle's time on the host cores beside them."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import fidget_amd as F
import oracle as O

def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
res = {}
m = os.path.join(ROOT, "models", "prospero.vm")
s = F.Shape.from_vm(m, hip=hip)
out2 = torch.zeros((4096, 4096), dtype=torch.float32, device="cuda")
res["C2 prospero 2D 4096^2"] = {"gpu_ms": timed(lambda: F.render2d(s, 4096, out=out2), 10)}
a = out2.cpu().numpy(); b, _, secs = O.render2d(O.Shape.from_vm(m), 4096)
res["C2 prospero 2D 4096^2"].update(oracle_ms=secs * 1e3, bit_exact=bool((a.view(np.uint32) == b.view(np.uint32)).all()))
m = os.path.join(ROOT, "models", "bear.vm")
s = F.Shape.from_vm(m, hip=hip)
out3 = torch.zeros((512, 512, 4), dtype=torch.int32, device="cuda")
res["C3 bear 3D 512^3"] = {"gpu_ms": timed(lambda: F.render3d(s, 512, out=out3), 10)}
g = out3.cpu().numpy().view(np.uint32); r, _, secs = O.render3d(O.Shape.from_vm(m), 512)
rn = r["normal"]; gn = g[..., :3].view(np.float32)
res["C3 bear 3D 512^3"].update(oracle_ms=secs * 1e3, depth_exact=bool((g[..., 3] == r["depth"]).all()),
                               normal_max_abs_err=float(np.abs(gn - rn).max()))
m = os.path.join(ROOT, "models", "prospero.vm")
s = F.Shape.from_vm(m, hip=hip)
n = 2048
out4 = torch.zeros((n, n, 4), dtype=torch.int32, device="cuda")
ms = timed(lambda: F.render3d(s, n, out=out4), 5)
g = out4.cpu().numpy().view(np.uint32); r, _, secs = O.render3d(O.Shape.from_vm(m), n)
res["prospero 3D 2048^3 (beyond BASELINE)"] = {"gpu_ms": ms, "mvoxel_per_s": n ** 3 / ms / 1e3, "oracle_ms": secs * 1e3,
    "bit_exact": bool((g[..., 3] == r["depth"]).all() and (g[..., :3].view(np.float32).view(np.uint32) == r["normal"].view(np.uint32)).all())}
print(json.dumps(res, indent=1))
