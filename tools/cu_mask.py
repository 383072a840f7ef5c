#!/usr/bin/env python3
"""GPU box: the pipelined frame rate with the side stream (level 1 of the coarse levels: ~500 waves, one per parent, each alone on its
SIMD at best) on compute units of its own (context option side_cus, fixed at creation: FHIP_SIDE_CUS), the pre-pass and tail streams on
the others.  Frames are queued on a stream of the caller's that is not the null stream (masked streams are blocking ones).
usage: tools/cu_mask.py [frames]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import gc
    import torch
    import fidget_amd as F
    frames = int(sys.argv[2])
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        hip = F.HipContext(0, s.cuda_stream)
        shape = F.Shape.from_vm(os.path.join(ROOT, "models", "prospero.vm"), hip=hip)
        out = torch.zeros((1024, 1024, 4), dtype=torch.int32, device="cuda")
        res = {}
        for general in (0, 1):
            hip.set_option("no_column_inv", general)
            for _ in range(6):
                F.render3d(shape, 1024, out=out)
            torch.cuda.synchronize()
            gc.collect(); gc.disable()
            t0 = time.perf_counter()
            for _ in range(frames):
                F.render3d(shape, 1024, out=out)
            torch.cuda.synchronize()
            res["general" if general else "default"] = round((time.perf_counter() - t0) / frames * 1e3, 4)
            gc.enable()
            res["sum_" + ("general" if general else "default")] = int(out.to(torch.int64).sum().item())
    print(os.environ.get("FHIP_SIDE_CUS", "0"), res, flush=True)
else:
    frames = sys.argv[1] if len(sys.argv) > 1 else "100"
    for n in ("0", "64", "96", "128", "160", "192"):
        subprocess.run([sys.executable, __file__, "child", frames], env=dict(os.environ, FHIP_SIDE_CUS=n))
