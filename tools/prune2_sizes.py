import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import fidget_amd as F
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
def t3(model, n, reps=10, **kw):
    shape = F.Shape.from_vm(os.path.join("/root/repo/models", model), hip=hip)
    out = torch.zeros((n, n, 4), dtype=torch.int32, device="cuda")
    for _ in range(3): F.render3d(shape, n, out=out, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): F.render3d(shape, n, out=out, **kw)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for opts in ({}, {"prune2": 0}):
    for k, v in opts.items(): hip.set_option(k, v)
    print(opts, "prospero 2048", round(t3("prospero.vm", 2048), 3), "prospero 512", round(t3("prospero.vm", 512), 3), "prospero 256", round(t3("prospero.vm", 256), 3),
          "bear 512", round(t3("bear.vm", 512), 3), "colonnade 1024", round(t3("colonnade.vm", 1024), 3), "colonnade 2048", round(t3("colonnade.vm", 2048), 3), flush=True)
