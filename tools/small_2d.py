#!/usr/bin/env python3
"""GPU box: small 2D images of prospero.vm, lone and queued, with the small-image passes (render2d_frame) and without (root32_max 0)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fidget_amd as F
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
hip.set_option("frame_lanes", int(os.environ.get("LANES", "0")))
shape = F.Shape.from_vm(os.path.join(ROOT, "models", "prospero.vm"), hip=hip)
for n in (64, 128, 256, 512, 704, 1024):
    imgs = []
    for r32 in (0, 4096):
        hip.set_option("root32_max", r32)
        out = torch.zeros((n, n), dtype=torch.float32, device="cuda")
        call = lambda: F.render2d(shape, n, out=out)
        call(); hip.sync()
        lone = []
        for _ in range(5):
            t0 = time.perf_counter(); call(); hip.sync(); lone.append((time.perf_counter() - t0) * 1e3)
        for _ in range(10): call()
        hip.sync()
        t0 = time.perf_counter()
        for _ in range(40): call()
        hip.sync()
        q = (time.perf_counter() - t0) / 40 * 1e3
        imgs.append(out.clone())
        print(f"2D {n:>5}^2 small-image passes {'on ' if r32 else 'off'}: lone {min(lone):.3f} ms, queued {q:.3f} ms", flush=True)
    print("   images equal:", bool(torch.equal(imgs[0].view(torch.int32), imgs[1].view(torch.int32))), flush=True)
