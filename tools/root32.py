#!/usr/bin/env python3
"""GPU box: 3D frames with root tiles of 32^3 straight above the leaves (tile_sizes [32, 8]: ONE coarse level - the root tape pruned per
32^3 tile by the linked prune, no level 1) against the default 128 / 32 / 8, lone and queued, image compared.
usage: root32.py [model]       (ROOT32_QUICK=1: no frame lanes, column invariance on, the library's tile choice only)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fidget_amd as F
model = sys.argv[1] if len(sys.argv) > 1 else "prospero.vm"
QUICK = bool(os.environ.get("ROOT32_QUICK"))
for lanes in ((0,) if QUICK else (0, 4)):
  hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
  hip.set_option("frame_lanes", lanes)
  shape = F.Shape.from_vm(os.path.join(ROOT, "models", model), hip=hip)
  for no_inv in ((0,) if QUICK else (0, 1)):
    hip.set_option("no_column_inv", no_inv)
    for what in ("512", "256", "128", "1024", "2048", "octant"):
        n = 1024 if what == "octant" else int(what)
        kw = {"block": (7, (2, 2, 2))} if what == "octant" else {}
        imgs = {}
        for tiles in (("auto",) if QUICK else ("128/32/8", "auto")):
            hip.set_option("root32_max", 0 if tiles == "128/32/8" else int(os.environ.get("ROOT32_MAX", "4096")))
            out = torch.zeros((n, n, 4), dtype=torch.int32, device="cuda")
            call = lambda: F.render3d(shape, n, out=out, **kw)
            try:
                call(); hip.sync()
                lone = []
                for _ in range(5):
                    t0 = time.perf_counter(); call(); hip.sync(); lone.append((time.perf_counter() - t0) * 1e3)
                for _ in range(80 if lanes else 10): call()
                hip.sync()
                t0 = time.perf_counter()
                for _ in range(40): call()
                hip.sync()
                q = (time.perf_counter() - t0) / 40 * 1e3
                imgs[str(tiles)] = out.clone()
                print(f"lanes {lanes} no_inv {no_inv} {what:>6} tiles {str(tiles):>8}: lone {min(lone):.3f} ms, queued {q:.3f} ms", flush=True)
            except Exception as e:
                print(f"lanes {lanes} no_inv {no_inv} {what:>6} tiles {tiles}: FAILED {e!r}", flush=True)
        if len(imgs) == 2:
            a, b = imgs.values()
            print(f"   images equal: {bool(torch.equal(a, b))}  (differing pixels {int((a != b).any(dim=2).sum())})", flush=True)
  del shape, hip
