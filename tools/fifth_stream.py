#!/usr/bin/env python3
"""GPU box: what a collective per frame on a stream of its own (as RCCL's) does to the pipelined frames - stand-in: a 16 MiB
elementwise kernel on a fifth stream, ordered after the frame and before the next frame's use of `out` by events."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fidget_amd as F
n = 1024
stream = torch.cuda.current_stream()
hip = F.HipContext(0, stream.cuda_stream)
shape = F.Shape.from_vm(os.path.join(ROOT, "models", "prospero.vm"), hip=hip)
out = torch.zeros((n, n, 4), dtype=torch.int32, device="cuda")
dummies = [torch.cuda.Stream() for _ in range(int(os.environ.get("DUMMIES", "0")))]
side = torch.cuda.Stream()
out2 = torch.zeros((n, n, 4), dtype=torch.int32, device="cuda")
mode = sys.argv[1] if len(sys.argv) > 1 else "none"


def step():
    F.render3d(shape, n, out=out)
    if mode == "record":            # an event on the caller's stream after the frame, nothing else
        ev = torch.cuda.Event()
        ev.record(stream)
    elif mode == "same":            # the elementwise kernel on the caller's stream itself
        out.add_(0)
    elif mode == "free":            # the kernel on a fifth stream, no ordering with the frame at all
        with torch.cuda.stream(side):
            out2.add_(0)
    elif mode == "waitonly":        # the fifth stream waits for the frame's end and does nothing
        ev = torch.cuda.Event()
        ev.record(stream)
        side.wait_event(ev)
    elif mode == "nowait":          # the kernel on a fifth stream after the frame; the caller's stream does not wait for it
        ev = torch.cuda.Event()
        ev.record(stream)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            out2.add_(0)
    elif mode == "fifth":
        ev = torch.cuda.Event()
        ev.record(stream)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            out.add_(0)
        ev2 = torch.cuda.Event()
        ev2.record(side)
        stream.wait_event(ev2)


for _ in range(5):
    step()
torch.cuda.synchronize()
K = 60
t0 = time.perf_counter()
for _ in range(K):
    step()
torch.cuda.synchronize()
print(f"{mode}: tail={os.environ.get('FHIP_TAIL_STREAM', 'default')} no_inv={os.environ.get('FHIP_NO_COLUMN_INV', '-')}  {(time.perf_counter() - t0) / K * 1e3:.3f} ms per frame")
