"""Throughput of the assembly bulk interpreter on a pruned prospero tape (run under rocprofv3
--kernel-trace --stats and read the fh_float_eval_* line)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fidget_amd as F
s = F.Shape.from_vm(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models", "prospero.vm"))
n = 1 << 24
x0 = np.random.rand(n).astype(np.float32)
z = np.zeros(n, np.float32)
for w in (0.01, 0.03, 0.06, 0.1):
    _, tr = s.eval_interval((0.10, 0.10 + w), (0.30, 0.30 + w), (0, 0))
    c = s.simplify(tr)
    print("w", w, "child len", c.size(), "regs", c.slot_count(), flush=True)
    if c.slot_count() > 32:
        continue
    x = x0 * w + 0.10
    y = x0[::-1] * w + 0.30
    for _ in range(2):
        out = c.eval_float_slice(x, y, z)
