#!/usr/bin/env python3
"""Opcode mix of the leaf tapes of the last slab, and how often an op writes the register it reads
(in-place: the interpreters then need no copy through temporaries).  GPU box."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import fidget_amd as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = sys.argv[2] if len(sys.argv) > 2 else "prospero.vm"
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
shape = F.Shape.from_vm(os.path.join(ROOT, "models", model), hip=hip)
out = torch.zeros((n, n, 4), dtype=torch.int32, device="cuda")
F.render3d(shape, n, out=out)
hip.sync()
lv = hip.last_leaves()
rng = np.random.default_rng(1)
pick = rng.choice(len(lv), size=min(4000, len(lv)), replace=False)
NAMES = "OUTPUT INPUT COPY_REG COPY_IMM NEG ABS RECIP SQRT SQUARE FLOOR CEIL ROUND SIN COS TAN ASIN ACOS ATAN EXP LN NOT RAND".split()
BIN = "ADD SUB MUL DIV ATAN2 COMPARE MIX MOD MIN MAX AND OR".split()
def name(op):
    if op < 22: return NAMES[op] if op < len(NAMES) else f"u{op}"
    if op < 34: return BIN[op - 22] + "_RR"
    if op < 46: return BIN[op - 34] + "_RI"
    return ["SUB", "DIV", "ATAN2", "COMPARE", "MIX", "MOD"][op - 46] + "_IR"
cnt = collections.Counter(); inpl_a = collections.Counter(); inpl_b = collections.Counter()
for i in pick:
    ops = hip.arena_ops(lv["off"][i], lv["len"][i])
    w0 = (ops & 0xFFFFFFFF).astype(np.uint32); w1 = (ops >> 32).astype(np.uint32)
    op = w0 & 0xFF; o = (w0 >> 8) & 0xFFF; a = w0 >> 20
    for k in range(len(ops)):
        nm = name(int(op[k])); cnt[nm] += 1
        has_a = op[k] not in (1, 3, 0)
        if has_a and a[k] == o[k]: inpl_a[nm] += 1
        elif 22 <= op[k] < 34 and w1[k] == o[k]: inpl_b[nm] += 1
tot = sum(cnt.values())
print(model, n, "leaves in the last slab", len(lv), "tape length p50/p90/max", np.percentile(lv["len"], [50, 90, 100]), "registers p50/p90/max", np.percentile(lv["regs"], [50, 90, 100]))
print("ops sampled", tot, "from", len(pick), "leaves")
for nm, c in cnt.most_common():
    print(f"{nm:12s} {100*c/tot:5.1f} %   out==a {100*inpl_a[nm]/c:5.1f} %   out==b {100*inpl_b[nm]/c:5.1f} %")
print("in place overall: a", 100*sum(inpl_a.values())/tot, " b", 100*sum(inpl_b.values())/tot)
