#!/usr/bin/env python3
"""Instructions per tape op of the assembly interpreters, counted on the gfx950 emulator (no GPU): the numbers behind the
"instruction issue" bound of DESIGN.md section 6.  prospero.vm's own tapes: a 32^3 parent of the per-slab level (fh_tiles_v32,
fh_tiles_v64, fh_tiles), a 128^3 root tile's tape (fh_tiles_v64, fh_tiles), the root tape (fh_prune1), a leaf tape (fh_columns).
usage: tools/emu_instr_counts.py [out.json]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import emu_util as U
import test_emu_tiles as T
import test_emu_prune as P
import test_emu_columns as C

res = {}


def counts(w):
    c = dict(w.counts)
    c["total"] = int(w.n_inst)
    return c


def per_op(c, n):
    return {k: round(v / n, 2) for k, v in c.items()}


ik, ch = T.chain()
for label, (tape, regs, nch, center, half), kernels in (("level-2 parent (32^3 tile, per-slab level)", ch[1], ("fh_tiles_v32", "fh_tiles_v64", "fh_tiles")),
                                                        ("level-1 parent (128^3 root tile)", ch[0], ("fh_tiles_v64", "fh_tiles"))):
    for kernel in kernels:
        r, kids, _ = T.check_slot(kernel, tape, T.children(center, half), ik, regs, nch)
        both = counts(r["wave"])
        # the same slot far outside the model: every child empty, nothing to prune -> the forward pass alone
        far = counts(T.run_tiles(kernel, tape, T.children((40.0, 40.0, 40.0), half), ik, regs, nch)["wave"])
        res[f"{kernel}: {label}"] = {"tape_ops": int(len(tape)), "registers": int(regs), "choices": int(nch), "children_pruned": len(kids),
                                     "kept_ops_mean": float(np.mean([len(k[0]) for k in kids.values()])) if kids else 0.0,
                                     "forward_pass_per_op": per_op(far, len(tape)),
                                     "forward_and_prune_per_op": per_op(both, len(tape)),
                                     "prune_sweep_per_op": per_op({k: both.get(k, 0) - far.get(k, 0) for k in both}, len(tape))}
        print(kernel, label, res[f"{kernel}: {label}"], flush=True)

# fh_prune1 on the root tape, children of a root tile
import fidget_amd as F
sh = F.Shape.from_vm(os.path.join(ROOT, "models", "prospero.vm"))
tape, n_regs = U.shape_tape(sh), sh.slot_count()
ikr = [3] * 16
for a in range(3):
    s = sh.axis_index(a)
    if s >= 0:
        ikr[s] = a
xyz = T.children((0.125, 0.125, 0.125), 0.125)
inputs = {s: (xyz[2 * k], xyz[2 * k + 1]) for s, k in enumerate(ikr) if k < 3}
el, eh, chs, _ = U.ref_interval(tape, inputs, 64)
marked = [int(k) for k in np.nonzero(~(eh < 0) & ~(el > 0) & (chs != 3).any(axis=0))[0]][:4]
waves = []
orig = U.E.launch
U.E.launch = lambda *a, **k: (lambda w: (waves.extend(w), w)[1])(orig(*a, **k))
r = P.run_prune1(tape, chs, marked, n_regs)
U.E.launch = orig
busy = [w for w in waves if w.n_inst > 200]
kept = [int(r["clen"][l]) for l in marked]
tot = {k: sum(counts(w)[k] for w in busy) for k in counts(busy[0])}
res["fh_prune1: root tape, children of a root tile"] = {"tape_ops": int(len(tape)), "registers": int(n_regs), "children": len(marked), "kept_ops": kept,
                                                        "batches_of_64_ops": (len(tape) + 63) // 64,
                                                        "per_child": {k: round(v / len(busy), 1) for k, v in tot.items()},
                                                        "per_kept_op (batch overhead included)": per_op(tot, sum(kept))}
print("fh_prune1", res["fh_prune1: root tape, children of a root tile"], flush=True)

# fh_columns on leaf tapes: children of the level-2 parent
tape2, regs2, nch2, center2, half2 = ch[1]
xyz = T.children(center2, half2)
inputs = {s: (xyz[2 * k], xyz[2 * k + 1]) for s, k in enumerate(ik) if k < 3}
el, eh, chs, _ = U.ref_interval(tape2, inputs, 64)
by_class = {}
for lane in np.nonzero(~(eh < 0) & ~(el > 0))[0]:
    ops, lregs, _ = U.ref_prune(tape2, chs[:, lane])
    cls = 10 if lregs <= 10 else (20 if lregs <= 20 else 32)       # (the leaf kernel's register-file shapes: 10 x 8 voxels, 20 x 4, 32 x 2)
    by_class.setdefault(cls, []).append((len(ops), lregs, np.array(ops, np.uint64)))
mat = np.eye(4, dtype=np.float32)
mat[:3, :3] *= 2.0 / 16
mat[:3, 3] = -1.0
real_kernarg = C.col_kernarg
for general in (False, True):
    # prospero's leaf tapes read no z: the kernel takes them as column-invariant (one voxel per pixel).  `general`: the kernarg of
    # FHIP_NO_COLUMN_INV (every input counts as varying along the column) - all 512 voxels, what a tape with z in it gets
    C.col_kernarg = (lambda *a, **kw: (lambda k: (k.__setitem__(4, 0xFFFFFFFF), k)[1])(real_kernarg(*a, **kw))) if general else real_kernarg
    for cls, leaves in sorted(by_class.items()):
        leaves.sort(key=lambda t: t[0])
        runs = []
        for n_ops, lregs, leaf in (leaves[0], leaves[-1]):
            # (the general path walks the table by groups of four layers of a footprint's column since round 6 - option column_group = 2 -,
            # the default path of a model without z by whole columns)
            zbuf, ws = C.run_columns(leaf, lregs, ik, mat.reshape(-1), (0, 0, 0), size=16, column_mode=True, group_log2=2 if general else 6)
            runs.append((n_ops, counts(max(ws, key=lambda w: w.counts.get('valu', 0)))))      # (the workgroup that finds the leaf)
        (n0, c0), (n1, c1) = runs
        e = {"leaf_tapes_ops": [n0, n1], "voxels_evaluated": 512 if general else 64, "workgroup_total": [c0, c1]}
        if n1 > n0:
            slope = {k: (c1.get(k, 0) - c0.get(k, 0)) / (n1 - n0) for k in c1}
            e["per_op (all passes over the tape)"] = {k: round(v, 2) for k, v in slope.items()}
            e["set_up_per_leaf"] = {k: round(c0.get(k, 0) - slope[k] * n0, 1) for k in c1}
        key = f"fh_columns: leaf tapes of <= {cls} registers, " + ("every voxel (general path)" if general else "column-invariant (one voxel per pixel)")
        res[key] = e
        print(key, e, flush=True)
C.col_kernarg = real_kernarg

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "instr_per_op.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
