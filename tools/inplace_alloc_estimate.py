#!/usr/bin/env python3
"""What would the leaf kernel save if the prune's register allocation preferred the dying operand's register (out == a: the in-place
handlers, no copies through temporaries)?  No GPU and no change to the kernels: prospero's leaf tapes of one 32^3 parent are pruned by
the Python model of the sweep (tests/emu_util.py ref_prune) with the allocation as it is - the lowest free register - and with the
preference, and the committed fh_columns is run on both sets of tapes in the gfx950 emulator, instructions counted.
usage: tools/inplace_alloc_estimate.py [out.json]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import emu_util as U
import test_emu_tiles as T
import test_emu_columns as C

decode, pack, is_choice, is_rr = U.decode, U.pack, U.is_choice, U.is_rr


def prune(tape, choices, prefer_in_place):
    """ref_prune with a switch: an operand `a` that is not live yet takes the register its op's output has just given back"""
    DEAD = -1
    m, used, high = {}, set(), [0]

    def take():
        r = 0
        while r in used:
            r += 1
        used.add(r); high[0] = max(high[0], r + 1)
        return r

    def use(r):
        if m.get(r, DEAD) == DEAD:
            m[r] = take()
        return m[r]
    rev, ci = [], len(choices)
    for w in reversed(list(tape)):
        op, ro, ra, w1 = decode(w)
        c = 3
        if is_choice(op):
            ci -= 1; c = int(choices[ci])
        if op == 0:
            rev.append(pack(0, 0, use(ra), w1)); continue
        no = m.get(ro, DEAD)
        if no == DEAD:
            continue
        m[ro] = DEAD
        alias, copy_imm = None, False
        if op == 2 or (is_choice(op) and c == 1):
            alias = ra
        elif is_choice(op) and c == 2:
            if is_rr(op): alias = w1
            else: copy_imm = True
        if alias is not None:
            if m.get(alias, DEAD) == DEAD:
                m[alias] = no; continue
            used.discard(no); rev.append(pack(2, no, m[alias], 0)); continue
        if copy_imm:
            used.discard(no); rev.append(pack(3, no, 0, w1)); continue
        na = nb = 0
        if op not in (1, 3):
            if prefer_in_place == 2 and op in (22, 24) and m.get(ra, DEAD) != DEAD and m.get(w1, DEAD) == DEAD and w1 != ra:
                ra, w1 = w1, ra          # ADD_RR / MUL_RR commute bit for bit (but for which NaN's payload survives): the dying operand first
            if prefer_in_place and m.get(ra, DEAD) == DEAD:
                m[ra] = no               # (the register stays in use: it changes hands)
                na = no
            else:
                used.discard(no)
                na = use(ra)
        else:
            used.discard(no)
        if is_rr(op):
            nb = use(w1)
        rev.append(pack(op, no, na, nb if is_rr(op) else w1))
    return rev[::-1], high[0]


ik, ch = T.chain()
tape2, regs2, nch2, center2, half2 = ch[1]
xyz = T.children(center2, half2)
inputs = {s: (xyz[2 * k], xyz[2 * k + 1]) for s, k in enumerate(ik) if k < 3}
el, eh, chs, _ = U.ref_interval(tape2, inputs, 64)
lanes = np.nonzero(~(eh < 0) & ~(el > 0))[0]
mat = np.eye(4, dtype=np.float32); mat[:3, :3] *= 2.0 / 16; mat[:3, 3] = -1.0
real_kernarg = C.col_kernarg
C.col_kernarg = lambda *a, **kw: (lambda k: (k.__setitem__(4, 0xFFFFFFFF), k)[1])(real_kernarg(*a, **kw))     # the general path: every voxel
res = {}
for name, pref in (("lowest free register (as built)", 0), ("the dying operand takes the output's register", 1),
                   ("... and add / mul put their dying operand first", 2)):
    tot = {"ops": 0, "in_place": 0, "binary_or_unary": 0, "valu": 0, "salu": 0, "total": 0, "leaves": 0, "regs": []}
    zs = []
    for lane in lanes:
        ops, lregs = prune(tape2, chs[:, lane], pref)
        if not pref:
            assert ops == U.ref_prune(tape2, chs[:, lane])[0]
        if lregs > 8:
            continue                      # (the 8-voxel class only: 90 % of prospero's leaves)
        for w in ops:
            op, ro, ra, w1 = decode(w)
            if op > 3:
                tot["binary_or_unary"] += 1
                tot["in_place"] += int(ro == ra)
        zbuf, ws = C.run_columns(np.array(ops, np.uint64), lregs, ik, mat.reshape(-1), (0, 0, 0), size=16)
        w = max(ws, key=lambda w: w.counts.get("valu", 0))
        tot["ops"] += len(ops); tot["leaves"] += 1; tot["regs"].append(lregs)
        tot["valu"] += w.counts.get("valu", 0); tot["salu"] += w.counts.get("salu", 0); tot["total"] += int(w.n_inst)
        zs.append(np.array(zbuf).copy())
    res[name] = {"leaves": tot["leaves"], "ops": tot["ops"], "in_place_fraction_of_computing_ops": round(tot["in_place"] / max(tot["binary_or_unary"], 1), 3),
                 "instructions_per_leaf": round(tot["total"] / tot["leaves"], 1), "valu_per_leaf": round(tot["valu"] / tot["leaves"], 1), "salu_per_leaf": round(tot["salu"] / tot["leaves"], 1),
                 "mean_registers": round(float(np.mean(tot["regs"])), 2)}
    res[name]["_z"] = zs
    print(name, {k: v for k, v in res[name].items() if k != "_z"}, flush=True)
a, b, c = (res[k] for k in list(res))
za, zb, zc = a.pop("_z"), b.pop("_z"), c.pop("_z")
res["same_depths_from_all_sets_of_tapes"] = bool(all((x == y).all() and (x == z).all() for x, y, z in zip(za, zb, zc)))
res["instructions_saved"] = [round(1 - v["instructions_per_leaf"] / a["instructions_per_leaf"], 4) for v in (b, c)]
res["valu_saved"] = [round(1 - v["valu_per_leaf"] / a["valu_per_leaf"], 4) for v in (b, c)]
print({k: v for k, v in res.items() if not isinstance(v, dict)})
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "inplace_alloc_estimate.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
