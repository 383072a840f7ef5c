"""The linked prune's choice resolution and liveness pass (prune2.hip, phases A and B1) restated in Python from the same tables the
kernel gets - the tape's links (host_graph.hpp compute_links) and its root chain (capi_tapes.hpp chain_table) - against what
VmData::simplify (fidget-core/src/vm/data.rs:123-318) keeps: a reverse sweep over the register tape under the same choices.  No GPU:
the tables are host-side, and what is checked is the ALGORITHM the kernel runs - in particular the head start of the liveness queue
(the kept ops of the root chain found from their choices alone, without walking the chain), which must queue wanted ops only.  The
kernel itself is checked on the GPU by tests/test_prune2.py (values of its tapes on points of the tile)."""
import os

import numpy as np
import pytest

import emu_util as U
import fidget_amd as F
import test_emu_tiles as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LK_OUT, LK_NONE, LK_A, LK_RR, LK_COPY, LK_CRR, LK_CRI = range(7)
CHOICE, IMM = 0x8000, 0x4000
LEFT, RIGHT, BOTH = 1, 2, 3


def kept_by_simplify(tape, ch):
    """op indices a reverse sweep keeps (register liveness; a decided choice passes its register on; a reg,imm choice that took its
    immediate stays as a copy of it; register copies are looked through, as the links do)"""
    ops = [U.decode(w) for w in tape]
    live, kept = set(), set()
    q = sum(1 for o in ops if U.is_choice(o[0]))
    for i in range(len(ops) - 1, -1, -1):
        op, ro, ra, w1 = ops[i]
        name = U.OPS[op]
        c = None
        if U.is_choice(op):
            q -= 1
            c = int(ch[q])
        if name == "OUTPUT":
            kept.add(i); live.add(ra)
            continue
        if ro not in live:
            continue
        live.discard(ro)
        if name == "COPY_REG":
            live.add(ra)
            continue
        if c == LEFT:
            live.add(ra)
            continue
        if c == RIGHT:
            if U.is_rr(op):
                live.add(w1)
            else:
                kept.add(i)              # (stays, as COPY_IMM)
            continue
        kept.add(i)
        if name in ("INPUT", "COPY_IMM"):
            continue
        live.add(ra)
        if U.is_rr(op):
            live.add(w1)
    return kept


def kept_by_the_linked_prune(links, chain, ch, head_start=True):
    """phases A and B1 of prune2.hip: E[q] = the op choice q's value really is; then the work queue from the OUTPUT op, with the kept ops
    of the root chain queued first.  Returns (kept op indices, rounds of the queue, what the head start queued)"""
    n = len(links)
    ordinal_op = {int(l[2]): i for i, l in enumerate(links) if l[1] >= LK_CRR}
    E = {}
    for q in sorted(ordinal_op):                       # (tape order: a pointer's target has a lower ordinal and is final)
        i = ordinal_op[q]
        _, cls, _, fa, fb = (int(v) for v in links[i])
        c = int(ch[q])
        e = fa if c == LEFT else ((fb if cls == LK_CRR else (i | IMM)) if c == RIGHT else i)
        if e & CHOICE and e != 0xFFFF:
            e = E[e & 0x7FFF]
        E[q] = e

    def producers(i):
        _, cls, q, fa, fb = (int(v) for v in links[i])
        imm = cls == LK_CRI and (E[q] & IMM) != 0
        out = []
        if not imm and cls != LK_NONE:
            out.append(E[fa & 0x7FFF] & 0x3FFF if (fa & CHOICE) else fa)
        if not imm and cls in (LK_RR, LK_CRR):
            out.append(E[fb & 0x7FFF] & 0x3FFF if (fb & CHOICE) else fb)
        return out

    wanted, queue, seeded = {n - 1}, [n - 1], []
    if head_start:
        for q, i in chain[::-1]:                       # from the chain's end
            c = int(ch[q])
            if c == RIGHT:
                break                                  # the chain ends here: what lies below is dead from this side
            if c == BOTH:
                wanted.add(int(i)); queue.append(int(i)); seeded.append(int(i))
    head = rounds = 0
    while head < len(queue):
        batch = queue[head:head + 64]
        head += len(batch)
        rounds += 1
        for i in batch:
            for t in producers(i):
                assert t != 0xFFFF and t < n
                if t not in wanted:
                    wanted.add(t); queue.append(t)
    return wanted, rounds, seeded


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_liveness_with_the_chains_head_start_keeps_what_simplify_keeps(seed):
    sh = F.Shape.from_vm(os.path.join(ROOT, "models", "prospero.vm"))
    tape, links, chain = U.shape_tape(sh), sh.links(), sh.chain()
    assert links is not None and len(links) == len(tape)
    assert len(chain) > 600 and (np.diff(chain[:, 0]) > 0).all() and (np.diff(chain[:, 1]) > 0).all()      # evaluation order = tape order
    names = [U.OPS[U.decode(tape[i])[0]] for i in chain[:, 1]]
    assert set(names) <= {"MIN_RR", "MIN_RI"} and int(chain[-1, 1]) == len(tape) - 2        # ... all the way to the OUTPUT op
    ik = [3] * 16
    for a in range(3):
        s = sh.axis_index(a)
        if s >= 0:
            ik[s] = a
    rng = np.random.default_rng(seed)
    checked = 0
    for half in (0.125, 0.5):                       # children of 32^3 and of 128^3 voxels of a 1024^3 frame
        centre = tuple(rng.uniform(-0.7, 0.7, 3))
        xyz = T.children(centre, half)
        inputs = {s: (xyz[2 * k], xyz[2 * k + 1]) for s, k in enumerate(ik) if k < 3}
        el, eh, chs, _ = U.ref_interval(tape, inputs, 64)
        for lane in np.nonzero(~(eh < 0) & ~(el > 0))[0][:6]:
            ch = chs[:, lane]
            want = kept_by_simplify(tape, ch)
            got, rounds, seeded = kept_by_the_linked_prune(links, chain, ch)
            assert got == want, (len(got), len(want), sorted(got ^ want)[:8])
            assert set(seeded) <= want                                         # the head start queues wanted ops only
            plain, rounds_plain, _ = kept_by_the_linked_prune(links, chain, ch, head_start=False)
            assert plain == want and rounds <= rounds_plain
            checked += 1
    assert checked >= 6
