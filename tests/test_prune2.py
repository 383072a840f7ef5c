"""The linked prune (fidget_amd/csrc/prune2.hip k_prune2: one wave per child of the root level, visiting only the ops the
child keeps, through the per-op links of host_graph.hpp compute_links) - VmData::simplify (fidget-core/src/vm/data.rs:123-318)
restricted to the live part of the tape's dependency graph.  Option prune2 (on by default; the assembly sweep fh_prune1 runs
behind it for the children it leaves marked, and alone with prune2 = 0).  Its tapes differ from the scalar sweep's (fh_prune1) in register
numbers and in the copies that sweep inserts; what must hold is what simplify promises: on its tile, a child tape computes the
parent tape's value, bit for bit - and the frames are the same images."""
import os

import numpy as np
import pytest

from conftest import model_path

import fidget_amd as F


def links_ref(ops):
    """numpy restatement of compute_links: (opcode, class, choice ordinal, fa, fb) per op"""
    last, field, ci, out = {}, {}, 0, []
    for i, w in enumerate(ops):
        w0, w1 = int(w) & 0xFFFFFFFF, int(w) >> 32
        op, ro, ra = w0 & 0xFF, (w0 >> 8) & 0xFFF, w0 >> 20
        rr, choice = 22 <= op <= 33, (30 <= op <= 33) or (42 <= op <= 45)
        kind = 0 if op == 0 else 1 if op in (1, 3) else 4 if op == 2 else (5 if rr else 6) if choice else 3 if rr else 2
        fa = field[last[ra]] if kind != 1 else 0xFFFF
        fb = field[last[w1]] if rr else 0xFFFF
        out.append((op, kind, ci, fa, fb))
        field[i] = fa if op == 2 else (0x8000 | ci) if choice else i
        ci += choice
        if op != 0:
            last[ro] = i
    return out


@pytest.mark.parametrize("name", ["prospero.vm", "colonnade.vm", "bear.vm", "hi.vm"])
def test_links_of_a_tape_name_the_producer_of_every_operand(name):
    """host side, no GPU: every link of a root tape names an earlier op that writes the operand's register (through register
    copies), with no other writer of that register in between; a choice op is named by its ordinal; the library's links are the
    restatement's"""
    s = F.Shape.from_vm(model_path(name))
    ops = s.words()
    lk = links_ref(ops)
    assert len(lk) == len(ops) and lk[-1][1] == 0
    if name == "prospero.vm":
        assert len(ops) == 6363
    outs = [(int(w) >> 8) & 0xFFF for w in ops]
    opc = [int(w) & 0xFF for w in ops]
    choice_ops = [i for i, o in enumerate(opc) if (30 <= o <= 33) or (42 <= o <= 45)]
    for i, (op, kind, ci, fa, fb) in enumerate(lk):
        w0, w1 = int(ops[i]) & 0xFFFFFFFF, int(ops[i]) >> 32
        for f, r in ((fa, w0 >> 20), (fb, w1)):
            if f == 0xFFFF:
                continue
            p = choice_ops[f & 0x7FFF] if f & 0x8000 else f
            assert p < i
            if opc[p] != 2 and not any(opc[k] == 2 for k in range(p, i)):       # (no copy on the way: the producer writes the operand's register itself)
                assert outs[p] == r and r not in outs[p + 1:i]
        assert ci == sum(1 for k in choice_ops if k < i)
    got = s.links()
    assert got is not None and (got == np.array(lk, np.int64)).all()


def _children(hip, shape, size, tile):
    """the child tapes the root level's prune left: 128^3 root tiles -> the queue of level 1; root tiles of 32^3 (one coarse level) -> the
    parents parked per z-slab"""
    img = F.render3d(shape, size, tile_sizes=[128, 32, 8] if tile == 128 else [32, 8])[0]
    tapes = {}
    lists = [hip.groups(0, 1)[0]] if tile == 128 else [hip.groups(1, k)[0] for k in range(8)]
    for g in lists:
        for e in g:
            tapes[(int(e["x"]), int(e["y"]), int(e["z"]))] = (hip.arena_ops(int(e["off"]), int(e["len"])), int(e["regs"]), int(e["choices"]))
    return img, tapes


@pytest.mark.gpu
@pytest.mark.parametrize("name,size,tile", [("prospero.vm", 512, 128), ("prospero.vm", 1024, 128), ("colonnade.vm", 512, 128), ("prospero.vm", 512, 32),
                                            ("prospero.vm", 1024, 32)])
def test_linked_prune_children_compute_the_root_tape_on_their_tile(name, size, tile):
    """simplify's contract, checked directly: every child tape the root level's prune wrote, evaluated (numpy f32, the emulator
    tests' restatement of the device ops) at random points of its tile - 128^3, or 32^3 when the root level stands straight above the
    leaves (round 5) - gives the root tape's value bit for bit; the frame is the scalar sweep's frame; and the linked prune really ran
    and pruned."""
    import emu_util as U
    hip = F.HipContext(0)
    s = F.Shape.from_vm(model_path(name), hip=hip)
    info = np.zeros(4, np.uint32)
    if F.lib().fhip_tape_term_plan(s._h, F._p(info)) == 0:
        pytest.skip("this tape is not split at its root (no term plan): the root level takes the other path")
    with hip.options(prune2=0):
        img_a, a = _children(hip, s, size, tile)           # the scalar sweep alone
    with hip.options(prune2=1):
        img_b, b = _children(hip, s, size, tile)
    assert (img_a["depth"] == img_b["depth"]).all() and (img_a["normal"].view(np.uint32) == img_b["normal"].view(np.uint32)).all()
    assert len(b) > 8 and a.keys() == b.keys()
    root = s.words()
    ik = [s.axis_index(ax) for ax in range(3)]
    mat = np.zeros(16, np.float32)
    F.lib().fhip_screen_to_world(F._p(np.array([size, size, size], np.uint32)), 3, F._p(mat))      # voxel -> model (identity camera)
    mat = mat.reshape(4, 4).astype(np.float64)
    rng = np.random.default_rng(7)
    shorter = fallback = 0
    for k in sorted(b)[:: max(1, len(b) // 40)]:            # a sample of the children (each costs two passes over the root tape in numpy)
        tape, regs, choices = b[k]
        outs = (tape.astype(np.uint64) >> np.uint64(8)) & np.uint64(0xFFF)
        assert int(outs.max()) < regs
        if regs > 64 or len(tape) > 1280:                  # (prune2.hip FH_P2_MAX_KEPT)
            # beyond the linked prune's limits: left to the scalar sweep launched behind it - its tape, word for word
            assert tape.tobytes() == a[k][0].tobytes()
            fallback += 1
        else:
            assert len(tape) <= len(a[k][0])                   # never longer than the scalar sweep's (which inserts copies)
            shorter += len(tape) < len(a[k][0])
            assert int((tape & np.uint64(0xFF)).tolist().count(2)) == 0      # no COPY_REG
        m = 256
        vox = np.stack([np.float64(k[ax]) + rng.random(m) * float(tile) for ax in range(3)] + [np.ones(m)])      # points of the tile, in voxels
        pts = (mat @ vox)[:3].astype(np.float32)
        inputs = {ik[ax]: pts[ax] for ax in range(3) if ik[ax] >= 0}
        want = U.ref_f32(root, inputs, m)
        got = U.ref_f32(tape, inputs, m)
        assert (want[0].view(np.uint32) == got[0].view(np.uint32)).all(), f"child {k}: values differ from the root tape's"
    print("children shorter than the scalar sweep's:", shorter, "left to the scalar sweep:", fallback)
    assert fallback < len(b) or size < 1024


