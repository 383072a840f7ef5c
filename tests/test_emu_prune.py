"""fh_prune1 - the prune sweep of one child tile per wavefront in scalar code (gen_prune.py) - executed instruction by
instruction on the CPU emulator (tools/gfx950_emu.py) against the numpy restatement of VmData::simplify's reverse sweep
(tests/emu_util.py ref_prune, following kernels.hip prune_sweep, i.e. fidget-core/src/vm/data.rs:123-318): every marked
child's tape, register count and kept-choice count, bit for bit; unmarked children untouched.  No GPU needed."""
import numpy as np
import pytest

import emu_util as U
from emu_util import E, F32, U32
from conftest import model_path
from test_emu_tiles import children, shape_of

ARENA_OPS = 1 << 17
COFF, CLEN, CRC = 40 + 11 * 256, 40 + 12 * 256, 40 + 13 * 256      # FhSlot: c_off[64], c_len[64], c_rc[64] (gen_tiles SL_*)


def run_prune1(tape, choices, marked, n_regs, level=1, big=0, mode=1):
    """choices: [n_choices, 64] (1 Left, 2 Right, 3 Both) as fh_tiles' export mode leaves them in S->chw; marked: lanes to prune.
    Every marked child owns an arena slot of len(tape) ops, the kernel writes its tape to the END of it."""
    off = U.offsets()
    mem = E.Memory()
    n, nch = len(tape), len(choices)
    arena = np.zeros(ARENA_OPS, np.uint64)
    arena[16:16 + n] = tape
    a_arena = mem.map(arena, "arena")
    slot = U.Blob(off["sizeof_slot"])
    slot.u32(0, 16); slot.u32(4, n); slot.u32(8, n_regs | (nch << 16)); slot.u32(12, 2)
    slot.u64(16, (1 << 64) - 1)
    coff = np.full(64, 16, U32)
    clen = np.full(64, n, U32)
    crc = np.full(64, n_regs | (nch << 16), U32)
    head = 16 + n + 16
    ends = {}
    for lane in marked:
        head += n
        ends[lane] = head
        coff[lane], clen[lane] = head, 0xFFFFFFFF
    slot.arr(COFF, coff); slot.arr(CLEN, clen); slot.arr(CRC, crc)
    a_slot = mem.map(slot.b, "slot")
    max_ch = max(nch, 16)
    words = (max_ch + 15) // 16
    chw = np.zeros((words, 64), U32)
    for ci in range(nch):
        chw[ci >> 4] |= (np.asarray(choices[ci], U32) & 3) << (2 * (ci & 15))
    a_chw = mem.map(chw.reshape(-1), "chw")
    st = U.Blob(off["sizeof_state"])
    st.u64(off["arena"], a_arena); st.u32(off["arena_cap"], ARENA_OPS - 64); st.u32(off["arena_head"], head)
    st.u64(off["slots"] + 8 * big, a_slot)
    if mode == 2:       # tape groups (level 0): the choice words come from S->chwr (k_tscatter3d), the slot is block * n_tgroups
        st.u64(off["chwr"], a_chw); st.u32(off["n_tgroups"], 1)
    else:
        st.u64(off["chw"] + 8 * big, a_chw)
    st.u32(off["n_slots"] + 4 * (big * 8 + level), 1)
    a_st = mem.map(st.b, "state")
    ka = np.zeros(6, U32)
    ka[0], ka[1] = a_st & 0xFFFFFFFF, a_st >> 32
    ka[2:6] = [level, big, max_ch, mode]
    E.launch(U.program(), mem, "fh_prune1", ka.tobytes(), 64, lds_bytes=16, n_vgpr=24)
    return dict(arena=arena, coff=slot.b[COFF:COFF + 256].view(U32).copy(), clen=slot.b[CLEN:CLEN + 256].view(U32).copy(),
                crc=slot.b[CRC:CRC + 256].view(U32).copy(), ends=ends)


def check(tape, xyz, in_kind, n_regs, only=None, **kw):
    inputs = {s: (xyz[2 * k], xyz[2 * k + 1]) for s, k in enumerate(in_kind) if k < 3}
    el, eh, ch, _ = U.ref_interval(tape, inputs, 64)
    amb = ~(eh < 0) & ~(el > 0)
    decided = (ch != 3).any(axis=0) if len(ch) else np.zeros(64, bool)
    marked = [int(k) for k in np.nonzero(amb & decided)[0]]
    if only is not None:
        marked = marked[:only]
    r = run_prune1(tape, ch, marked, n_regs, **kw)
    for lane in range(64):
        if lane in marked:
            ops, regs, kept = U.ref_prune(tape, ch[:, lane])
            assert r["clen"][lane] == len(ops), (lane, r["clen"][lane], len(ops))
            assert r["coff"][lane] == r["ends"][lane] - len(ops)
            got = r["arena"][r["coff"][lane]: r["coff"][lane] + len(ops)]
            assert (got == np.array(ops, np.uint64)).all(), f"lane {lane}: pruned tape differs"
            assert r["crc"][lane] == (regs | (kept << 16)), f"lane {lane}: regs / choices {r['crc'][lane]:#x} vs {regs}, {kept}"
        else:
            assert r["coff"][lane] == 16 and r["clen"][lane] == len(tape)
    return len(marked)


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 7, 8, 9])
def test_random_shapes(seed):
    sh, tape, ik = shape_of(seed)
    n_regs, n_choices = sh.slot_count(), sh.choice_count()
    if n_regs > 128:
        pytest.skip("fh_prune1 keeps its register map in two VGPRs: 128 registers")
    rng = np.random.default_rng(seed)
    n = 0
    for k in range(8):          # (boxes until two of them had children to prune)
        c = rng.uniform(-0.8, 0.8, 3)
        n += check(tape, children(c, rng.uniform(0.1, 0.5)), ik, n_regs, only=6) > 0
        if n == 2:
            break
    assert n > 0 or n_choices == 0


def test_prospero_root_tile_children():
    """the root tape of prospero.vm (6363 ops, more than 64 registers, 2878 choices: both halves of the register map, many batches of choice words)
    pruned for children of a root tile"""
    import fidget_amd as F
    sh = F.Shape.from_vm(model_path("prospero.vm"))
    tape, n_regs = U.shape_tape(sh), sh.slot_count()
    ik = [3] * 16
    for a in range(3):
        s = sh.axis_index(a)
        if s >= 0:
            ik[s] = a
    assert len(tape) > 6000 and 64 < n_regs <= 128
    assert check(tape, children((0.125, 0.125, 0.125), 0.125), ik, n_regs, only=3) == 3


@pytest.mark.parametrize("big,mode", [(1, 1), (0, 2)])
def test_big_list_and_tape_group_mode(big, mode):
    """the second slot list (big = 1) and the level-0 tape-group mode (choice words from S->chwr)"""
    sh, tape, ik = shape_of(2)
    assert check(tape, children((0.1, -0.2, 0.0), 0.4), ik, sh.slot_count(), only=4, big=big, mode=mode, level=0 if mode == 2 else 1) > 0


def run_export_then_prune(tape, xyz, in_kind, n_regs, n_choices, level=1):
    """The pre-pass pair as capi.hip launches it for a level with long tapes: fh_tiles in export mode (flags bit 1: forward pass,
    classification, choice words to S->chw, the children to prune only MARKED: c_len = ~0, c_off = end of their arena slot),
    then fh_prune1, one wave per child, on the same state."""
    from test_emu_tiles import ARENA_OPS as A_OPS
    off = U.offsets()
    mem = E.Memory()
    n = len(tape)
    arena = np.zeros(A_OPS, np.uint64)
    arena[16:16 + n] = tape
    a_arena = mem.map(arena, "arena")
    st, slot = U.Blob(off["sizeof_state"]), U.Blob(off["sizeof_slot"])
    slot.u32(0, 16); slot.u32(4, n); slot.u32(8, n_regs | (n_choices << 16)); slot.u32(12, 2)
    slot.u64(16, (1 << 64) - 1)
    for k in range(6):
        slot.arr(40 + 256 * k, np.asarray(xyz[k], F32))
    a_slot = mem.map(slot.b, "slot")
    head0 = 16 + n + 16
    mr, mc = max(n_regs, 32), max(n_choices, 256)
    words = (mc + 15) // 16
    chw = np.zeros(words * 64, U32)
    a_chw = mem.map(chw, "chw")
    st.u64(off["arena"], a_arena); st.u32(off["arena_cap"], A_OPS - 64); st.u32(off["arena_head"], head0)
    st.u64(off["slots"], a_slot); st.u64(off["slots"] + 8, a_slot); st.u64(off["chw"], a_chw); st.u64(off["chw"] + 8, a_chw)
    st.u32(off["slot_cap"], 1); st.u32(off["slot_cap"] + 4, 1)
    for big in (0, 1):
        st.u32(off["n_slots"] + 4 * (big * 8 + level), 1)
    for s in range(16):
        st.u32(off["P.in_kind"] + 4 * s, in_kind[s] if s < len(in_kind) else 3)
    a_st = mem.map(st.b, "state")
    ka = np.zeros(10, U32)
    ka[0], ka[1] = a_st & 0xFFFFFFFF, a_st >> 32
    ka[2:10] = [level, 0, mr, mc, 1, 2 | (words << 16), 0, 0]
    lds = mr * 512 + words * 256 + mr * 64 + 256
    E.launch(U.program(), mem, "fh_tiles", ka.tobytes(), 1, lds_bytes=lds, n_vgpr=84)
    g = lambda k: slot.b[40 + k * 256: 40 + (k + 1) * 256]
    marked = np.nonzero(g(12).view(U32) == 0xFFFFFFFF)[0]
    ends = g(11).view(U32).copy()
    head = int(st.get_u32(off["arena_head"])[0])
    kp = np.zeros(6, U32)
    kp[0], kp[1] = a_st & 0xFFFFFFFF, a_st >> 32
    kp[2:6] = [level, 0, words * 16, 1]
    E.launch(U.program(), mem, "fh_prune1", kp.tobytes(), 64, lds_bytes=16, n_vgpr=24)
    return dict(res=(g(9).view(F32).copy(), g(10).view(F32).copy()), coff=g(11).view(U32).copy(), clen=g(12).view(U32).copy(),
                crc=g(13).view(U32).copy(), arena=arena, marked=marked, ends=ends, head=head, head0=head0)


@pytest.mark.parametrize("seed", [0, 2, 3])
def test_export_mode_then_prune1(seed):
    """fh_tiles' export mode and fh_prune1 agree about where the choice words and the marks live: together they give the
    interval results and the child tapes of the one-kernel path (and of the numpy restatement)"""
    sh, tape, ik = shape_of(seed)
    n_regs, n_choices = sh.slot_count(), sh.choice_count()
    if n_regs > 128:
        pytest.skip("fh_prune1: 128 registers")
    rng = np.random.default_rng(100 + seed)
    done = 0
    for _ in range(6):
        xyz = children(rng.uniform(-0.7, 0.7, 3), rng.uniform(0.15, 0.5))
        r = run_export_then_prune(tape, xyz, ik, n_regs, n_choices)
        inputs = {s: (xyz[2 * k], xyz[2 * k + 1]) for s, k in enumerate(ik) if k < 3}
        el, eh, ch, _ = U.ref_interval(tape, inputs, 64)
        assert (el.view(U32) == r["res"][0].view(U32)).all() and (eh.view(U32) == r["res"][1].view(U32)).all()
        amb = ~(eh < 0) & ~(el > 0)
        decided = (ch != 3).any(axis=0) if len(ch) else np.zeros(64, bool)
        want = np.nonzero(amb & decided)[0]
        assert list(r["marked"]) == list(want)
        assert r["head"] == r["head0"] + len(want) * len(tape)
        for lane in want:
            ops, regs, kept = U.ref_prune(tape, ch[:, lane])
            assert r["clen"][lane] == len(ops) and r["coff"][lane] == r["ends"][lane] - len(ops)
            got = r["arena"][r["coff"][lane]: r["coff"][lane] + len(ops)]
            assert (got == np.array(ops, np.uint64)).all(), f"lane {lane}"
            assert r["crc"][lane] == (regs | (kept << 16))
        done += len(want) > 0
        if done == 2:
            break
    assert done > 0 or n_choices == 0
