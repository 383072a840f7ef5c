"""The assembly tile kernels (fh_tiles, fh_tiles_v32, fh_tiles_v64) executed instruction by instruction on the CPU
emulator (tools/gfx950_emu.py: the assembled code object, hazard checks on) against the numpy restatement of the
interval evaluator and of the prune sweep (tests/emu_util.py, following dev_ops.hpp / kernels.hip prune_sweep, i.e.
fidget-core/src/vm/mod.rs:325-538 and vm/data.rs:123-318): interval results, and every pruned child tape, bit for bit.

No GPU needed: this is how the assembly is developed and regression-tested in the build container."""
import os

import numpy as np
import pytest

import emu_util as U
from emu_util import E, F32, U32
from conftest import model_path
from test_render_random import build

N_VGPR = {"fh_tiles": 84, "fh_tiles_v32": 128, "fh_tiles_v64": 208, "fh_tiles_t": 110, "fh_tiles_v32_t": 154, "fh_tiles_v64_t": 234}
LIMITS = {"fh_tiles": (128, 4096), "fh_tiles_v32": (32, 256), "fh_tiles_v64": (64, 512), "fh_tiles_t": (128, 4096), "fh_tiles_v32_t": (32, 256),
          "fh_tiles_v64_t": (64, 512)}
# the *_t kernels (handlers for the transcendental opcodes): where the compiled routines sit in each (gen_tiles.py / gen_tilesv.py)
T_HOOKS = {"fh_tiles_t": ("fh_til_", 84), "fh_tiles_v32_t": ("fh_ti32_", 128), "fh_tiles_v64_t": ("fh_ti64_", 208)}
T_SREGS = (78, 79, 80, 81, 82, 83, 84, 85, 88, 89)
ARENA_OPS = 1 << 17


def run_tiles(kernel, tape, xyz, in_kind, n_regs, n_choices, act=(1 << 64) - 1, arena_cap=None, in_value=None, skip=(0, 0), limits=None):
    """One FhSlot through `kernel`.  Returns dict(res, coff, clen, crc, arena, wave, head, overflow)."""
    off = U.offsets()
    mem = E.Memory()
    n = len(tape)
    arena = np.zeros(ARENA_OPS, np.uint64)
    arena[16:16 + n] = tape
    a_arena = mem.map(arena, "arena")
    st, slot = U.Blob(off["sizeof_state"]), U.Blob(off["sizeof_slot"])
    slot.u32(0, 16); slot.u32(4, n); slot.u32(8, n_regs | (n_choices << 16)); slot.u32(12, 2)
    slot.u64(16, act)
    for k in range(6):
        slot.arr(40 + 256 * k, np.asarray(xyz[k], F32))
    a_slot = mem.map(slot.b, "slot")
    head0 = 16 + n + 16
    st.u64(off["arena"], a_arena); st.u32(off["arena_cap"], arena_cap if arena_cap is not None else ARENA_OPS - 64); st.u32(off["arena_head"], head0)
    st.u64(off["slots"], a_slot); st.u64(off["slots"] + 8, a_slot)
    st.u32(off["slot_cap"], 1); st.u32(off["slot_cap"] + 4, 1)
    level = 2
    for big in (0, 1):
        st.u32(off["n_slots"] + 4 * (big * 8 + level), 1)
    for s in range(16):
        st.u32(off["P.in_kind"] + 4 * s, in_kind[s] if s < len(in_kind) else 3)
        if in_value is not None and s < len(in_value):
            st.f32(off["P.in_value"] + 4 * s, in_value[s])
    a_st = mem.map(st.b, "state")
    mr, mc = limits or LIMITS[kernel]
    if kernel in ("fh_tiles", "fh_tiles_t"):
        mr, mc = max(n_regs, 32), max(n_choices, 256)
    ka = np.zeros(10, U32)
    ka[0], ka[1] = a_st & 0xFFFFFFFF, a_st >> 32
    ka[2:10] = [level, 0, mr, mc, 1, 0, skip[0], skip[1]]
    lds = mr * 512 + ((mc + 15) // 16) * 256 + mr * 64 + 256 if kernel in ("fh_tiles", "fh_tiles_t") else 0
    hooks = U.trans_hooks(U.program(), T_HOOKS[kernel][0], T_HOOKS[kernel][1], T_SREGS) if kernel in T_HOOKS else None
    w = E.launch(U.program(), mem, kernel, ka.tobytes(), 1, lds_bytes=max(lds, 16), n_vgpr=N_VGPR[kernel], hooks=hooks)[0]
    g = lambda k: slot.b[40 + k * 256: 40 + (k + 1) * 256]
    return dict(res=(g(9).view(F32).copy(), g(10).view(F32).copy()), coff=g(11).view(U32).copy(), clen=g(12).view(U32).copy(),
                crc=g(13).view(U32).copy(), arena=arena, wave=w, head=int(st.get_u32(off["arena_head"])[0]), head0=head0,
                overflow=int(st.get_u32(off["arena_overflow"])[0]))


def children(center, half, n=4):
    """x.lo x.hi y.lo y.hi z.lo z.hi of the n^3 (n = 4: 64) children of the cube center +- half, lane = x + n*y + n*n*z"""
    lanes = np.arange(64)
    idx = (lanes % n, (lanes // n) % n, lanes // (n * n))
    step = 2.0 * half / n
    out = []
    for ax in range(3):
        lo = (center[ax] - half + idx[ax] * step).astype(F32)
        out += [lo, (lo + F32(step)).astype(F32)]
    return out


def check_slot(kernel, tape, xyz, in_kind, n_regs, n_choices, act=(1 << 64) - 1, **kw):
    r = run_tiles(kernel, tape, xyz, in_kind, n_regs, n_choices, act=act, **kw)
    inputs = {s: (xyz[2 * k], xyz[2 * k + 1]) for s, k in enumerate(in_kind) if k < 3}
    el, eh, ch, _ = U.ref_interval(tape, inputs, 64)
    rl, rh = r["res"]
    # (bit for bit; a NaN equals any NaN - its sign and payload are not part of the contract, DESIGN.md section 2: a NaN that has gone
    # through a subtraction's negated operand carries the other sign on the device)
    same = lambda e, g: ((e.view(U32) == g.view(U32)) | (np.isnan(e) & np.isnan(g))).all()
    assert same(el, rl) and same(eh, rh), "interval results differ"
    actm = np.array([(act >> i) & 1 for i in range(64)], bool)
    amb = actm & ~(eh < 0) & ~(el > 0)
    decided = (ch != 3).any(axis=0) if len(ch) else np.zeros(64, bool)
    pruned = amb & decided
    kids = {}
    for lane in range(64):
        if pruned[lane]:
            ops, regs, kept = U.ref_prune(tape, ch[:, lane])
            got = r["arena"][r["coff"][lane]: r["coff"][lane] + r["clen"][lane]]
            assert r["clen"][lane] == len(ops) and (got == np.array(ops, np.uint64)).all(), f"lane {lane}: pruned tape differs"
            assert r["crc"][lane] == (regs | (kept << 16)), f"lane {lane}: regs / choices {r['crc'][lane]:#x} vs {regs}, {kept}"
            assert r["coff"][lane] + r["clen"][lane] <= r["head"] and r["coff"][lane] >= r["head0"]
            kids[lane] = (np.array(ops, np.uint64), regs, kept)
        else:
            assert r["coff"][lane] == 16 and r["clen"][lane] == len(tape) and r["crc"][lane] == (n_regs | (n_choices << 16))
    assert r["head"] == r["head0"] + int(pruned.sum()) * len(tape)
    return r, kids, (el, eh)


def shape_of(seed):
    import fidget_amd as F
    ctx = F.Context()
    sh = F.Shape(ctx, build(ctx, seed))
    ik = [3] * 16
    for a in range(3):
        s = sh.axis_index(a)
        if s >= 0:
            ik[s] = a
    return sh, U.shape_tape(sh), ik


@pytest.mark.parametrize("kernel", ["fh_tiles", "fh_tiles_v32", "fh_tiles_v64"])
@pytest.mark.parametrize("seed", range(8))
def test_random_shapes(kernel, seed):
    sh, tape, ik = shape_of(seed)
    if sh.slot_count() > LIMITS[kernel][0] or sh.choice_count() > LIMITS[kernel][1]:
        pytest.skip("tape outside this kernel's register file")
    rng = np.random.default_rng(seed)
    total = 0
    for _ in range(2):
        c, h = rng.uniform(-0.7, 0.7, 3), rng.uniform(0.1, 0.6)
        _, kids, _ = check_slot(kernel, tape, children(c, h), ik, sh.slot_count(), sh.choice_count())
        total += len(kids)
    assert total >= 0


def prospero_chain():
    """root tape -> a level-1 parent tape (128^3 tile of a 1024^3 render) -> a level-2 parent tape (32^3 tile), by the reference"""
    import fidget_amd as F
    sh = F.Shape.from_vm(model_path("prospero.vm"))
    tape = U.shape_tape(sh)
    ik = [3] * 16
    for a in range(3):
        ik[sh.axis_index(a)] = a
    # the 8^3 = 512 root tiles of 128^3 voxels in [-1, 1]^3: pick an ambiguous one with a long pruned tape
    out = []
    cur, regs, nch = tape, sh.slot_count(), sh.choice_count()
    center, half = np.zeros(3), 1.0
    for level in range(2):
        n = 8 if level == 0 else 4
        best = None
        if level == 0:     # 8 x 8 x 8 children: evaluate them 64 at a time (z layers)
            cands = []
            for zl in range(8):
                lanes = np.arange(64)
                xs, ys = lanes % 8, lanes // 8
                xyz = []
                for idx in (xs, ys, np.full(64, zl)):
                    lo = (-1.0 + idx * 0.25).astype(F32)
                    xyz += [lo, (lo + F32(0.25)).astype(F32)]
                inputs = {s: (xyz[2 * k], xyz[2 * k + 1]) for s, k in enumerate(ik) if k < 3}
                el, eh, ch, _ = U.ref_interval(cur, inputs, 64)
                amb = ~(eh < 0) & ~(el > 0)
                for lane in np.nonzero(amb)[0]:
                    cands.append((zl, lane, ch[:, lane].copy(), [float(xyz[2 * a][lane]) for a in range(3)]))
            # the child with the longest pruned tape
            for zl, lane, c, lo in cands:
                ops, r, k = U.ref_prune(cur, c)
                if best is None or len(ops) > len(best[0]):
                    best = (ops, r, k, lo, 0.125)
        else:
            xyz = children(center, half)
            inputs = {s: (xyz[2 * k], xyz[2 * k + 1]) for s, k in enumerate(ik) if k < 3}
            el, eh, ch, _ = U.ref_interval(cur, inputs, 64)
            amb = ~(eh < 0) & ~(el > 0)
            for lane in np.nonzero(amb)[0]:
                ops, r, k = U.ref_prune(cur, ch[:, lane])
                if best is None or len(ops) > len(best[0]):
                    best = (ops, r, k, [float(xyz[2 * a][lane]) for a in range(3)], half / 4)
        ops, r, k, lo, h = best
        cur, regs, nch = np.array(ops, np.uint64), r, k
        center, half = np.array(lo) + h, h
        out.append((cur, regs, nch, center.copy(), half))
    return ik, out


_chain = None


def chain():
    global _chain
    if _chain is None:
        _chain = prospero_chain()
    return _chain


def test_prospero_level1_parent_v64():
    """a level-1 parent of prospero.vm (hundreds of ops, > 32 registers) through the 64-register kernel, and through fh_tiles"""
    ik, ch = chain()
    tape, regs, nch, center, half = ch[0]
    assert 32 < regs <= 64 and nch <= 512 and len(tape) > 300
    r, kids, _ = check_slot("fh_tiles_v64", tape, children(center, half), ik, regs, nch)
    assert len(kids) > 8
    # fh_tiles_v32 leaves the slot alone (tape outside its register file)
    r32 = run_tiles("fh_tiles_v32", tape, children(center, half), ik, regs, nch)
    assert (r32["clen"] == 0).all() and r32["head"] == r32["head0"]


def test_prospero_level2_parent_v32():
    ik, ch = chain()
    tape, regs, nch, center, half = ch[1]
    assert regs <= 32 and nch <= 256 and len(tape) > 64
    for kernel in ("fh_tiles_v32", "fh_tiles_v64", "fh_tiles"):
        r, kids, _ = check_slot(kernel, tape, children(center, half), ik, regs, nch)
        assert len(kids) > 4
    # static cost of the two designs on the same parent (instructions per tape op of the whole slot)
    a = run_tiles("fh_tiles", tape, children(center, half), ik, regs, nch)["wave"]
    b = run_tiles("fh_tiles_v32", tape, children(center, half), ik, regs, nch)["wave"]
    assert b.counts.get("lds", 0) == 0 and b.n_inst < 0.7 * a.n_inst


def test_partial_activity_and_skip_rules():
    sh, tape, ik = shape_of(2)
    xyz = children(np.array([0.1, -0.2, 0.0]), 0.5)
    act = 0x0F0F_F0F0_1234_5678
    check_slot("fh_tiles_v32", tape, xyz, ik, sh.slot_count(), sh.choice_count(), act=act)
    # a slot that fits the (skip_regs, skip_choices) layout is left to the other launch
    r = run_tiles("fh_tiles_v64", tape, xyz, ik, sh.slot_count(), sh.choice_count(), skip=(32, 256))
    assert (r["clen"] == 0).all() and r["head"] == r["head0"]
    # inactive slot
    r = run_tiles("fh_tiles_v32", tape, xyz, ik, sh.slot_count(), sh.choice_count(), act=0)
    assert (r["clen"] == 0).all()


def test_arena_overflow_keeps_parent_tape():
    sh, tape, ik = shape_of(0)
    xyz = children(np.array([0.0, 0.0, 0.0]), 0.5)
    r = run_tiles("fh_tiles_v32", tape, xyz, ik, sh.slot_count(), sh.choice_count(), arena_cap=16 + len(tape) + 20)
    assert r["overflow"] == 1 and (r["coff"] == 16).all() and (r["clen"] == len(tape)).all()
    # the failed reservation clamps the bump pointer to the capacity: it never falls below a range another wave was granted
    # (a give-back by subtraction could, between two failures with a success in between)
    assert r["head"] == 16 + len(tape) + 20 and r["head"] >= r["head0"]


@pytest.mark.parametrize("kernel", ["fh_tiles_v32", "fh_tiles_v64"])
def test_both_slot_lists_in_one_launch(kernel):
    """flags bit 4 (a pre-pass level's few hundred parents: one launch instead of two): the waves walk the `big` list, then the
    other; each slot comes out as when it is the only one of its launch."""
    off = U.offsets()
    sh, tape, ik = shape_of(2)
    n = len(tape)
    boxes = [children((0.1, -0.2, 0.3), 0.5), children((-0.3, 0.25, 0.0), 0.45)]
    alone = [run_tiles(kernel, tape, b, ik, sh.slot_count(), sh.choice_count()) for b in boxes]
    mem = E.Memory()
    arena = np.zeros(ARENA_OPS, np.uint64)
    arena[16:16 + n] = tape
    a_arena = mem.map(arena, "arena")
    st = U.Blob(off["sizeof_state"])
    slots = []
    for b in boxes:
        slot = U.Blob(off["sizeof_slot"])
        slot.u32(0, 16); slot.u32(4, n); slot.u32(8, sh.slot_count() | (sh.choice_count() << 16)); slot.u32(12, 2)
        slot.u64(16, (1 << 64) - 1)
        for k in range(6):
            slot.arr(40 + 256 * k, np.asarray(b[k], F32))
        slots.append(slot)
    a0, a1 = mem.map(slots[0].b, "slot0"), mem.map(slots[1].b, "slot1")
    head0 = 16 + n + 16
    st.u64(off["arena"], a_arena); st.u32(off["arena_cap"], ARENA_OPS - 64); st.u32(off["arena_head"], head0)
    st.u64(off["slots"], a0); st.u64(off["slots"] + 8, a1)
    st.u32(off["slot_cap"], 1); st.u32(off["slot_cap"] + 4, 1)
    level = 2
    for big in (0, 1):
        st.u32(off["n_slots"] + 4 * (big * 8 + level), 1)
    for s in range(16):
        st.u32(off["P.in_kind"] + 4 * s, ik[s] if s < len(ik) else 3)
    a_st = mem.map(st.b, "state")
    mr, mc = LIMITS[kernel]
    ka = np.zeros(10, U32)
    ka[0], ka[1] = a_st & 0xFFFFFFFF, a_st >> 32
    ka[2:10] = [level, 1, mr, mc, 1, 16, 0, 0]
    E.launch(U.program(), mem, kernel, ka.tobytes(), 1, lds_bytes=16, n_vgpr=N_VGPR[kernel])
    for sl, ref in zip(slots, alone):
        g = lambda k: sl.b[40 + k * 256: 40 + (k + 1) * 256]
        assert (g(9).view(U32) == ref["res"][0].view(U32)).all() and (g(10).view(U32) == ref["res"][1].view(U32)).all()
        assert (g(12).view(U32) == ref["clen"]).all() and (g(13).view(U32) == ref["crc"]).all()


def test_v64_exports_choices_of_parents_that_carry_links():
    """flags bit 1 (level 1 behind the linked prune, prune2.hip): a parent whose tape has this frame's stamp in front - frame_stamp << 32
    | len | choices << 16 at off - len - choices - 1 - gets its choice words written to chw[list] ([slot][word][lane], 16 words per
    slot in list 0, flags[31:16] in list 1) and its ambiguous, decided children marked (c_len = ~0, c_off = end of their arena slot);
    a parent without the stamp (or with another frame's) is pruned in the kernel as ever.  Both lists in one launch."""
    off = U.offsets()
    sh, tape, ik = shape_of(2)
    n, regs, nch = len(tape), sh.slot_count(), sh.choice_count()
    boxes = [children((0.1, -0.2, 0.3), 0.5), children((-0.3, 0.25, 0.0), 0.45), children((0.2, 0.1, -0.1), 0.4)]
    alone = [run_tiles("fh_tiles_v64", tape, b, ik, regs, nch) for b in boxes]
    STAMP, STRIDE1 = 77, 40
    mem = E.Memory()
    arena = np.zeros(ARENA_OPS, np.uint64)
    offs = [2048, 4096, 8192]                       # three copies of the tape: stamped, not stamped, stamped by another frame
    for o_ in offs:
        arena[o_:o_ + n] = tape
    arena[offs[0] - n - nch - 1] = (STAMP << 32) | n | (nch << 16)
    arena[offs[2] - n - nch - 1] = ((STAMP - 1) << 32) | n | (nch << 16)
    a_arena = mem.map(arena, "arena")
    st = U.Blob(off["sizeof_state"])
    lists = [U.Blob(off["sizeof_slot"] * 2), U.Blob(off["sizeof_slot"] * 2)]           # list 1 (big): slots A (stamped), C; list 0: slot B... and A again
    plan = {(1, 0): (0, 0), (1, 1): (2, 2), (0, 0): (1, 1), (0, 1): (0, 2)}             # (list, index) -> (tape copy, box)
    for (lst, idx), (tcopy, box) in plan.items():
        b0 = off["sizeof_slot"] * idx
        sl = lists[lst]
        sl.u32(b0 + 0, offs[tcopy]); sl.u32(b0 + 4, n); sl.u32(b0 + 8, regs | (nch << 16)); sl.u32(b0 + 12, 2)
        sl.u64(b0 + 16, (1 << 64) - 1)
        for k in range(6):
            sl.arr(b0 + 40 + 256 * k, np.asarray(boxes[box][k], F32))
    a_l0, a_l1 = mem.map(lists[0].b, "slots0"), mem.map(lists[1].b, "slots1")
    chw = [np.full(2 * 16 * 64, 0xDEADBEEF, U32), np.full(2 * STRIDE1 * 64, 0xDEADBEEF, U32)]
    a_c0, a_c1 = mem.map(chw[0], "chw0"), mem.map(chw[1], "chw1")
    head0 = 16384
    st.u64(off["arena"], a_arena); st.u32(off["arena_cap"], ARENA_OPS - 64); st.u32(off["arena_head"], head0)
    st.u64(off["slots"], a_l0); st.u64(off["slots"] + 8, a_l1)
    st.u64(off["chw"], a_c0); st.u64(off["chw"] + 8, a_c1)
    st.u32(off["frame_stamp"], STAMP)
    st.u32(off["slot_cap"], 2); st.u32(off["slot_cap"] + 4, 2)
    level = 1
    for big in (0, 1):
        st.u32(off["n_slots"] + 4 * (big * 8 + level), 2)
    for s in range(16):
        st.u32(off["P.in_kind"] + 4 * s, ik[s] if s < len(ik) else 3)
    a_st = mem.map(st.b, "state")
    ka = np.zeros(10, U32)
    ka[0], ka[1] = a_st & 0xFFFFFFFF, a_st >> 32
    ka[2:10] = [level, 1, 64, 512, 1, 16 | 2 | (STRIDE1 << 16), 0, 0]
    E.launch(U.program(), mem, "fh_tiles_v64", ka.tobytes(), 1, lds_bytes=16, n_vgpr=N_VGPR["fh_tiles_v64"])
    head = int(st.get_u32(off["arena_head"])[0])
    for (lst, idx), (tcopy, box) in plan.items():
        b0 = off["sizeof_slot"] * idx
        g = lambda k: lists[lst].b[b0 + 40 + k * 256: b0 + 40 + (k + 1) * 256]
        ref = alone[box]
        assert (g(9).view(U32) == ref["res"][0].view(U32)).all() and (g(10).view(U32) == ref["res"][1].view(U32)).all()
        pruned = ref["clen"] != n if True else None          # (children the kernel pruned when it ran the slot alone)
        coff, clen, crc = g(11).view(U32), g(12).view(U32), g(13).view(U32)
        if tcopy != 0:
            # no stamp of this frame: pruned here, as alone (the tapes land elsewhere in the arena: compare contents)
            assert (clen == ref["clen"]).all() and (crc == ref["crc"]).all()
            for lane in np.nonzero(pruned)[0]:
                assert (arena[coff[lane]:coff[lane] + clen[lane]] == ref["arena"][ref["coff"][lane]:ref["coff"][lane] + ref["clen"][lane]]).all()
            continue
        assert pruned.any()
        assert (clen[pruned] == 0xFFFFFFFF).all() and (clen[~pruned] == n).all() and (coff[~pruned] == offs[0]).all()
        assert (crc == (regs | (nch << 16))).all()
        ends = np.sort(coff[pruned])
        assert ((ends[1:] - ends[:-1]) == n).all() and ends[0] - n >= head0 and ends[-1] <= head, "one arena slot of the parent's length per marked child"
        xyz = boxes[box]
        inputs = {s: (xyz[2 * k], xyz[2 * k + 1]) for s, k in enumerate(ik) if k < 3}
        _, _, ch, _ = U.ref_interval(tape, inputs, 64)
        stride = STRIDE1 if lst == 1 else 16
        words = chw[lst][idx * stride * 64:(idx + 1) * stride * 64].reshape(stride, 64)
        for q in range(nch):
            got = (words[q >> 4] >> U32((q & 15) * 2)) & U32(3)
            assert (got[pruned] == ch[q][pruned]).all(), f"choice {q}"
        assert (words[(nch + 15) // 16:] == 0xDEADBEEF).all()


def trans_shape(which):
    """shapes with the transcendental opcodes, every interval rule of types/interval.rs:136-302 exercised over the boxes below"""
    import fidget_amd as F
    c = F.Context()
    x, y, z = c.x(), c.y(), c.z()
    if which == 0:      # gyroid-like: sin / cos of scaled axes (quadrant tables, the >= TAU and >= PI cases at large scales)
        n = c.sub(c.add(c.add(c.mul(c.sin(c.mul(x, 7.0)), c.cos(c.mul(y, 5.0))), c.mul(c.sin(c.mul(y, 2.5)), c.cos(c.mul(z, 11.0)))),
                        c.min(c.sin(c.mul(z, 0.7)), c.cos(x))), 0.2)
    elif which == 1:    # exp / ln / atan, a union with a sphere
        r2 = c.add(c.add(c.square(x), c.square(y)), c.square(z))
        n = c.min(c.sub(c.exp(c.neg(r2)), 0.5), c.add(c.ln(c.add(r2, 0.05)), c.atan(c.mul(x, 4.0))))
    elif which == 2:    # tan / asin / acos: domains left on purpose in part of the boxes
        n = c.max(c.sub(c.tan(c.mul(x, 1.3)), c.asin(c.mul(y, 1.4))), c.sub(c.acos(c.mul(z, 1.2)), 1.0))
    elif which == 3:    # atan2 in its three operand forms (every sign case of y and x over the boxes, the whole circle where y has 0 and x < 0)
        n = c.min(c.sub(c.atan2(y, x), c.mul(z, 2.0)), c.add(c.atan2(c.add(z, 0.1), 0.3), c.atan2(-0.2, c.sub(x, y))))
    elif which == 4:    # modulo: a point divisor (reg % imm: the same-floor rule), an interval divisor with and without 0, imm % reg
        n = c.max(c.sub(c.modulo(c.mul(x, 3.0), 0.7), 0.3), c.min(c.modulo(y, c.add(z, 1.5)), c.modulo(2.5, c.add(c.square(x), 0.25))))
    elif which == 5:    # rand: the point where the operand is ONE bit pattern (floor of a narrow interval), [0, 1] elsewhere
        fx = c.floor(c.mul(x, 0.4))
        n = c.min(c.sub(c.add(c.rand(fx), c.mul(c.rand(y), z)), 0.6), c.sub(c.rand(c.floor(c.add(c.mul(z, 3.0), 0.5))), 0.5))
    else:               # mix in its three operand forms: points where both operands are one bit pattern, NaN in the other lanes
        fx, fy, fz = c.floor(c.mul(x, 0.4)), c.floor(c.mul(y, 0.3)), c.floor(c.add(c.mul(z, 0.2), 0.5))
        n = c.sub(c.add(c.mul(c.mix(fy, 3.0), 1.0e-9), c.mul(c.mix(fz, fx), 1.0e-9)), c.mul(c.mix(2.0, fy), 1.0e-9))
    sh = F.Shape(c, n)
    ik = [3] * 16
    for a in range(3):
        s_ = sh.axis_index(a)
        if s_ >= 0:
            ik[s_] = a
    return sh, U.shape_tape(sh), ik


@pytest.mark.parametrize("kernel", ["fh_tiles_t", "fh_tiles_v32_t", "fh_tiles_v64_t"])
@pytest.mark.parametrize("which", [0, 1, 2, 3, 4, 5, 6])
def test_transcendental_interval_handlers(kernel, which):
    """the *_t tile kernels: interval sin cos tan asin acos atan exp ln and (round 5) atan2 / modulo / rand / mix (dev_ops.hpp iv_sincos ..
    iv_ln, iv_atan2, iv_rem_euclid, iv_rand, iv_mix around the compiled f32 routines) - results, choices and pruned child tapes against the numpy restatement, small and large boxes (a box many periods wide
    is [-1, 1]; one inside a quadrant is monotonic; domains of asin / acos / ln / tan left in some children)"""
    sh, tape, ik = trans_shape(which)
    if sh.slot_count() > LIMITS[kernel][0] or sh.choice_count() > LIMITS[kernel][1]:
        pytest.skip("tape outside this kernel's register file")
    for center, half in (((0.1, -0.2, 0.3), 0.5), ((-0.3, 0.25, 0.0), 0.06), ((0.4, 0.4, -0.4), 2.5), ((0.7, -0.6, 0.2), 0.2)):
        check_slot(kernel, tape, children(center, half), ik, sh.slot_count(), sh.choice_count())
