"""fh_columns (the assembly leaf interpreter: f32 point evaluation of 8x8x8 voxel leaves, z-buffer by 64-bit atomic max) on
the CPU emulator against a numpy evaluation of the same tape at the same voxels (tests/emu_util.py ref_f32, which follows
fidget-core/src/vm/mod.rs:794-1086 via dev_ops.hpp; the voxel -> model transform is dev_ops.hpp xf_point, i.e. nalgebra's
transform_point as shape/mod.rs:906-916 calls it): depth words bit for bit, affine AND projective matrices."""
import numpy as np
import pytest

import emu_util as U
from emu_util import E, F32, U32
from test_emu_tiles import shape_of


def xf_point(m, x, y, z):
    m = m.astype(F32)
    x, y, z = (np.asarray(v, F32) for v in (x, y, z))
    rows = [((m[4 * r] * x + m[4 * r + 1] * y) + m[4 * r + 2] * z) + m[4 * r + 3] for r in range(4)]
    with np.errstate(all="ignore"):
        return [np.where(rows[3] != 0, rows[r] / rows[3], rows[r]).astype(F32) for r in range(3)]


def col_kernarg(a_st, in_kind, mat, size=16, column_mode=False, group_log2=6, a_tab=0, layers=None):
    """the leaf kernel's kernarg as capi_render.hpp builds it: state, n_waves = 0, axis slots, inputs varying along a column, flags, and
    floor(2^32 / blocks of four footprints per layer) for the kernel's block rotation (0 when there is one block: the subtraction loop)"""
    u = np.asarray(mat, F32).view(U32)
    proj = bool(((int(u[12]) | int(u[13]) | int(u[14])) & 0x7FFFFFFF) | (int(u[15]) ^ 0x3F800000))
    slot = [-1, -1, -1]
    for s_, k in enumerate(list(in_kind) + [3] * (16 - len(in_kind))):
        if k < 3:
            slot[k] = s_
    slots = sum((0xFF if slot[ax] < 0 else slot[ax]) << (8 * ax) for ax in range(3))
    dep, flags = 0, 0x10000 if proj else 0
    for ax in range(3):
        varies = proj or (int(u[4 * ax + 2]) & 0x7FFFFFFF)
        if slot[ax] >= 0 and varies:
            dep |= 1 << slot[ax]
        if varies:
            flags |= 0x20000 << ax          # (bits 17 .. 19: this axis of the model changes along a pixel column)
    n_blocks = (((size + 7) // 8) ** 2 + 3) // 4
    if column_mode:
        flags |= (1 << 20) | (group_log2 << 24)   # (one footprint column per wave, lane = layer; 2^g layers of it per wave, grid y = the group)
    # (round 6: the slab's leaf table, its footprints per layer and its layers come with the kernarg - a wave looks at its entries before the state's words arrive)
    return np.array([a_st & 0xFFFFFFFF, a_st >> 32, 0, slots, dep, flags, (1 << 32) // n_blocks if n_blocks > 1 else 0, 0,
                     a_tab & 0xFFFFFFFF, a_tab >> 32, ((size + 7) // 8) ** 2, size // 8 if layers is None else layers], U32)


def run_columns(tape, n_regs, in_kind, mat, leaf_xyz, size=16, zbuf_init=None, n_choices=0, kernel="fh_columns", column_mode=False, more_leaves=(), group_log2=6):
    off = U.offsets()
    mem = E.Memory()
    arena = np.zeros(4096, np.uint64)
    arena[16:16 + len(tape)] = tape
    a_arena = mem.map(arena)
    nfp = (size // 8) ** 2
    layers = size // 8
    table = np.zeros((layers * nfp, 4), U32)       # FhLeafRef: id + 1, tape offset, length | regs << 24, x | y << 16
    leaves = np.zeros((4, 6), U32)
    lx, ly, lz = leaf_xyz
    leaves[0] = [16, len(tape), n_regs | (n_choices << 16), lx, ly, lz]
    table[(lz % size) // 8 * nfp + (ly // 8) * (size // 8) + lx // 8] = [1, 16, len(tape) | (n_regs << 24), lx | (ly << 16)]
    for k, (mx, my, mz) in enumerate(more_leaves):      # (the same tape at other places of the same slab: leaf ids 2, 3, ..)
        assert mz - mz % size == lz - lz % size
        table[(mz % size) // 8 * nfp + (my // 8) * (size // 8) + mx // 8] = [2 + k, 16, len(tape) | (n_regs << 24), mx | (my << 16)]
    zbuf = np.zeros(size * size, np.uint64) if zbuf_init is None else zbuf_init.copy()
    a_tab, a_leaves, a_z = mem.map(table), mem.map(leaves), mem.map(zbuf)
    st = U.Blob(off["sizeof_state"])
    st.arr(off["P.mat"], np.asarray(mat, F32))
    st.u32(off["P.width"], size); st.u32(off["P.height"], size); st.u32(off["P.tiles"], size); st.u32(off["P.slab"], size)
    for s in range(16):
        st.u32(off["P.in_kind"] + 4 * s, in_kind[s] if s < len(in_kind) else 3)
    st.u64(off["arena"], a_arena); st.u64(off["leaves"], a_leaves); st.u64(off["leaf_table"], a_tab); st.u64(off["zbuf"], a_z)
    st.u32(off["slab_z"], lz - lz % size)
    a_st = mem.map(st.b)
    ka = col_kernarg(a_st, in_kind, mat, size, column_mode, group_log2, a_tab=a_tab, layers=layers)
    trans = kernel == "fh_columns_t"
    gx, gy = ((nfp + 63) // 64 * 64, (layers + (1 << group_log2) - 1) >> group_log2) if column_mode else ((nfp + 3) // 4, layers)
    waves = E.launch(U.program(), mem, kernel, ka.tobytes(), gx, grid_y=gy, lds_bytes=16, n_vgpr=256 if trans else 128,
                     hooks=U.trans_hooks(U.program(), v_base=224, window=22) if trans else None)     # (window registers 22, 23: the hand-written expf's table)
    return zbuf, waves


def expect(tape, in_kind, mat, leaf_xyz, size, zbuf_init=None):
    lx, ly, lz = leaf_xyz
    z0 = np.zeros(size * size, np.uint64) if zbuf_init is None else zbuf_init.copy()
    for py in range(ly, min(ly + 8, size)):
        for px in range(lx, min(lx + 8, size)):
            i = py * size + px
            if (int(z0[i]) >> 32) >= lz + 8:
                continue
            zs = np.arange(lz + 7, lz - 1, -1)
            X, Y, Z = xf_point(np.asarray(mat, F32), np.full(8, px), np.full(8, py), zs)
            inputs = {s: (X, Y, Z)[k] for s, k in enumerate(in_kind) if k < 3}
            v = U.ref_f32(tape, inputs, 8)[0]
            hit = np.nonzero(v < 0)[0]
            if len(hit):
                z0[i] = max(int(z0[i]), ((int(zs[hit[0]]) + 1) << 32) | 1)
    return z0


AFFINE = [0.125, 0, 0, -1, 0, -0.125, 0, 0.875, 0, 0, 0.125, -1, 0, 0, 0, 1]                 # screen_to_world of a 16^3 volume
AFFINE32 = [0.0625, 0, 0, -1, 0, -0.0625, 0, 0.9375, 0, 0, 0.0625, -1, 0, 0, 0, 1]            # ... of a 32^3 volume
ROTATED = [0.1, 0.05, 0.02, -1.1, -0.04, -0.11, 0.03, 0.9, 0.01, 0.02, 0.12, -0.95, 0, 0, 0, 1]
PERSPECTIVE = [0.125, 0, 0, -1, 0, -0.125, 0, 0.875, 0, 0, 0.125, -1, 0.01, -0.005, 0.0375, 0.7]   # w = 0.7 + ... (never 0 here)
W_ZERO = [0.125, 0, 0, -1, 0, -0.125, 0, 0.875, 0, 0, 0.125, -1, 0, 0, 0.125, -1.0]              # w = z/8 - 1: exactly 0 at z = 8


@pytest.mark.parametrize("mat", [AFFINE, ROTATED, PERSPECTIVE, W_ZERO], ids=["affine", "rotated", "perspective", "w_zero"])
@pytest.mark.parametrize("seed", [0, 2, 3, 5])
def test_leaf_against_numpy(seed, mat):
    sh, tape, ik = shape_of(seed)
    if sh.slot_count() > 32:
        pytest.skip("needs the LDS leaf kernel")
    for leaf in ((0, 8, 0), (8, 0, 8)):
        got, waves = run_columns(tape, sh.slot_count(), ik, mat, leaf)
        want = expect(tape, ik, mat, leaf, 16)
        assert (got == want).all(), f"{(got != want).sum()} z-buffer words differ"


def test_occluded_pixels_are_left_alone():
    sh, tape, ik = shape_of(0)
    z = np.zeros(256, np.uint64)
    z[0:128] = np.uint64((16 << 32) | 7)     # front rows already hit by a nearer leaf
    got, _ = run_columns(tape, sh.slot_count(), ik, AFFINE, (0, 0, 0), zbuf_init=z)
    assert (got == expect(tape, ik, AFFINE, (0, 0, 0), 16, z)).all()


def trans_shape(kind):
    import fidget_amd as F
    c = F.Context()
    x, y, z = c.x(), c.y(), c.z()
    if kind == 0:      # unary transcendentals, every one of them
        n = c.add(c.mul(c.sin(c.mul(x, 3.0)), c.cos(c.mul(y, 2.0))), c.sub(c.exp(c.mul(z, 0.5)), 1.2))
        n = c.add(n, c.mul(c.tan(c.mul(x, 0.4)), 0.1))
        n = c.add(n, c.mul(c.asin(c.mul(y, 0.5)), c.acos(c.mul(z, 0.5))))
        n = c.sub(n, c.mul(c.atan(x), c.ln(c.add(c.abs(y), 0.3))))
    elif kind == 1:    # atan2 / mod in the three operand forms
        n = c.sub(c.atan2(y, x), c.mul(z, 2.0))
        n = c.add(n, c.sub(c.modulo(c.mul(x, 3.0), 0.7), 0.3))
        n = c.add(n, c.mul(c.modulo(2.5, c.add(c.abs(y), 0.4)), 0.2))
        n = c.add(n, c.mul(c.atan2(0.3, z), c.atan2(x, -0.6)))
        n = c.add(n, c.mul(c.modulo(c.add(x, 2.0), c.add(c.square(z), 0.5)), 0.1))
    else:              # rng
        n = c.sub(c.add(c.rand(c.mul(x, 7.0)), c.mix(y, z)), c.add(c.mix(x, 0.25), c.mix(1.5, z)))
        n = c.sub(n, 0.2)
    sh = F.Shape(c, n)
    ik = [3] * 16
    for a in range(3):
        s = sh.axis_index(a)
        if s >= 0:
            ik[s] = a
    return sh, U.shape_tape(sh), ik


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_transcendental_leaf_kernel(kind):
    """fh_columns_t: the opcodes that call the compiled routines (stood in for natively in the emulator: what is tested is the
    call sequence, the register window and the operand forms) and the rng opcodes (integer code, emulated exactly)"""
    sh, tape, ik = trans_shape(kind)
    assert sh.slot_count() <= 32
    for mat in (AFFINE, PERSPECTIVE):
        for leaf in ((0, 8, 0), (8, 0, 8)):
            got, _ = run_columns(tape, sh.slot_count(), ik, mat, leaf, kernel="fh_columns_t")
            want = expect(tape, ik, mat, leaf, 16)
            assert (got == want).all(), f"{(got != want).sum()} z-buffer words differ"
    # and a tape without such opcodes gives the same words through either kernel
    sh2, tape2, ik2 = shape_of(2)
    a, _ = run_columns(tape2, sh2.slot_count(), ik2, ROTATED, (0, 8, 0))
    b, _ = run_columns(tape2, sh2.slot_count(), ik2, ROTATED, (0, 8, 0), kernel="fh_columns_t")
    assert (a == b).all()


def column_shape(kind):
    """kind 0: a function of x and y only (an extrusion); 1: the same plus a z term"""
    import fidget_amd as F
    c = F.Context()
    x, y, z = c.x(), c.y(), c.z()
    n = c.sub(c.sqrt(c.add(c.square(c.sub(x, 0.2)), c.square(c.mul(y, 1.3)))), 0.55)
    n = c.max(n, c.sub(0.25, c.abs(c.add(x, c.mul(y, 0.5)))))
    n = c.min(n, c.sub(c.abs(c.sub(y, 0.4)), 0.1))
    if kind == 1:
        n = c.max(n, c.sub(c.abs(z), 0.6))
    sh = F.Shape(c, n)
    ik = [3] * 16
    for a in range(3):
        s = sh.axis_index(a)
        if s >= 0:
            ik[s] = a
    return sh, U.shape_tape(sh), ik


@pytest.mark.parametrize("mat", [AFFINE, ROTATED, PERSPECTIVE], ids=["affine", "rotated", "perspective"])
@pytest.mark.parametrize("kind", [0, 1])
def test_column_invariant_leaves(kind, mat):
    """A leaf tape that reads no input varying along the pixel column (no z under an axis-aligned camera) is evaluated once
    per pixel instead of once per voxel; with a rotated or projective camera x and y vary with z and it is not.  Same words
    either way, and the short cut really is taken (fewer instructions) exactly when it applies."""
    sh, tape, ik = column_shape(kind)
    counts = {}
    for leaf in ((0, 8, 0), (8, 0, 8), (8, 8, 0)):
        z = np.zeros(256, np.uint64)
        z[5::7] = np.uint64((3 << 32) | 9)        # some pixels already hit further back: still pending
        got, waves = run_columns(tape, sh.slot_count(), ik, mat, leaf, zbuf_init=z)
        want = expect(tape, ik, mat, leaf, 16, z)
        assert (got == want).all(), f"{(got != want).sum()} z-buffer words differ"
        counts[leaf] = sum(w.counts.get("valu", 0) for w in waves)
    if kind == 0:       # vector instructions of the same leaf under the axis-aligned camera (short cut) and the rotated one (none)
        _, wa = run_columns(tape, sh.slot_count(), ik, AFFINE, (0, 8, 0))
        _, wr = run_columns(tape, sh.slot_count(), ik, ROTATED, (0, 8, 0))
        va, vr = (sum(w.counts.get("valu", 0) for w in ws) for ws in (wa, wr))
        assert va < 0.6 * vr, (va, vr)


def run_block(leaves_spec, in_kind, mat, size=16, zbuf_init=None, kernel="fh_columns"):
    """several leaves in one block of footprints: leaves_spec = [(tape, n_regs, (x, y, z))], tapes laid out one after the other"""
    off = U.offsets()
    mem = E.Memory()
    arena = np.zeros(8192, np.uint64)
    nfp = (size // 8) ** 2
    layers = size // 8
    table = np.zeros((layers * nfp, 4), U32)
    leaves = np.zeros((len(leaves_spec) + 1, 6), U32)
    pos = 16
    slab_z = None
    for i, (tape, n_regs, (lx, ly, lz)) in enumerate(leaves_spec):
        arena[pos:pos + len(tape)] = tape
        leaves[i] = [pos, len(tape), n_regs, lx, ly, lz]
        table[(lz % size) // 8 * nfp + (ly // 8) * (size // 8) + lx // 8] = [i + 1, pos, len(tape) | (n_regs << 24), lx | (ly << 16)]
        pos += len(tape) + 24
        slab_z = lz - lz % size
    zbuf = np.zeros(size * size, np.uint64) if zbuf_init is None else zbuf_init.copy()
    a_arena, a_tab, a_leaves, a_z = mem.map(arena), mem.map(table), mem.map(leaves), mem.map(zbuf)
    st = U.Blob(off["sizeof_state"])
    st.arr(off["P.mat"], np.asarray(mat, F32))
    st.u32(off["P.width"], size); st.u32(off["P.height"], size); st.u32(off["P.tiles"], size); st.u32(off["P.slab"], size)
    for s in range(16):
        st.u32(off["P.in_kind"] + 4 * s, in_kind[s] if s < len(in_kind) else 3)
    st.u64(off["arena"], a_arena); st.u64(off["leaves"], a_leaves); st.u64(off["leaf_table"], a_tab); st.u64(off["zbuf"], a_z)
    st.u32(off["slab_z"], slab_z)
    a_st = mem.map(st.b)
    ka = col_kernarg(a_st, in_kind, mat, size, a_tab=a_tab, layers=layers)
    E.launch(U.program(), mem, kernel, ka.tobytes(), (nfp + 3) // 4, grid_y=layers, lds_bytes=16, n_vgpr=128)
    return zbuf


def test_block_of_leaves_with_tapes_requested_ahead():
    """All four footprints of a layer hold a leaf (one block for one wave): short tapes are requested one leaf ahead, a tape of
    more than 64 ops and a leaf of the LDS class (left to another kernel) sit in between; every pixel as if each leaf were alone."""
    sh0, t0, ik = column_shape(1)
    sh1, t1, _ = column_shape(0)
    import fidget_amd as F
    c = F.Context()
    x, y, z = c.x(), c.y(), c.z()
    n = c.sub(c.add(c.square(x), c.add(c.square(y), c.square(z))), 0.5)
    for k in range(30):                     # a long chain: more than 64 ops, few registers
        n = c.min(c.add(n, 0.01), c.sub(c.add(c.mul(x, 0.1 * (k + 1)), c.mul(y, 0.03 * k)), c.mul(z, 0.2)))
    shl = F.Shape(c, n)
    tl = U.shape_tape(shl)
    assert len(tl) > 64 and shl.slot_count() <= 32
    assert all(shl.axis_index(a) == sh0.axis_index(a) for a in range(3))      # same input slots: one in_kind for the block
    for mat in (AFFINE, ROTATED):
        for order in ([t0, tl, t1, t0], [tl, t1, t0, tl], [t1, t0, t0, t1]):
            regs = {id(t0): sh0.slot_count(), id(t1): sh1.slot_count(), id(tl): shl.slot_count()}
            spec = [(t, regs[id(t)], (8 * (i % 2), 8 * (i // 2), 8)) for i, t in enumerate(order)]
            got = run_block(spec, ik, mat)
            want = np.zeros(256, np.uint64)
            for i, (t, r, xyz) in enumerate(spec):
                e = expect(t, ik, mat, xyz, 16)
                hit = e != 0
                want[hit] = (e[hit] & ~np.uint64(0xFFFFFFFF)) | np.uint64(i + 1)
            assert (got == want).all(), f"{(got != want).sum()} z-buffer words differ"
        # an LDS-class leaf (more than 40 registers: beyond the kernel's largest shape, 40 x 2) in the block is skipped, its neighbours are not
        # disturbed; a leaf that says it needs 40 is taken (the 40 x 2 shape: four passes of two voxels)
        spec = [(t0, sh0.slot_count(), (0, 0, 0)), (t1, 41, (8, 0, 0)), (t1, 40, (0, 8, 0)), (t0, sh0.slot_count(), (8, 8, 0))]
        got = run_block(spec, ik, mat)
        want = np.zeros(256, np.uint64)
        for i, (t, r, xyz) in enumerate(spec):
            if r > 40:
                continue
            e = expect(t, ik, mat, xyz, 16)
            hit = e != 0
            want[hit] = (e[hit] & ~np.uint64(0xFFFFFFFF)) | np.uint64(i + 1)
        assert (got == want).all()


def wide_trans_shape(nt=18, trans=True):
    """a smooth blend in the manner of bear.vm - 18 exp(-k r_i^2) terms, each used twice (a sum of them over a sum of pairwise
    products), so that all are alive at once: 22 registers, fh_columns_t's 32 x 4 class (two passes of four voxels per lane)"""
    import fidget_amd as F
    c = F.Context()
    x, y, z = c.x(), c.y(), c.z()
    t = []
    for i in range(nt):
        cx, cy, cz = 0.3 * np.cos(i), 0.3 * np.sin(2 * i), 0.2 * np.cos(3 * i + 1)
        r2 = c.add(c.add(c.square(c.sub(x, float(cx))), c.square(c.sub(y, float(cy)))), c.square(c.sub(z, float(cz))))
        t.append(c.exp(c.mul(r2, -3.0 - 0.5 * i)) if trans else c.div(1.0, c.add(c.mul(r2, 3.0 + 0.5 * i), 0.2)))
    A = t[0]
    for k in range(1, nt):
        A = c.add(A, t[k])
    B = c.mul(t[0], t[nt - 1])
    for k in range(1, nt - 1):
        B = c.add(B, c.mul(t[k], t[nt - 1 - k]))
    n = c.sub(c.div(A, c.add(B, 0.1)), c.sqrt(c.add(c.ln(c.add(c.square(x), 1.5)), c.sin(y))) if trans else c.mul(c.sqrt(c.add(c.square(x), 0.3)), 9.0))
    sh = F.Shape(c, n)
    ik = [3] * 16
    for a in range(3):
        s = sh.axis_index(a)
        if s >= 0:
            ik[s] = a
    return sh, U.shape_tape(sh), ik


def test_transcendental_leaf_kernel_32x4_class():
    sh, tape, ik = wide_trans_shape()
    assert 16 < sh.slot_count() <= 32, sh.slot_count()
    hits = misses = 0
    for mat in (AFFINE, ROTATED):
        for leaf in ((0, 8, 0), (8, 0, 8), (8, 8, 0)):
            got, _ = run_columns(tape, sh.slot_count(), ik, mat, leaf, kernel="fh_columns_t")
            want = expect(tape, ik, mat, leaf, 16)
            assert (got == want).all(), f"{(got != want).sum()} z-buffer words differ"
            hits += int((want != 0).sum()); misses += int((want == 0).sum())
    print("pixels hit", hits, "not hit", misses)
    assert hits > 0


@pytest.mark.parametrize("kernel,nt,lo,hi", [("fh_columns_t", 34, 32, 44), ("fh_columns", 34, 32, 40), ("fh_columns", 22, 20, 32)])
def test_leaves_of_more_than_32_registers_in_the_largest_shapes(kernel, nt, lo, hi):
    """the leaf kernels' largest register-file shapes (fh_columns 40 x 2: four passes of two voxels; fh_columns_t 44 x 4) take leaves the
    C++ kernel with its LDS file had until round 6 (a 128^3 frame of prospero.vm: ten such leaves, 0.9 ms)"""
    sh, tape, ik = wide_trans_shape(nt, trans=kernel.endswith("_t"))
    assert lo < sh.slot_count() <= hi, sh.slot_count()
    hits = 0
    for mat in (AFFINE, ROTATED):
        for leaf in ((0, 8, 0), (8, 0, 8)):
            got, _ = run_columns(tape, sh.slot_count(), ik, mat, leaf, kernel=kernel)
            want = expect(tape, ik, mat, leaf, 16)
            assert (got == want).all(), f"{(got != want).sum()} z-buffer words differ"
            hits += int((want != 0).sum())
    assert hits > 0


ROTATED32 = [0.05, 0.025, 0.01, -1.1, -0.02, -0.055, 0.015, 0.9, 0.005, 0.01, 0.06, -0.95, 0, 0, 0, 1]


@pytest.mark.parametrize("mat", [AFFINE32, ROTATED32], ids=["affine", "rotated"])
@pytest.mark.parametrize("kind", [0, 1])
def test_column_mode_gives_the_layer_modes_words(kind, mat):
    """kernarg flags bit 20: one footprint column per wave (lane = layer, front layer first) instead of one block of four footprints
    of one layer - the arrangement for frames whose leaf table is nearly empty.  Same z-buffer words, leaf ids included, with several
    leaves in a column and leaves in other columns; and an empty column costs a wave one load and nothing else."""
    sh, tape, ik = column_shape(kind)
    size = 32
    leaves = [(8, 16, 8), (8, 16, 24), (8, 16, 0), (24, 0, 16), (0, 24, 24)]
    z = np.zeros(size * size, np.uint64)
    z[3::11] = np.uint64((5 << 32) | 9)
    a, wa = run_columns(tape, sh.slot_count(), ik, mat, leaves[0], size=size, zbuf_init=z, more_leaves=leaves[1:])
    b, wb = run_columns(tape, sh.slot_count(), ik, mat, leaves[0], size=size, zbuf_init=z, more_leaves=leaves[1:], column_mode=True)
    assert (a != z).any()
    assert (a == b).all(), f"{(a != b).sum()} z-buffer words differ"
    # the words themselves: leaf by leaf against numpy (ids 1 .. 5; the maximum per pixel, as the atomic takes it)
    want = z.copy()
    for k, leaf in enumerate(leaves):
        w = expect(tape, ik, mat, leaf, size, z)
        hit = w != z
        w = np.where(hit, (w & ~np.uint64(0xFFFFFFFF)) | np.uint64(k + 1), w)
        want = np.maximum(want, w)
    # (a pixel that a nearer leaf of its column hit is not evaluated again by the leaves behind it: the words are the same either way)
    assert (b == want).all(), f"{(b != want).sum()} z-buffer words differ from numpy's"
    busy = [w for w in wb if w.counts.get("vmem", 0) > 1]
    assert len(wb) == 64 and len(busy) == 3, (len(wb), len(busy))        # 16 footprints in a grid of 64, three columns with leaves


@pytest.mark.parametrize("g", [0, 1])
@pytest.mark.parametrize("kind", [0, 1])
def test_column_mode_by_groups_of_layers(kind, g):
    """column mode with 2^g layers of a footprint's column per wave (grid y = the group, front group first): what frames with a leaf
    in most layers take - the pixel's set-up, its z-buffer word and its hits stay in the wave from leaf to leaf, one atomic per wave.
    Same words as the layer walk and as numpy; the pixel hit by a wave's nearer leaf is not evaluated by the one behind it."""
    sh, tape, ik = column_shape(kind)
    size = 32
    leaves = [(8, 16, 8), (8, 16, 24), (8, 16, 0), (8, 16, 16), (24, 0, 16), (0, 24, 24), (0, 24, 16)]
    z = np.zeros(size * size, np.uint64)
    z[3::11] = np.uint64((5 << 32) | 9)
    for mat in (AFFINE32, ROTATED32):
        a, _ = run_columns(tape, sh.slot_count(), ik, mat, leaves[0], size=size, zbuf_init=z, more_leaves=leaves[1:])
        b, wb = run_columns(tape, sh.slot_count(), ik, mat, leaves[0], size=size, zbuf_init=z, more_leaves=leaves[1:], column_mode=True, group_log2=g)
        assert (a != z).any()
        assert (a == b).all(), f"{(a != b).sum()} z-buffer words differ"
        assert len(wb) == 64 * (4 >> g)


def test_column_mode_in_the_transcendental_kernel():
    """fh_columns_t has the same column walk (the kernels share their generator): a tape with transcendental opcodes, two leaves in one
    column and one beside it, the words of the layer walk."""
    sh, tape, ik = trans_shape(0)
    size = 32
    leaves = [(8, 16, 24), (8, 16, 8), (16, 16, 16)]
    a, _ = run_columns(tape, sh.slot_count(), ik, AFFINE32, leaves[0], size=size, more_leaves=leaves[1:], kernel="fh_columns_t")
    b, wb = run_columns(tape, sh.slot_count(), ik, AFFINE32, leaves[0], size=size, more_leaves=leaves[1:], kernel="fh_columns_t", column_mode=True)
    assert (a != 0).any() and (a == b).all(), f"{(a != b).sum()} z-buffer words differ"
    assert len(wb) == 64


def _leaf_values(tape, n_regs, ik, mat, leaf=(0, 0, 0), kernel="fh_columns"):
    """run ONE leaf of an 8-voxel-per-lane class and return the tape's output for its 64 pixels x 8 voxels as the kernel left it in
    VT (v18 .. v25: the OUTPUT handler of the compact register map leaves it there; sample j = voxel z + 7 - j), next to numpy's"""
    _, ws = run_columns(tape, n_regs, ik, mat, leaf, kernel=kernel)
    w = max(ws, key=lambda w: w.counts.get("valu", 0))
    got = np.stack([np.asarray(w.v[18 + j]).view(F32) for j in range(8)], axis=1)       # [lane][sample]: VT, where the compact register map leaves the output
    lx, ly, lz = leaf
    want = np.zeros((64, 8), F32)
    for lane in range(64):
        px, py = lx + lane % 8, ly + lane // 8
        zs = np.arange(lz + 7, lz - 1, -1)
        X, Y, Z = xf_point(np.asarray(mat, F32), np.full(8, px), np.full(8, py), zs)
        inputs = {s: (X, Y, Z)[k] for s, k in enumerate(ik) if k < 3}
        want[lane] = U.ref_f32(tape, inputs, 8)[0]
    return got, want


def _same_bits(a, b):
    return ((a.view(U32) == b.view(U32)) | (np.isnan(a) & np.isnan(b))).all()


@pytest.mark.parametrize("mat", [AFFINE, ROTATED], ids=["affine", "rotated"])
def test_delta_handlers_every_op_and_distance(mat):
    """Threaded dispatch, round 6: ops that read one file register and write another have a handler per (op, distance between the two
    registers) - family U: unary / register-immediate ops with out != a, family B: in-place RR ops with b elsewhere (gen_interp.py
    handler_delta).  Every op of both families at distances -7 .. 7, on operands that hold exact zeros of both signs, infinities and
    NaNs in some lanes (min / max by v_minimum3_f32 with the zero guard and its slow path): the output VALUES, bit for bit, for
    every lane and voxel against numpy."""
    P, OP = U.pack, U.OPN
    ik = [0, 1, 2] + [3] * 13
    f = U.f2u
    # registers 0 .. 7: r0 = x, r1 = y, r2 = x - x (+0 everywhere), r3 = -(x - x) (-0), r4 = sqrt(y - 0.3) (NaN where y < 0.3), r5 = 1 / r2 ... built per test
    def prelude():
        return [P(OP["INPUT"], 0, 0, 0), P(OP["INPUT"], 1, 0, 1), P(OP["INPUT"], 7, 0, 2),
                P(OP["SUB_RR"], 2, 0, 0), P(OP["NEG"], 3, 2, 0), P(OP["SUB_RI"], 4, 1, f(0.3)), P(OP["SQRT"], 4, 4, 0),
                P(OP["MUL_RR"], 5, 0, 7), P(OP["FLOOR"], 6, 5, 0), P(OP["SUB_RR"], 5, 5, 6), P(OP["SUB_RI"], 5, 5, f(0.5))]     # r5: a fraction - 0.5, both signs
    tapes = []
    # family U: out = op(a) for every (out, a) pair of distance d among registers holding different kinds of values
    for op, w1 in (("COPY_REG", 0), ("NEG", 0), ("ABS", 0), ("SQUARE", 0), ("ADD_RI", f(0.25)), ("SUB_RI", f(1.5)), ("MUL_RI", f(-3.0)), ("SUB_IR", f(0.75))):
        for out, src in ((6, 5), (5, 6), (7, 0), (0, 7), (6, 3), (3, 6), (6, 4), (2, 5), (6, 2), (1, 4)):
            t = prelude()
            if src == 6:
                t.append(P(OP["ADD_RR"], 6, 5, 1))
            t += [P(OP[op], out, src, w1), P(OP["OUTPUT"], 0, out, 0)]
            tapes.append((f"{op} r{out} <- r{src}", t))
    # family B: a = op(a, b) for distances b - a of both signs; a and b among zeros of both signs, NaNs, ordinary values
    for op in ("ADD_RR", "SUB_RR", "MUL_RR", "MIN_RR", "MAX_RR"):
        for ra, rb in ((5, 1), (1, 5), (2, 3), (3, 2), (2, 5), (5, 2), (3, 5), (4, 5), (5, 4), (0, 7), (7, 0), (4, 2), (3, 4)):
            t = prelude() + [P(OP[op], ra, ra, rb), P(OP["OUTPUT"], 0, ra, 0)]
            tapes.append((f"{op} r{ra} <- r{ra}, r{rb}", t))
    bad = []
    for name, t in tapes:
        got, want = _leaf_values(np.array(t, np.uint64), 8, ik, mat)
        if not _same_bits(got, want):
            bad.append((name, int(((got.view(U32) != want.view(U32)) & ~(np.isnan(got) & np.isnan(want))).sum())))
    assert not bad, bad
    # the operands did hold what the cases are about (zeros of both signs and NaNs among the voxels)
    g, w = _leaf_values(np.array(prelude() + [P(OP["MIN_RR"], 3, 3, 2), P(OP["OUTPUT"], 0, 3, 0)], np.uint64), 8, ik, mat)
    assert _same_bits(g, w) and (w.view(U32) == 0).all()                     # min(-0, +0) = +0: the guard's slow path
    g, w = _leaf_values(np.array(prelude() + [P(OP["OUTPUT"], 0, 4, 0)], np.uint64), 8, ik, mat)
    assert np.isnan(w).any() and not np.isnan(w).all()


@pytest.mark.parametrize("mat", [AFFINE, PERSPECTIVE], ids=["affine", "perspective"])
def test_generic_binary_ops_in_the_ten_register_file(mat):
    """fh_columns' compact register map (round 6): a file of ten registers of eight voxels, the handlers' results built in place over
    their first operand (no separate result registers).  Every two-operand op without a form of its own - div, compare, and, or, min /
    max with out != a - in the RR, RI and IR forms, between registers 0 .. 9, on operands with zeros of both signs, infinities and
    NaNs: the output values of every lane and voxel against numpy."""
    P, OP = U.pack, U.OPN
    ik = [0, 1, 2] + [3] * 13
    f = U.f2u
    def prelude():       # r0 = x, r1 = y, r9 = z, r2 = +0, r3 = -0, r4 = sqrt(y - 0.3) (NaN where y < 0.3), r5 = a fraction - 0.5, r8 = 1 / r2 = inf
        return [P(OP["INPUT"], 0, 0, 0), P(OP["INPUT"], 1, 0, 1), P(OP["INPUT"], 9, 0, 2),
                P(OP["SUB_RR"], 2, 0, 0), P(OP["NEG"], 3, 2, 0), P(OP["SUB_RI"], 4, 1, f(0.3)), P(OP["SQRT"], 4, 4, 0),
                P(OP["MUL_RR"], 5, 0, 9), P(OP["FLOOR"], 6, 5, 0), P(OP["SUB_RR"], 5, 5, 6), P(OP["SUB_RI"], 5, 5, f(0.5)),
                P(OP["RECIP"], 8, 2, 0), P(OP["NEG"], 7, 8, 0)]
    bad = []
    for base in ("DIV", "COMPARE", "AND", "OR", "MIN", "MAX"):
        for out, ra, rb in ((6, 5, 1), (9, 1, 5), (6, 2, 3), (6, 3, 2), (0, 4, 5), (9, 5, 4), (6, 8, 7), (6, 7, 8), (9, 8, 2), (6, 5, 5)):
            cases = [(f"{base}_RR r{out} <- r{ra}, r{rb}", P(OP[base + "_RR"], out, ra, rb))]
            for imm in (0.0, -0.0, 0.25, float("inf")):
                cases.append((f"{base}_RI r{out} <- r{ra}, {imm}", P(OP[base + "_RI"], out, ra, f(imm))))
                if base + "_IR" in OP:
                    cases.append((f"{base}_IR r{out} <- {imm}, r{ra}", P(OP[base + "_IR"], out, ra, f(imm))))
            for name, op in cases[:3] if out != 6 else cases:
                t = prelude() + [op, P(OP["OUTPUT"], 0, out, 0)]
                got, want = _leaf_values(np.array(t, np.uint64), 10, ik, mat)
                if not _same_bits(got, want):
                    bad.append((name, int(((got.view(U32) != want.view(U32)) & ~(np.isnan(got) & np.isnan(want))).sum())))
    assert not bad, bad


def test_hand_written_expf_in_the_transcendental_kernel():
    """fh_columns_t's EXP handler holds expf itself (gen_trans.py exp_pair: glibc's operations in binary64 by hand, the 2^(i/32) table in
    two VGPRs) - executed here instruction by instruction (the emulator's v_fma_f64 is exact rational arithmetic) against the host
    libm, bit for bit: arguments of every size - fractions, tens, beyond +-88 (the special cases: the compiled routine, sample by
    sample, for the whole op), infinities, NaNs, zeros of both signs, subnormals."""
    P, OP = U.pack, U.OPN
    ik = [0, 1, 2] + [3] * 13
    f = U.f2u
    bad = []
    for scale in (0.37, 3.0, 41.0, 87.9, 95.0, 300.0, 1e30, 1e-40, 0.0, -0.0):
        t = [P(OP["INPUT"], 0, 0, 0), P(OP["INPUT"], 1, 0, 2), P(OP["MUL_RR"], 0, 0, 1), P(OP["MUL_RI"], 0, 0, f(scale)), P(OP["EXP"], 2, 0, 0), P(OP["OUTPUT"], 0, 2, 0)]
        got, want = _leaf_values(np.array(t, np.uint64), 3, ik, ROTATED, kernel="fh_columns_t")
        if not _same_bits(got, want):
            bad.append((scale, int(((got.view(U32) != want.view(U32)) & ~(np.isnan(got) & np.isnan(want))).sum())))
    # ... and of NaN (sqrt of a negative number in some lanes) and of infinities of either sign (x / 0)
    for sign in ("COPY_REG", "NEG"):
        t = [P(OP["INPUT"], 0, 0, 0), P(OP["INPUT"], 1, 0, 1), P(OP["SUB_RI"], 1, 1, f(0.3)), P(OP["SQRT"], 1, 1, 0), P(OP["EXP"], 2, 1, 0),
             P(OP["SUB_RR"], 3, 0, 0), P(OP["RECIP"], 3, 3, 0), P(OP["MUL_RR"], 3, 3, 0), P(OP[sign], 3, 3, 0), P(OP["EXP"], 4, 3, 0), P(OP["OUTPUT"], 0, 4, 0)]
        got, want = _leaf_values(np.array(t, np.uint64), 5, ik, ROTATED, kernel="fh_columns_t")
        assert (np.isinf(want).any() if sign == "NEG" else (want == 0).any())
        if not _same_bits(got, want):
            bad.append(("inf " + sign, int(((got.view(U32) != want.view(U32)) & ~(np.isnan(got) & np.isnan(want))).sum())))
        got, want = _leaf_values(np.array(t[:5] + [P(OP["OUTPUT"], 0, 2, 0)], np.uint64), 5, ik, ROTATED, kernel="fh_columns_t")
        assert np.isnan(want).any() and np.isfinite(want).any()
        if not _same_bits(got, want):
            bad.append(("nan", int(((got.view(U32) != want.view(U32)) & ~(np.isnan(got) & np.isnan(want))).sum())))
    assert not bad, bad


def test_hand_written_logf_in_the_transcendental_kernel():
    """... and the LN handler's logf (gen_trans.py ln_pair: the {1/c, log c} table in four VGPRs), the same way: ordinary arguments of every
    size, 1 exactly, and the special ones - zero, negative, subnormal, infinite, NaN - that send the whole op to the compiled routine."""
    P, OP = U.pack, U.OPN
    ik = [0, 1, 2] + [3] * 13
    f = U.f2u
    bad = []
    for scale, shift in ((0.37, 1.3), (3.0, 5.0), (1e-20, 1e-19), (1e30, 2e30), (1e-38, 1.2e-38), (0.0, 1.0), (1.0, 0.0), (1e-42, 3e-42), (1e38, float("inf")), (-1.0, -3.0)):
        t = [P(OP["INPUT"], 0, 0, 0), P(OP["INPUT"], 1, 0, 2), P(OP["MUL_RR"], 0, 0, 1), P(OP["MUL_RI"], 0, 0, f(scale)), P(OP["ADD_RI"], 0, 0, f(shift)), P(OP["LN"], 2, 0, 0),
             P(OP["OUTPUT"], 0, 2, 0)]
        got, want = _leaf_values(np.array(t, np.uint64), 3, ik, ROTATED, kernel="fh_columns_t")
        if not _same_bits(got, want):
            bad.append((scale, shift, int(((got.view(U32) != want.view(U32)) & ~(np.isnan(got) & np.isnan(want))).sum())))
    # NaN arguments (sqrt of a negative number in some lanes)
    t = [P(OP["INPUT"], 0, 0, 0), P(OP["INPUT"], 1, 0, 1), P(OP["SUB_RI"], 1, 1, f(0.3)), P(OP["SQRT"], 1, 1, 0), P(OP["LN"], 2, 1, 0), P(OP["OUTPUT"], 0, 2, 0)]
    got, want = _leaf_values(np.array(t, np.uint64), 3, ik, ROTATED, kernel="fh_columns_t")
    assert np.isnan(want).any() and np.isfinite(want).any()
    if not _same_bits(got, want):
        bad.append(("nan", int(((got.view(U32) != want.view(U32)) & ~(np.isnan(got) & np.isnan(want))).sum())))
    assert not bad, bad


@pytest.mark.parametrize("kernel", ["fh_columns", "fh_columns_t"])
def test_division_by_an_immediate(kernel):
    """DIV_RI's short sequence (gen_interp.py f_div_imm: the reciprocal's refinement once per op, the quotient's two samples per
    instruction, no scaling when both operands lie within 2^-40 .. 2^40) against IEEE division, bit for bit: divisors of every kind -
    not powers of two, negative, tiny, huge (those take the general sequence), numerators from 1e-13 to 1e13, and ops whose
    numerators hold a zero, an infinity or a NaN in some lane (the general sequence for the whole op)."""
    P, OP = U.pack, U.OPN
    ik = [0, 1, 2] + [3] * 13
    f = U.f2u
    bad = []
    for imm in (0.3, 3.0, -7.77, 1e-5, 123456.7, float.fromhex("0x1.fffffep+39"), 2.0 ** -40, 1.5 * 2.0 ** -41, 2.0 ** 41, 1e-30, 1e30, 1.0, -0.5):
        for scale in (1.0, 1e-12, 1e11, 3e12, 1e-13):
            t = [P(OP["INPUT"], 0, 0, 0), P(OP["INPUT"], 1, 0, 2), P(OP["MUL_RR"], 0, 0, 1), P(OP["ADD_RI"], 0, 0, f(0.013)), P(OP["MUL_RI"], 0, 0, f(scale)),
                 P(OP["DIV_RI"], 2, 0, f(imm)), P(OP["OUTPUT"], 0, 2, 0)]
            got, want = _leaf_values(np.array(t, np.uint64), 3, ik, ROTATED, kernel=kernel)
            if not _same_bits(got, want):
                bad.append((imm, scale, int(((got.view(U32) != want.view(U32)) & ~(np.isnan(got) & np.isnan(want))).sum())))
    # zeros (x - x), infinities (1 / 0) and NaNs (sqrt of a negative number) among the numerators, in place as well
    for pre in ([P(OP["SUB_RR"], 0, 0, 0)], [P(OP["SUB_RR"], 0, 0, 0), P(OP["RECIP"], 0, 0, 0)], [P(OP["SUB_RI"], 0, 0, f(0.3)), P(OP["SQRT"], 0, 0, 0)]):
        t = [P(OP["INPUT"], 0, 0, 1)] + pre + [P(OP["DIV_RI"], 0, 0, f(0.7)), P(OP["OUTPUT"], 0, 0, 0)]
        got, want = _leaf_values(np.array(t, np.uint64), 1, ik, ROTATED, kernel=kernel)
        if not _same_bits(got, want):
            bad.append(("special", len(pre), int(((got.view(U32) != want.view(U32)) & ~(np.isnan(got) & np.isnan(want))).sum())))
    assert not bad, bad


@pytest.mark.parametrize("fn", ["SIN", "COS"])
def test_hand_written_sinf_cosf_in_the_transcendental_kernel(fn):
    """... and the SIN / COS handlers' sinf / cosf (gen_trans.py sincos_pair: glibc's fast reduction and both polynomials in binary64):
    arguments in every quadrant and of every size below 120, tiny ones (|y| < 2^-12: y, 1), zeros of both signs, NaNs; 120 and
    beyond, infinities: the compiled routine for the whole op."""
    P, OP = U.pack, U.OPN
    ik = [0, 1, 2] + [3] * 13
    f = U.f2u
    bad = []
    for scale, shift in ((0.37, 0.0), (3.0, 0.5), (25.0, -7.0), (119.0, 0.0), (1e-4, 0.0), (1e-5, 2e-4), (0.0, 0.0), (-0.0, 0.0), (1e-30, 0.0), (130.0, 0.0), (1e6, 0.0), (1e30, 0.0),
                         (1.0, 1.5707964), (1.0, 3.1415927), (0.001, 0.7853982)):
        t = [P(OP["INPUT"], 0, 0, 0), P(OP["INPUT"], 1, 0, 2), P(OP["MUL_RR"], 0, 0, 1), P(OP["MUL_RI"], 0, 0, f(scale)), P(OP["ADD_RI"], 0, 0, f(shift)), P(OP[fn], 2, 0, 0),
             P(OP["OUTPUT"], 0, 2, 0)]
        got, want = _leaf_values(np.array(t, np.uint64), 3, ik, ROTATED, kernel="fh_columns_t")
        if not _same_bits(got, want):
            bad.append((scale, shift, int(((got.view(U32) != want.view(U32)) & ~(np.isnan(got) & np.isnan(want))).sum())))
    for pre in ([P(OP["SUB_RI"], 0, 0, f(0.3)), P(OP["SQRT"], 0, 0, 0)], [P(OP["SUB_RR"], 0, 0, 0), P(OP["RECIP"], 0, 0, 0)], [P(OP["SUB_RR"], 0, 0, 0), P(OP["NEG"], 0, 0, 0)]):
        t = [P(OP["INPUT"], 0, 0, 1)] + pre + [P(OP[fn], 1, 0, 0), P(OP["OUTPUT"], 0, 1, 0)]       # NaN in some lanes; infinities; -0
        got, want = _leaf_values(np.array(t, np.uint64), 2, ik, ROTATED, kernel="fh_columns_t")
        if not _same_bits(got, want):
            bad.append(("special", len(pre), int(((got.view(U32) != want.view(U32)) & ~(np.isnan(got) & np.isnan(want))).sum())))
    assert not bad, bad
