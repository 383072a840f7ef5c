#!/usr/bin/env python3
"""Generator of the hand-scheduled gfx950 (CDNA4) assembly interpreters of fidget-hip.

Why assembly: a tape interpreter is one indirect jump per operation.  The AMDGPU backend of
clang has neither jump tables nor computed goto, so a C++ `switch` becomes a compare/branch
tree (~20 taken branches per tape op) and every dynamically indexed write to a register-file
array in VGPRs is preceded by a copy of the whole array.  Here the dispatch is one
`s_setpc_b64` into a table of fixed-size handlers and the register file is addressed in place
with `s_set_gpr_idx_on` (M0-relative VGPR operands).

Kernels emitted (f32 point evaluation of a tape, fidget-core/src/vm/mod.rs:788-1049 semantics,
bit-exact with the C++ kernels in kernels.hip which they replace on their fast path):

  fh_columns_{NR}x{ZB}     3D leaf stage: one 8x8 pixel footprint per wave, leaves front to
                           back, ZB voxels per lane per pass (same algorithm as k_leaves3d)
  fh_float_eval_{NR}x{ZB}  BulkEvaluator<f32>: 64*ZB samples per wave

NR = registers of the VGPR register file, ZB = samples per lane.  The file occupies
v[FILE .. FILE+NR*ZB): register r of sample slot j is v[FILE + r*ZB + j] - the ZB samples of a
register are consecutive, so that copies and add / sub / mul take them two at a time
(v_pk_mov_b32, v_pk_add_f32, v_pk_mul_f32: 64-bit operands, M0-relative like any other), and an
op whose output register is one of its operands (more than half of a pruned tape's ops) works on
the file in place, with no copy at all.

Dispatch.  What bounds these kernels is the scalar unit, which all the waves of a CU share
(PMC: 36 scalar instructions per tape op in the first version): the vector work of an op is issued
beside it for free.  So the decode exists in eight copies, one per slot of the two 4-op SGPR
batches the tape comes in by (no queue shifting, no counters; a handler ends with a jump to the
next copy), the handler table has an "in place" half that the decode selects when out == a, and
the leaf kernel, whose tapes are short, takes a whole tape of up to 64 ops with ONE vector load,
lane = op, decodes it with vector code into three VGPRs (handler address, file indices of out and
a in one, word 1) and dispatches with three v_readlane and a jump.  Shape tapes have one output and it is
their last op: the OUTPUT handler of the leaf kernel returns to the caller, nothing is counted.

Tape format: tape_format.h (8 bytes per op: opcode | out<<8 | a<<20, then b / imm / slot).
Tapes with transcendental, modulo or rng ops, more than 32 registers, or a projective
screen-to-model matrix stay on the C++ kernels.

usage: gen_interp.py offsets.json out.s
"""
import json
import os
import sys

EXP = os.environ.get("FH_EXP", "")     # experiments (tools/build_variant.py): never set in a product build

# ---- opcodes (tape_format.h) ---------------------------------------------------------------
OPS = ["OUTPUT", "INPUT", "COPY_REG", "COPY_IMM",
       "NEG", "ABS", "RECIP", "SQRT", "SQUARE", "FLOOR", "CEIL", "ROUND", "SIN", "COS", "TAN", "ASIN", "ACOS", "ATAN",
       "EXP", "LN", "NOT", "RAND"]
BIN = ["ADD", "SUB", "MUL", "DIV", "ATAN2", "COMPARE", "MIX", "MOD", "MIN", "MAX", "AND", "OR"]
OPS += [b + "_RR" for b in BIN] + [b + "_RI" for b in BIN] + [b + "_IR" for b in ("SUB", "DIV", "ATAN2", "COMPARE", "MIX", "MOD")]
assert len(OPS) == 52
UNSUPPORTED = {"SIN", "COS", "TAN", "ASIN", "ACOS", "ATAN", "EXP", "LN", "RAND", "ATAN2", "MIX", "MOD"}

HSTRIDE_LOG2 = 7  # handler slots of 128 bytes

# ---- fixed SGPRs ---------------------------------------------------------------------------
S_KERNARG = "s[0:1]"
S_WG = "s2"
S_STATE = "s[4:5]"
S_MAT = 8               # s[8:23]    screen -> model matrix, row major
S_WIDTH, S_HEIGHT = "s24", "s25"
S_LAYERS, S_FW = "s26", "s27"
S_SIGN = "s28"          # 0x80000000
S_ABSM = "s29"          # 0x7fffffff
S_ARENA = "s[30:31]"
S_NXTV = "s32"           # columns: the next leaf's tape is on its way into V_NXT
S_SLABZ = "s61"          # columns: z of the slab's first voxel
S_TABLE = "s[34:35]"
S_ZBUF = "s[36:37]"
S_FPLIST = "s[38:39]"
S_NFP = "s40"
S_WI = "s41"
S_HBASE = "s[42:43]"    # address of handler 0
S_TAPE = "s[44:45]"     # interpreter argument: first op
S_LEN = "s46"           # interpreter argument: ops left
S_LEN0 = "s47"
S_QA = 48               # s[48:55]  current batch of 4 ops
S_QB = 56               # s[56:63]  next batch
S_W0, S_W1 = "s64", "s65"
S_CUR = "s[64:65]"
S_T0 = "s66"
S_OUT = "s67"
S_A = "s68"
S_T1 = "s69"
S_NEXT = "s[70:71]"      # address of the decode copy of the next op
S_NX, S_NX_LO, S_NX_HI = "s[48:49]", "s48", "s49"   # threaded dispatch: the NEXT op's out | a << 8 and word 1 (the scalar batches are not used then)
S_JN, S_JN_LO, S_JN_HI = "s[50:51]", "s50", "s51"   # ... and its handler's address
S_TCUR = "s[52:53]"     # threaded dispatch, tapes of more than 64 ops: the next chunk's first op
S_REM = "s54"           # ... and the ops from there to the end of the tape
S_LONG = "s[56:57]"     # ... and the address of the code that fetches and decodes a chunk (the NEXT_CHUNK pseudo-op's target)
S_JMP = "s[44:45]"       # handler address (the interpreter's tape argument is consumed by then)
S_FETCH = "s[72:73]"
S_RET = "s[74:75]"
S_FX, S_FY = "s76", "s77"
S_ID = "s78"
S_LZ = "s79"
S_LAYMASK = "s[80:81]"
S_PEND = "s[82:83]"
S_TBASE = "s[84:85]"
S_PC = "s[86:87]"
S_SAVE = "s[88:89]"
S_M = [f"s[{90 + 2 * j}:{91 + 2 * j}]" for j in range(4)]   # per-slot lane masks s[90:97]
S_K = "s98"
S_ZL = "s99"
S_INV = "s99"            # columns: the leaf's tape reads no input that changes along a pixel column (S_ZL is dead by then)
S_DEPMASK = "s39"        # columns: input slots that change along a pixel column
S_HB = "s[100:101]"       # columns: handler table of the leaf's register class
S_PROJFLAG = "s101"       # columns: bit 16 = projective screen-to-model matrix (row 3 != 0 0 0 1); (low half: workgroup id y)
S_SLOTX, S_SLOTY, S_SLOTZ, S_RC = "s0", "s1", "s3", "s2"   # columns: input slots of x, y, z; regs | choices << 16
S_N = "s6"              # bulk: number of samples
S_ACT = [f"s[{8 + 2 * j}:{9 + 2 * j}]" for j in range(4)]  # bulk: lanes of slot j holding a sample
S_VARS = "s[30:31]"     # bulk
S_OUTP = "s[32:33]"     # bulk

# ---- fixed VGPRs (ZB <= 8) ------------------------------------------------------------------
V_LANE = "v0"
V_PXF, V_PYF = "v1", "v6"
V_PIX = "v2"                # byte offset of this lane's pixel in the z-buffer
V_HIT, V_DEPTH = "v4", "v5"     # stored together as the 64-bit z-buffer word
V_IDS = "v54"            # (= V_ENT[0])
V_AX, V_AY, V_AZ = "v8", "v9", "v59"
VRES = [f"v{10 + j}" for j in range(8)]
VT = [f"v{18 + j}" for j in range(8)]
VU = [f"v{26 + j}" for j in range(8)]
VW = [f"v{34 + j}" for j in range(8)]
VD = [f"v{42 + i}" for i in range(8)]   # scratch of the division / sqrt sequences
V_PINF, V_NINF = "s55", "s60"    # columns (threaded): +inf / -inf, the neutral first operand of v_minimum3_f32 / v_maximum3_f32 in the delta handlers (scalar: src0 of a VOP3)
V_LUT1, V_LUT2 = "v3", "v62"     # columns (threaded): decode tables, lane = opcode (LUT_BITS below) / lane = input slot (that INPUT op's handler offset); bit 30 of S_WGY: loaded
S_ONES = "s[58:59]"              # columns (threaded): (1.0, 1.0)
V_QNAN = "v50"
V_SQRTC = "v51"
V_NXT = ("v52", "v53")   # columns: the NEXT leaf's tape words, requested while this leaf is interpreted
V_S0, V_S1, V_S2, V_S3 = "v42", "v43", "v44", "v45"   # set-up temporaries (the division / sqrt scratch: dead between leaves)
V_S4 = "v46"
V_ENT = ("v54", "v55", "v56", "v57")   # columns: the block's leaf table entries, lane = footprint: id + 1, tape offset, length | regs << 24, x | y << 16
V_IDV = "v58"
V_AW = "v7"              # columns: m[12] * x + m[13] * y of this lane's pixel
VOFF = [f"v{60 + j}" for j in range(4)]  # bulk: byte offset of sample j
V_DEC = ["v60", "v61", "v62", "v63"]   # columns: the leaf's tape decoded, lane = op: handler address, out index, a index, word 1
FILE = 64

SRC0, SRC1, SRC2, DST = 1, 2, 4, 8

# Two maps of the leaf kernels' vector registers.  The default (above): 64 fixed registers + a file from v64 - the bulk kernels, the
# normals kernels and fh_columns_t (file of 128 + the routines' window).  COMPACT (fh_columns, round 6): 48 fixed + a file of 80 from v48 =
# 128 registers, four waves per SIMD as before, but the file holds TEN registers of eight voxels: leaves of 9 and 10 registers (12 % of
# prospero.vm's, a quarter of the kernel's dispatches) take one pass instead of two of four voxels.  The sixteen registers come from
# VRES and VW: the OUTPUT handler leaves the result in VT, and the handlers that built a result in VW build it in place over their
# first operand (VW is VT here - the places where that needed another order of instructions test `VW is VT`).
_MAPS = {
    "default": dict(FILE=64, VRES=[f"v{10 + j}" for j in range(8)], VT=[f"v{18 + j}" for j in range(8)], VU=[f"v{26 + j}" for j in range(8)],
                    VW=[f"v{34 + j}" for j in range(8)], VD=[f"v{42 + i}" for i in range(8)], V_QNAN="v50", V_SQRTC="v51", V_NXT=("v52", "v53"),
                    V_ENT=("v54", "v55", "v56", "v57"), V_IDV="v58", V_AZ="v59", V_DEC=["v60", "v61", "v62", "v63"], V_LUT1="v3", V_LUT2="v62"),
    "compact": dict(FILE=48, V_QNAN="v38", V_SQRTC="v39", V_NXT=("v14", "v15"), V_ENT=("v34", "v35", "v36", "v37"), V_IDV="v16", V_AZ="v17",
                    V_DEC=["v10", "v11", "v13", "v12"], V_LUT1="v3", V_LUT2="v13", VD=[f"v{42 + i}" for i in range(6)] + ["v40", "v41"]),
}


def set_reg_map(name):
    g = globals()
    m = dict(_MAPS["default"])
    m.update(_MAPS[name])
    if name == "compact":
        m["VRES"] = m["VW"] = m["VT"]
    g.update(m)
    g["REG_MAP"] = name


REG_MAP = "default"


class Asm:
    def __init__(self):
        self.lines = []
        self.uid = 0

    def __call__(self, s=""):
        for ln in s.strip("\n").split("\n"):
            self.lines.append(ln.rstrip())

    def label(self, stem):
        self.uid += 1
        return f".L{stem}_{self.uid}"

    def text(self):
        return "\n".join(self.lines) + "\n"


def hexf(x):
    import struct
    return "0x%08x" % struct.unpack("<I", struct.pack("<f", x))[0]


class Interp:
    """One interpreter instance (dispatch loop + handlers) for a given NR x ZB and I/O kind."""

    def __init__(self, a, name, nr, zb, kind, off, trans=False):
        self.a, self.name, self.nr, self.zb, self.kind, self.off = a, name, nr, zb, kind, off
        self.trans = trans   # handlers for the transcendental / modulo / rng opcodes (they call the routines of gen_trans.py)
        self.t_base = 128    # ... whose register window starts here (behind the register file)
        self.wide_trans = False   # the four-sample routines are embedded too (window of gen_trans.WIDE_V registers)
        self.exp2 = False         # EXP / LN / SIN / COS by the hand-written two-sample routines (gen_trans.exp_pair ...) inside the handlers
        self.sincos2 = False      # ... SIN / COS alone (the bulk kernels: no table registers)
        self.t_prefix = "fh_t_"   # ... and whose labels start with this (every kernel embeds its own copies)
        self.lg = {2: 1, 4: 2, 8: 3}[zb]
        self.hl = HSTRIDE_LOG2
        self.next = f".L{name}_next"
        self.ool = []  # out-of-line handler bodies: (label, callable)
        # Threaded dispatch (the leaf kernel, round 6): there is no dispatcher to return to - a handler starts by reading the NEXT op's
        # three decoded words out of the lane that holds it (behind its own vector work in the pipeline: the v_readlane -> scalar use
        # latency, which was a third of a lone wave's time per op, is off the critical path) and ends with ONE jump, straight into the
        # next op's handler.  This op's words were read by the handler before it, into S_NX / S_JN.
        self.threaded = kind == "columns"
        self.s_out = S_W0 if self.threaded else S_OUT     # bits 7:0: file index of `out` (threaded: word 0 of S_CUR is out | a << 8)
        self.s_a = S_A                                    # file index of `a` (threaded: handlers that need it shift it out of s_out)
        if self.threaded:
            self.hl = 8            # slots of 256 bytes: the prologue, and min / max in place without a jump to their body
        self.ip_slot_log2 = self.hl
        self.delta_setup()

    def F(self, j):
        """sample j of the register selected by the index (M0 = register * ZB)"""
        return f"v{FILE + j}"

    def FP(self, k):
        return f"v[{FILE + 2 * k}:{FILE + 2 * k + 1}]"

    @staticmethod
    def P(regs, k):
        n = int(regs[2 * k][1:])
        return f"v[{n}:{n + 1}]"

    # -- small helpers ---------------------------------------------------------------------
    def idx_on(self, sreg, mode):
        self.a(f"\ts_set_gpr_idx_on {sreg}, {mode}")

    def idx_idx(self, sreg):
        self.a(f"\ts_set_gpr_idx_idx {sreg}")

    def idx_off(self):
        self.a("\ts_set_gpr_idx_off")

    def b_index(self):
        """-> the SGPR whose bits 7:0 are the file index of the register named by word 1 (threaded: the decode has shifted it)"""
        if self.threaded:
            return S_W1
        self.a(f"\ts_lshl_b32 {S_T1}, {S_W1}, {self.lg}")
        return S_T1

    def prologue(self, need_a=False, prefetch=True):
        """threaded dispatch: this op's words S_NX -> S_CUR, the next op's words -> S_NX / S_JN.  (v_readlane honours SRC0-relative
        index mode and no other - tools/probe_isa.py -: no handler leaves through ret() with SRC0-relative mode on.)"""
        if not self.threaded:
            return
        a = self.a
        if prefetch:
            a(f"""
	s_mov_b64 {S_CUR}, {S_NX}
	v_readlane_b32 {S_JN_LO}, {V_DEC[0]}, {S_LEN}
	v_readlane_b32 {S_NX_LO}, {V_DEC[1]}, {S_LEN}
	v_readlane_b32 {S_NX_HI}, {V_DEC[3]}, {S_LEN}
	s_add_u32 {S_LEN}, {S_LEN}, 1""")
        else:
            a(f"\ts_mov_b64 {S_CUR}, {S_NX}")
        if need_a:
            a(f"\ts_lshr_b32 {S_A}, {S_W0}, 8")

    def pk_mov(self, dst, src):
        self.a(f"\tv_pk_mov_b32 {dst}, {src}, {src} op_sel:[0,1]")

    def read_a(self, dst):
        """dst = file[a]; leaves the index mode on (SRC0 | SRC1 relative)."""
        self.idx_on(self.s_a, SRC0 | SRC1)
        for k in range(self.zb // 2):
            self.pk_mov(self.P(dst, k), self.FP(k))

    def read_b(self, dst, already_on=True):
        sb = self.b_index()
        if already_on:
            self.idx_idx(sb)
        else:
            self.idx_on(sb, SRC0 | SRC1)
        for k in range(self.zb // 2):
            self.pk_mov(self.P(dst, k), self.FP(k))

    def imm_b(self, dst):
        for j in range(self.zb):
            self.a(f"\tv_mov_b32 {dst[j]}, {S_W1}")

    def ret(self, src0_on=False):
        """end of a handler.  src0_on: the index mode in force is SRC0-relative (threaded: switched off, the next handler starts
        with v_readlane)"""
        if self.threaded:
            if src0_on:
                self.idx_off()
            return self.a(f"\ts_setpc_b64 {S_JN}")
        self.a(f"\ts_setpc_b64 {S_NEXT}")

    def call(self, fn):
        """call the embedded routine fh_t_<fn> (gen_trans.py): argument(s) v128 (, v129), result v128, return address s[96:97];
        clobbers v128..v153, s86..s97 and vcc (the mask scratch of the handlers: nothing live)"""
        here, ret = self.a.label("call"), self.a.label("ret")
        if self.threaded:       # (the handler tables of the threaded dispatch put the routines beyond s_branch's 128 KB: a computed jump)
            far = self.a.label("fcall")
            return self.a(f"""
	s_getpc_b64 s[96:97]
{here}:
	s_add_u32 s96, s96, {ret} - {here}
	s_addc_u32 s97, s97, 0
	s_getpc_b64 s[70:71]
{far}:
	s_mov_b32 s72, {self.t_prefix}{fn} - {far}
	s_ashr_i32 s73, s72, 31
	s_add_u32 s70, s70, s72
	s_addc_u32 s71, s71, s73
	s_setpc_b64 s[70:71]
{ret}:""")
        self.a(f"""
	s_getpc_b64 s[96:97]
{here}:
	s_add_u32 s96, s96, {ret} - {here}
	s_addc_u32 s97, s97, 0
	s_branch {self.t_prefix}{fn}
{ret}:""")

    def pcg_consts(self):
        self.a("\ts_mov_b32 s90, 747796405\n\ts_mov_b32 s91, 0xac564b05\n\ts_mov_b32 s92, 277803737")

    def pcg(self, x, r):
        """r = rng::hash(x) (rng/mod.rs:8-13, the PCG output permutation); r may be x; scratch {VD[0]}, {VD[1]}"""
        t0, t1 = VD[0], VD[1]
        self.a(f"""
	v_mul_lo_u32 {t0}, {x}, s90
	v_add_u32 {t0}, s91, {t0}
	v_lshrrev_b32 {t1}, 28, {t0}
	v_add_u32 {t1}, 4, {t1}
	v_lshrrev_b32 {t1}, {t1}, {t0}
	v_xor_b32 {t1}, {t1}, {t0}
	v_mul_lo_u32 {t1}, {t1}, s92
	v_lshrrev_b32 {t0}, 22, {t1}
	v_xor_b32 {r}, {t0}, {t1}""")

    def write_out(self, src, done=True):
        """file[out] = src; ends the handler."""
        self.idx_on(self.s_out, DST)
        for k in range(self.zb // 2):
            self.pk_mov(self.FP(k), self.P(src, k))
        if done:
            self.ret()

    def mask_gap(self):
        """VALU-written SGPR read as a lane mask by a VALU op needs 2 wait states (gfx940+)."""
        if self.zb < 3:
            self.a(f"\ts_nop {2 - self.zb}")

    # -- arithmetic on plain VGPR operands (index mode off) ------------------------------------
    def mask_pass(self, cmp, sel):
        """per sample: a compare into a lane mask, then a select on it; 4 masks at a time"""
        z = list(range(self.zb))
        for g in (z[i:i + 4] for i in range(0, len(z), 4)):
            for j in g:
                self.a("\t" + cmp(j, S_M[j % 4]))
            if len(g) < 3:
                self.a(f"\ts_nop {2 - len(g)}")   # VALU-written mask -> VALU read: 2 wait states
            for j in g:
                self.a("\t" + sel(j, S_M[j % 4]))

    def zero_guard(self, A, t, plain=True):
        """vcc = one of the samples A is a zero (either sign); t: scratch.  (v_min3_f32 drops a NaN operand: V_QNAN pads the first.)"""
        a, zb = self.a, self.zb
        a(f"\tv_min3_f32 {t}, {V_QNAN}, |{A[0]}|, |{A[1]}|")
        for j in range(2, zb, 2):
            a(f"\tv_min3_f32 {t}, {t}, |{A[j]}|, |{A[j + 1]}|")
        a(f"\tv_cmp_eq_f32_e64 vcc, {t}, 0")

    def f_minmax(self, is_min, A, B, R):
        # min: a < b ? a : b, max: a > b ? a : b; either NaN -> NaN  (dev_ops.hpp f_min / f_max).
        # gfx950's v_minimum3_f32 / v_maximum3_f32 (IEEE-754-2019 minimum / maximum: a NaN operand wins) return exactly that but for
        # min(-0, +0) = -0 (wanted: b, +0) and max(+0, -0) = +0 (wanted: -0) - every pair of special values and 4 M random pairs,
        # tools/probe_minimum3.cpp -: one instruction per sample unless a sample of `a` is a zero, which takes the compares and selects.
        a = self.a
        slow, done = a.label("mm_zero"), a.label("mm_done")
        self.zero_guard(A, VD[6])
        a(f"\ts_cbranch_vccnz {slow}")
        for j in range(self.zb):
            a(f"\t{'v_minimum3_f32' if is_min else 'v_maximum3_f32'} {R[j]}, {A[j]}, {B[j]}, {B[j]}")
        a(f"\ts_branch {done}\n{slow}:")
        cmp = "v_cmp_lt_f32_e64" if is_min else "v_cmp_gt_f32_e64"
        if R is A or R is B:        # (the compact register map: both tests of a sample before its selects)
            for j in range(self.zb):
                a(f"""
	{cmp} {S_M[0]}, {A[j]}, {B[j]}
	v_cmp_u_f32_e64 {S_M[1]}, {A[j]}, {B[j]}
	s_nop 1
	v_cndmask_b32_e64 {R[j]}, {B[j]}, {A[j]}, {S_M[0]}
	v_cndmask_b32_e64 {R[j]}, {R[j]}, {V_QNAN}, {S_M[1]}""")
        else:
            self.mask_pass(lambda j, m: f"{cmp} {m}, {A[j]}, {B[j]}", lambda j, m: f"v_cndmask_b32_e64 {R[j]}, {B[j]}, {A[j]}, {m}")
            self.mask_pass(lambda j, m: f"v_cmp_u_f32_e64 {m}, {A[j]}, {B[j]}", lambda j, m: f"v_cndmask_b32_e64 {R[j]}, {R[j]}, {V_QNAN}, {m}")
        a(f"{done}:")

    def f_andor(self, is_and, A, B, R):
        # and: a == 0 ? a : b ; or: a != 0 ? a : b
        cmp = "v_cmp_eq_f32_e64" if is_and else "v_cmp_neq_f32_e64"
        self.mask_pass(lambda j, m: f"{cmp} {m}, 0, {A[j]}", lambda j, m: f"v_cndmask_b32_e64 {R[j]}, {B[j]}, {A[j]}, {m}")

    def f_compare(self, A, B, R):
        # a < b ? -1 : (a == b ? 0 : (a > b ? 1 : NaN))
        if R is A or R is B:        # (the compact register map: the result over an operand - a sample's three tests before its selects)
            for j in range(self.zb):
                self.a(f"""
	v_cmp_gt_f32_e64 {S_M[0]}, {A[j]}, {B[j]}
	v_cmp_eq_f32_e64 {S_M[1]}, {A[j]}, {B[j]}
	v_cmp_lt_f32_e64 {S_M[2]}, {A[j]}, {B[j]}
	v_cndmask_b32_e64 {R[j]}, {V_QNAN}, 1.0, {S_M[0]}
	s_nop 0
	v_cndmask_b32_e64 {R[j]}, {R[j]}, 0, {S_M[1]}
	v_cndmask_b32_e64 {R[j]}, {R[j]}, -1.0, {S_M[2]}""")
            return
        self.mask_pass(lambda j, m: f"v_cmp_gt_f32_e64 {m}, {A[j]}, {B[j]}", lambda j, m: f"v_cndmask_b32_e64 {R[j]}, {V_QNAN}, 1.0, {m}")
        self.mask_pass(lambda j, m: f"v_cmp_eq_f32_e64 {m}, {A[j]}, {B[j]}", lambda j, m: f"v_cndmask_b32_e64 {R[j]}, {R[j]}, 0, {m}")
        self.mask_pass(lambda j, m: f"v_cmp_lt_f32_e64 {m}, {A[j]}, {B[j]}", lambda j, m: f"v_cndmask_b32_e64 {R[j]}, {R[j]}, -1.0, {m}")

    def f_div(self, A, B, R):
        # IEEE-correct a / b: the div_scale / rcp / fma / div_fmas / div_fixup sequence
        d = VD
        for j in range(self.zb):
            a, b = A[j], B[j]
            self.a(f"""
	v_div_scale_f32 {d[0]}, {S_SAVE}, {b}, {b}, {a}
	v_rcp_f32 {d[1]}, {d[0]}
	v_div_scale_f32 {d[2]}, vcc, {a}, {b}, {a}
	v_fma_f32 {d[3]}, -{d[0]}, {d[1]}, 1.0
	v_fmac_f32 {d[1]}, {d[3]}, {d[1]}
	v_mul_f32 {d[3]}, {d[2]}, {d[1]}
	v_fma_f32 {d[4]}, -{d[0]}, {d[3]}, {d[2]}
	v_fmac_f32 {d[3]}, {d[4]}, {d[1]}
	v_fma_f32 {d[0]}, -{d[0]}, {d[3]}, {d[2]}
	v_div_fmas_f32 {d[0]}, {d[0]}, {d[1]}, {d[3]}
	v_div_fixup_f32 {R[j]}, {d[0]}, {b}, {a}""")

    def f_div_imm(self, A, B, R):
        """R = A / immediate (B: the immediate in every sample, for the general sequence).  The division sequence of f_div with the work
        that only the divisor needs - the reciprocal and its refinement - done once for the op, and the rest two samples per instruction:
        when neither operand needs v_div_scale_f32's scaling (both within 2^-40 .. 2^40: the exponents differ by less than 96, no
        denormal anywhere) the scaled operands are the operands and v_div_fmas_f32 is a plain fused multiply-add, so these are the same
        operations with the same roundings; v_div_fixup_f32 ends both.  3.5 + 1 instructions per sample instead of 11; any sample
        outside the range (zeros among them) in any lane: the general sequence for the op."""
        a, d = self.a, VD
        slow, done = a.label("div_general"), a.label("div_done")
        n = self.zb
        ab = [f"|{x}|" for x in A[:n]]
        lo, hi, r, e = d[6], d[7], d[0], d[1]
        a(f"""
	s_bfe_u32 {S_T0}, {S_W1}, 0x80017
	s_sub_u32 {S_T0}, {S_T0}, 87
	s_cmp_gt_u32 {S_T0}, 80
	s_cbranch_scc1 {slow}""")
        if n == 2:
            a(f"\tv_min_f32_e64 {lo}, {ab[0]}, {ab[1]}\n\tv_max_f32_e64 {hi}, {ab[0]}, {ab[1]}")
        else:
            a(f"\tv_min3_f32 {lo}, {ab[0]}, {ab[1]}, {ab[2]}\n\tv_max3_f32 {hi}, {ab[0]}, {ab[1]}, {ab[2]}")
            k = 3
            while n - k >= 2:
                a(f"\tv_min3_f32 {lo}, {lo}, {ab[k]}, {ab[k + 1]}\n\tv_max3_f32 {hi}, {hi}, {ab[k]}, {ab[k + 1]}")
                k += 2
            if k < n:
                a(f"\tv_min_f32_e64 {lo}, {ab[k]}, {lo}\n\tv_max_f32_e64 {hi}, {ab[k]}, {hi}")
        a(f"""
	v_rcp_f32 {r}, {S_W1}
	v_cmp_gt_f32 vcc, 0x2b800000, {lo}
	s_cbranch_vccnz {slow}
	v_cmp_le_f32 vcc, 0x53800000, {hi}
	s_cbranch_vccnz {slow}
	v_fma_f32 {e}, -{S_W1}, {r}, 1.0
	v_fmac_f32 {r}, {e}, {r}""")
        # per pair: q0 = a r; e1 = a - c q0; q1 = q0 + e1 r; e2 = a - c q1; q = q1 + e2 r   (c: the high half of the op's SGPR pair)
        tq, rr = [self.P(d, 1), self.P(d, 2)], self.P(d, 0)      # (r: the low half of its pair, selected for both samples)
        for k in range(n // 2):
            Ak, q = self.P(A, k), self.P(VU, k)
            ee = tq[k % 2]
            a(f"\tv_pk_mul_f32 {q}, {Ak}, {rr} op_sel_hi:[1,0]")
            a(f"\tv_pk_fma_f32 {ee}, {S_CUR}, {q}, {Ak} op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]")
            a(f"\tv_pk_fma_f32 {q}, {ee}, {rr}, {q} op_sel_hi:[1,0,1]")
            a(f"\tv_pk_fma_f32 {ee}, {S_CUR}, {q}, {Ak} op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]")
            a(f"\tv_pk_fma_f32 {q}, {ee}, {rr}, {q} op_sel_hi:[1,0,1]")
        for j in range(n):
            a(f"\tv_div_fixup_f32 {R[j]}, {VU[j]}, {S_W1}, {A[j]}")
        a(f"\ts_branch {done}\n{slow}:")
        self.imm_b(B)
        self.f_div(A, B, R)
        a(f"{done}:")

    def f_sqrt(self, A, R):
        # correctly rounded sqrtf: v_sqrt_f32 (1 ulp), then one ulp either way by the sign of the exact residuals.  Arguments
        # below 2^-96 (denormal results of the residuals), zeros and negative ones take the full sequence (scaled by 2^32, the
        # result by 2^-16, zeros / +inf passed through); when no sample of the op is one of those - the smallest sample is
        # tested - 9 instructions per sample do, with the same result: nothing is scaled, and +inf / NaN come out of the
        # residuals' NaN compares unchanged.
        d = VD
        slow, done = self.a.label("sqrt_small"), self.a.label("sqrt_done")
        t = d[6]
        if self.zb == 8:
            self.a(f"\tv_min3_f32 {t}, {A[0]}, {A[1]}, {A[2]}\n\tv_min3_f32 {t}, {t}, {A[3]}, {A[4]}\n\tv_min3_f32 {t}, {t}, {A[5]}, {A[6]}\n\tv_min_f32 {t}, {t}, {A[7]}")
        elif self.zb == 4:
            self.a(f"\tv_min3_f32 {t}, {A[0]}, {A[1]}, {A[2]}\n\tv_min_f32 {t}, {t}, {A[3]}")
        else:
            self.a(f"\tv_min_f32 {t}, {A[0]}, {A[1]}")
        self.a(f"\tv_cmp_gt_f32 vcc, {V_SQRTC}, {t}\n\ts_cbranch_vccnz {slow}")
        for j in range(self.zb):
            a = A[j]
            self.a(f"""
	v_sqrt_f32 {d[0]}, {a}
	s_nop 0
	v_add_u32 {d[2]}, -1, {d[0]}
	v_add_u32 {d[3]}, 1, {d[0]}
	v_fma_f32 {d[4]}, -{d[2]}, {d[0]}, {a}
	v_fma_f32 {d[5]}, -{d[3]}, {d[0]}, {a}
	v_cmp_ge_f32_e64 {S_M[0]}, 0, {d[4]}
	v_cmp_lt_f32_e64 {S_M[1]}, 0, {d[5]}
	s_nop 0
	v_cndmask_b32_e64 {d[0]}, {d[0]}, {d[2]}, {S_M[0]}
	v_cndmask_b32_e64 {R[j]}, {d[0]}, {d[3]}, {S_M[1]}""")
        self.a(f"\ts_branch {done}\n{slow}:")
        for j in range(self.zb):
            a = A[j]
            self.a(f"""
	v_mul_f32 {d[0]}, 0x4f800000, {a}
	v_cmp_gt_f32 vcc, {V_SQRTC}, {a}
	s_nop 1
	v_cndmask_b32 {d[1]}, {a}, {d[0]}, vcc
	v_sqrt_f32 {d[0]}, {d[1]}
	s_nop 0
	v_add_u32 {d[2]}, -1, {d[0]}
	v_add_u32 {d[3]}, 1, {d[0]}
	v_fma_f32 {d[4]}, -{d[2]}, {d[0]}, {d[1]}
	v_fma_f32 {d[5]}, -{d[3]}, {d[0]}, {d[1]}
	v_cmp_ge_f32_e64 {S_M[0]}, 0, {d[4]}
	s_nop 1
	v_cndmask_b32_e64 {d[0]}, {d[0]}, {d[2]}, {S_M[0]}
	v_cmp_lt_f32_e64 {S_M[0]}, 0, {d[5]}
	s_nop 1
	v_cndmask_b32_e64 {d[0]}, {d[0]}, {d[3]}, {S_M[0]}
	v_mul_f32 {d[2]}, 0x37800000, {d[0]}
	v_cndmask_b32 {d[0]}, {d[0]}, {d[2]}, vcc
	v_mov_b32 {d[3]}, 0x260
	v_cmp_class_f32 vcc, {d[1]}, {d[3]}
	s_nop 1
	v_cndmask_b32 {R[j]}, {d[0]}, {d[1]}, vcc""")
        self.a(f"{done}:")

    def f_round(self, A, R):
        # roundf: half away from zero
        d = VD
        for j in range(self.zb):
            a = A[j]
            self.a(f"""
	v_trunc_f32 {d[0]}, {a}
	v_sub_f32 {d[1]}, {a}, {d[0]}
	v_cmp_ge_f32_e64 {S_M[0]}, |{d[1]}|, 0.5
	s_nop 1
	v_cndmask_b32_e64 {d[1]}, 0, 1.0, {S_M[0]}
	v_bfi_b32 {d[1]}, {S_ABSM}, {d[1]}, {a}
	v_add_f32 {R[j]}, {d[0]}, {d[1]}""")

    # -- handlers ------------------------------------------------------------------------------
    def out_of_line(self, stem, fn):
        lab = f".L{self.name}_{stem}"
        self.ool.append((lab, fn))
        self.a(f"\ts_branch {lab}")

    def ool_label(self, stem, fn):
        """register an out-of-line body, return its label (the caller branches)"""
        lab = f".L{self.name}_{stem}"
        self.ool.append((lab, fn))
        return lab

    INPLACE = {"NEG", "ABS", "FLOOR", "CEIL", "SQUARE", "ADD_RI", "SUB_RI", "MUL_RI", "SUB_IR", "ADD_RR", "SUB_RR", "MUL_RR",
               "MIN_RR", "MAX_RR"}

    def handler(self, op, inplace=False):
        """One handler slot.  `inplace`: the decode found out == a (the in-place table, ops of INPLACE only): the
        op then works on the file entry itself.  Threaded dispatch: every handler starts with prologue() (this op's words,
        the next op's prefetch) and none leaves with SRC0-relative index mode on: the in-place forms keep the file operand
        in src1 where the instruction allows."""
        a, zb, F = self.a, self.zb, self.F
        Z, PZ = range(zb), range(zb // 2)
        T = self.threaded
        s_ip = self.s_out if T else S_A          # in place: out == a, either index will do (threaded: the one that needs no shift)
        if op == "OUTPUT":
            return self.h_output()
        if op == "INPUT":
            self.prologue()
            return self.out_of_line("input", self.h_input)
        if op in ("INPUT_X", "INPUT_Y", "INPUT_Z"):        # (threaded decode only: an INPUT op whose slot is the axis')
            self.prologue()
            return self.h_input_axis("XYZ".index(op[-1]))
        if op == "NEXT_CHUNK":
            return a(f"\ts_setpc_b64 {S_LONG}")
        if op == "COPY_REG":
            self.prologue(need_a=True)
            self.read_a(VT)
            return self.write_out(VT)
        if op == "COPY_IMM":
            self.prologue()
            self.idx_on(self.s_out, DST)
            for j in Z:
                a(f"\tv_mov_b32 {F(j)}, {S_W1}")
            return self.ret()
        if op in ("NEG", "ABS", "FLOOR", "CEIL"):
            self.prologue(need_a=not inplace)
            ins = {"NEG": f"v_xor_b32 {{d}}, {S_SIGN}, {{s}}", "ABS": f"v_and_b32 {{d}}, {S_ABSM}, {{s}}",
                   "FLOOR": "v_floor_f32 {d}, {s}", "CEIL": "v_ceil_f32 {d}, {s}"}[op]
            src = SRC1 if op in ("NEG", "ABS") else SRC0
            self.idx_on(s_ip if inplace else self.s_a, src | (DST if inplace else 0))
            for j in Z:
                a("\t" + ins.format(d=F(j) if inplace else VT[j], s=F(j)))
            return self.ret(src0_on=src == SRC0) if inplace else self.write_out(VT)
        if op == "SQUARE":
            self.prologue(need_a=not inplace)
            self.idx_on(s_ip if inplace else self.s_a, SRC0 | SRC1 | (DST if inplace else 0))
            for k in PZ:
                a(f"\tv_pk_mul_f32 {self.FP(k) if inplace else self.P(VT, k)}, {self.FP(k)}, {self.FP(k)}")
            return self.ret(src0_on=True) if inplace else self.write_out(VT)
        if op == "NOT":
            self.prologue(need_a=True)
            def body():
                self.read_a(VT)
                self.idx_off()
                self.mask_pass(lambda j, m: f"v_cmp_eq_f32_e64 {m}, 0, {VT[j]}", lambda j, m: f"v_cndmask_b32_e64 {VU[j]}, 0, 1.0, {m}")
                self.write_out(VU)
            return self.out_of_line("not", body)
        if op in ("RECIP", "SQRT", "ROUND"):
            self.prologue(need_a=True)
            def body(op=op):
                self.read_a(VT)
                self.idx_off()
                if op == "RECIP":
                    one = ["1.0"] * 8
                    self.f_div(one, VT, VU)
                elif op == "SQRT" and "nosqrt" in EXP.split(","):      # experiment: what the rounding of the root costs
                    for j in Z:
                        a(f"\tv_sqrt_f32 {VU[j]}, {VT[j]}")
                elif op == "SQRT":
                    self.f_sqrt(VT, VU)
                    if "twicesqrt" in EXP.split(","):     # experiment: the root's cost once more, same results
                        self.f_sqrt(VT, VU)
                else:
                    self.f_round(VT, VU)
                self.write_out(VU)
            return self.out_of_line(op.lower(), body)
        if op in ("SIN", "COS", "TAN", "ASIN", "ACOS", "ATAN", "EXP", "LN"):
            self.prologue(need_a=True)
            def body(fn=op.lower()):
                self.read_a(VT)
                self.idx_off()
                if "no" + fn in EXP.split(","):       # experiment: what the routine costs (a copy in its place)
                    return self.write_out(VT)
                if (fn in ("exp", "ln", "sin", "cos") and self.exp2) or (fn in ("sin", "cos") and self.sincos2):      # expf / logf / sinf / cosf by hand, two samples at a time (gen_trans.exp_pair ...)
                    import gen_trans
                    slow, join = a.label(fn + "_special"), a.label(fn + "_done")
                    consts, special, two = gen_trans.hand(fn)
                    consts(a, self.t_base)
                    special(a, self.t_base, VT[:self.zb], slow)
                    for j0 in range(0, self.zb, 2):
                        two(a, self.t_base, VT[j0:j0 + 2], VU[j0:j0 + 2])
                    a(f"{join}:")
                    self.write_out(VU)
                    a(f"{slow}:")                   # some lane's argument is one of glibc's special cases: the compiled routine, sample by sample
                    for j in Z:
                        a(f"\tv_mov_b32 v{self.t_base}, {VT[j]}")
                        self.call(fn)
                        a(f"\tv_mov_b32 {VU[j]}, v{self.t_base}")
                    return a(f"\ts_branch {join}")
                for rep in range(2 if "twice" + fn in EXP.split(",") else 1):     # experiment: the routine's cost once more, same results
                  if self.zb % 4 == 0 and ((self.wide_trans is True and fn in ("sin", "cos", "exp", "ln")) or (self.wide_trans == "sincos" and fn in ("sin", "cos"))):
                      for j0 in range(0, self.zb, 4):       # four samples per call (gen_trans.FUNCS4)
                          for k in range(4):
                              a(f"\tv_mov_b32 v{self.t_base + k}, {VT[j0 + k]}")
                          self.call(fn + "4")
                          for k in range(4):
                              a(f"\tv_mov_b32 {VU[j0 + k]}, v{self.t_base + k}")
                  else:
                      for j in Z:
                          a(f"\tv_mov_b32 v{self.t_base}, {VT[j]}")
                          self.call(fn)
                          a(f"\tv_mov_b32 {VU[j]}, v{self.t_base}")
                self.write_out(VU)
            return self.out_of_line(op.lower(), body)
        if op == "RAND":
            self.prologue(need_a=True)
            def body():
                self.read_a(VT)
                self.idx_off()
                self.pcg_consts()
                for j in Z:                       # rng::rand (rng/mod.rs:19-23): bits (hash >> 9) | 1.0, minus 1
                    self.pcg(VT[j], VU[j])
                    a(f"\tv_lshrrev_b32 {VU[j]}, 9, {VU[j]}\n\tv_or_b32 {VU[j]}, 0x3f800000, {VU[j]}\n\tv_add_f32 {VU[j]}, -1.0, {VU[j]}")
                self.write_out(VU)
            return self.out_of_line("rand", body)
        base, form = op.rsplit("_", 1)
        if base in ("ATAN2", "MOD", "MIX"):
            self.prologue(need_a=True)
            def body(base=base, form=form):
                self.read_a(VT)
                if form == "RR":
                    self.read_b(VU)
                self.idx_off()
                if form != "RR":
                    self.imm_b(VU)
                A, B = (VT, VU) if form != "IR" else (VU, VT)
                if base == "MIX":                 # rng::mix (rng/mod.rs:30-33): hash(a + hash(b)) on the bit patterns
                    self.pcg_consts()
                    for j in Z:                   # (through VD[2]: VW may be VT - the compact map)
                        self.pcg(B[j], VD[2])
                        a(f"\tv_add_u32 {VD[2]}, {A[j]}, {VD[2]}")
                        self.pcg(VD[2], VW[j])
                else:
                    for j in Z:
                        a(f"\tv_mov_b32 v{self.t_base}, {A[j]}\n\tv_mov_b32 v{self.t_base + 1}, {B[j]}")
                        self.call(base.lower())
                        a(f"\tv_mov_b32 {VW[j]}, v{self.t_base}")
                self.write_out(VW)
            return self.out_of_line(op.lower(), body)
        if base in ("ADD", "SUB", "MUL") and form != "RR":
            # register (op) immediate, two samples per instruction: the immediate is the high half of the op's
            # SGPR pair, selected for both samples, in src0; the file operand in src1 (no SRC0-relative mode: see prologue).
            # a - imm = (-imm) + a and imm - a = imm + (-a), exactly.
            self.prologue(need_a=not inplace)
            ins = "v_pk_mul_f32" if base == "MUL" else "v_pk_add_f32"
            mod = {("SUB", "RI"): " neg_lo:[1,0] neg_hi:[1,0]", ("SUB", "IR"): " neg_lo:[0,1] neg_hi:[0,1]"}.get((base, form), "")
            self.idx_on(s_ip if inplace else self.s_a, SRC1 | (DST if inplace else 0))
            for k in PZ:
                a(f"\t{ins} {self.FP(k) if inplace else self.P(VT, k)}, {S_CUR}, {self.FP(k)} op_sel:[1,0] op_sel_hi:[1,1]{mod}")
            return self.ret() if inplace else self.write_out(VT)
        if base in ("ADD", "SUB", "MUL"):
            ins = "v_pk_mul_f32" if base == "MUL" else "v_pk_add_f32"
            if inplace:                              # file[a] = file[a] (op) b, b through VU:  b (op) a for add / mul, (-b) + a for sub
                self.prologue()
                self.read_b(VU, already_on=False)
                self.idx_on(s_ip, SRC1 | DST)
                negb = " neg_lo:[1,0] neg_hi:[1,0]" if base == "SUB" else ""
                for k in PZ:
                    a(f"\t{ins} {self.FP(k)}, {self.P(VU, k)}, {self.FP(k)}{negb}")
                return self.ret()
            self.prologue(need_a=True)
            negb = " neg_lo:[0,1] neg_hi:[0,1]" if base == "SUB" else ""
            self.read_a(VT)                          # a in VT; then VT = VT (op) file[b]
            self.idx_on(self.b_index(), SRC1)
            for k in PZ:
                a(f"\t{ins} {self.P(VT, k)}, {self.P(VT, k)}, {self.FP(k)}{negb}")
            return self.write_out(VT)
        if base in ("MIN", "MAX") and form == "RR" and inplace:
            # out == a (nearly always: the accumulator of a union / intersection): the file entry is updated in place.
            # min: a < b ? a : b (max: a > b ? a : b) with b in src0 and a, the relative operand, in src1: b > a (b < a) selects a.
            cmp = "v_cmp_gt_f32_e64" if base == "MIN" else "v_cmp_lt_f32_e64"
            self.prologue()
            def body(cmp=cmp):
                # One v_minimum3_f32 / v_maximum3_f32 per sample (see f_minmax) unless a sample of a is a zero: the two tests and
                # two selects per sample (dev_ops.hpp f_min / f_max to the letter) are left to that case.
                slow = a.label("mm_zero")
                self.mm_slow = getattr(self, "mm_slow", {})
                self.mm_slow[base] = slow
                self.read_b(VU, already_on=False)
                self.idx_on(s_ip, SRC1 | SRC2)      # a's samples relative, plain destination: is one of them a zero?
                self.zero_guard([F(j) for j in Z], VD[7])
                self.idx_on(s_ip, SRC1 | SRC2 | DST)
                a(f"\ts_cbranch_vccnz {slow}")
                for j in Z:
                    a(f"\t{'v_minimum3_f32' if base == 'MIN' else 'v_maximum3_f32'} {F(j)}, {VU[j]}, {F(j)}, {F(j)}")
                self.ret()
                def slow_body():
                    for j in range(0, zb, 2):          # both tests of a sample before it is overwritten; 4 masks = 2 samples
                        for q in (0, 1):
                            a(f"\t{cmp} {S_M[2 * q]}, {VU[j + q]}, {F(j + q)}")
                            a(f"\tv_cmp_o_f32_e64 {S_M[2 * q + 1]}, {VU[j + q]}, {F(j + q)}")
                        for q in (0, 1):
                            a(f"\tv_cndmask_b32_e64 {F(j + q)}, {VU[j + q]}, {F(j + q)}, {S_M[2 * q]}")
                            a(f"\tv_cndmask_b32_e64 {F(j + q)}, {V_QNAN}, {F(j + q)}, {S_M[2 * q + 1]}")
                    self.ret()
                if T:
                    self.ool.append((slow, slow_body))
                else:
                    a(f"{slow}:")
                    slow_body()
            if T:           # (threaded: slots of 256 bytes, no jump to the body)
                return body()
            return self.out_of_line(op.lower() + "_i", body)
        # two plain operands A, B in VT / VU, result in VW
        self.prologue(need_a=True)
        def body(base=base, form=form):
            self.read_a(VT)
            if form == "RR":
                self.read_b(VU)
            self.idx_off()
            fast_div = base == "DIV" and form == "RI" and self.kind in ("columns", "bulk") and self.zb >= 2 and not (set(EXP.split(",")) & {"nofastdiv", "nodiv"})
            if form != "RR" and not fast_div:      # (the division by an immediate fills VU itself, where it takes the general sequence)
                self.imm_b(VU)
            A, B = (VT, VU) if form != "IR" else (VU, VT)
            if base == "DIV" and "nodiv" in EXP.split(","):      # experiment: what the division costs (a product in its place)
                for k in PZ:
                    a(f"\tv_pk_mul_f32 {self.P(VW, k)}, {self.P(A, k)}, {self.P(B, k)}")
            elif fast_div:
                self.f_div_imm(A, B, VW)
            elif base == "DIV":
                self.f_div(A, B, VW)
                if "twicediv" in EXP.split(","):     # experiment: the division's cost once more, same results
                    self.f_div(A, B, VW)
            elif base == "COMPARE":
                self.f_compare(A, B, VW)
            elif base in ("MIN", "MAX"):
                self.f_minmax(base == "MIN", A, B, VW)
            else:
                self.f_andor(base == "AND", A, B, VW)
            self.write_out(VW)
        return self.out_of_line(op.lower(), body)

    def h_output(self):
        a = self.a
        if self.kind == "columns":
            # the one output of a shape tape is its last op: back to the caller
            if self.threaded:
                a(f"\ts_lshr_b32 {S_A}, {S_NX_LO}, 8")
            self.read_a(VRES)
            self.idx_off()
            return a(f"\ts_waitcnt lgkmcnt(0)\n\ts_setpc_b64 {S_RET}")
        self.out_of_line("output", self.h_output_bulk)

    def h_output_bulk(self):
        a = self.a
        self.read_a(VT)
        self.idx_off()
        # out + (slot * n) * 4, 64-bit
        a(f"""
	s_mul_hi_u32 s77, {S_W1}, {S_N}
	s_mul_i32 s76, {S_W1}, {S_N}
	s_lshl_b64 {S_PC}, s[76:77], 2
	s_add_u32 s86, s86, s32
	s_addc_u32 s87, s87, s33
	s_mov_b64 {S_SAVE}, exec""")
        for j in range(self.zb):
            a(f"""
	s_mov_b64 exec, {S_ACT[j]}
	global_store_dword {VOFF[j]}, {VT[j]}, {S_PC}""")
        a(f"""
	s_mov_b64 exec, {S_SAVE}""")
        self.ret()

    def h_input(self):
        a = self.a
        self.idx_off()      # (plain vector code below; the previous handler may have left the index mode on)
        if self.kind == "bulk":
            a(f"""
	s_mul_hi_u32 s77, {S_W1}, {S_N}
	s_mul_i32 s76, {S_W1}, {S_N}
	s_lshl_b64 {S_PC}, s[76:77], 2
	s_add_u32 s86, s86, s30
	s_addc_u32 s87, s87, s31""")
            for j in range(self.zb):
                a(f"\tglobal_load_dword {VT[j]}, {VOFF[j]}, {S_PC}")
            a("\ts_waitcnt vmcnt(0)")
            return self.write_out(VT)
        # columns: the slot is x, y or z (model coordinates of the ZB voxels of this pass, computed
        # here: ((m[4r] x + m[4r+1] y) + m[4r+2] z) + m[4r+3], dev_ops.hpp xf_point) or a bound constant
        lab = {k: a.label("in_" + k) for k in ("x", "y", "z", "done")}
        a(f"""
	s_cmp_eq_u32 {S_W1}, {S_SLOTX}
	s_cbranch_scc1 {lab['x']}
	s_cmp_eq_u32 {S_W1}, {S_SLOTY}
	s_cbranch_scc1 {lab['y']}
	s_cmp_eq_u32 {S_W1}, {S_SLOTZ}
	s_cbranch_scc1 {lab['z']}
	s_lshl_b32 {S_T0}, {S_W1}, 2
	s_add_u32 s86, s4, {S_T0}
	s_addc_u32 s87, s5, 0
	s_load_dword {S_T1}, {S_PC}, {self.off['P.in_value']}
	s_waitcnt lgkmcnt(0)""")
        self.idx_on(self.s_out, DST)
        for j in range(self.zb):
            a(f"\tv_mov_b32 {self.F(j)}, {S_T1}")
        a(f"\ts_branch {lab['done']}")
        for axis in range(3):
            a(f"{lab['xyz'[axis]]}:")
            vary = a.label("in_vary")
            self.axis_const(axis, vary)
            a(f"\ts_branch {lab['done']}\n{vary}:")
            self.axis_vary(axis)
            if axis != 2:
                a(f"\ts_branch {lab['done']}")
        a(f"{lab['done']}:")
        self.idx_off()
        self.ret()

    def h_input_axis(self, axis):
        """threaded decode: an INPUT op of the x / y / z slot has its own handler (no slot compares, no jump for the common case)"""
        vary = self.ool_label(f"in_vary_{'xyz'[axis]}", lambda: (self.axis_vary(axis), self.ret()))
        self.idx_off()          # (plain vector code first: the handler before this one may have left a mode on)
        self.axis_const(axis, vary)
        self.ret()

    def axis_const(self, axis, vary):
        # The axis' matrix row has no z coefficient and the matrix is not projective (kernarg flags bit 17 + row clear: from
        # the camera alone, whatever the tape-level short cuts are set to): m[4r+2] is a zero, m[4r+2] * z is that same zero
        # for every voxel (z >= 0), so the 8 samples are ONE value, ((m[4r] x + m[4r+1] y) + m[4r+2]) + m[4r+3] - the sum in
        # the order of the general path, bit for bit - instead of a convert, a multiply and two adds per sample.
        a, m, row, acc = self.a, S_MAT, axis, (V_AX, V_AY, V_AZ)[axis]
        a(f"\ts_bitcmp1_b32 {S_PROJFLAG}, {17 + row}\n\ts_cbranch_scc1 {vary}")
        a(f"\tv_add_f32 {VT[0]}, s{m + 4 * row + 2}, {acc}\n\tv_add_f32 {VT[0]}, s{m + 4 * row + 3}, {VT[0]}\n\tv_mov_b32 {VT[1]}, {VT[0]}")
        self.idx_on(self.s_out, DST)
        for k in range(self.zb // 2):
            a(f"\tv_pk_mov_b32 {self.FP(k)}, {self.P(VT, 0)}, {self.P(VT, 0)} op_sel:[0,1]")

    def axis_vary(self, axis):
        """file[out] = the axis' model coordinate of the ZB voxels of this pass (index mode: off on entry, DST-relative on exit)"""
        a, m, row, acc = self.a, S_MAT, axis, (V_AX, V_AY, V_AZ)[axis]
        self.idx_off()
        a(f"\ts_add_u32 {S_T0}, {S_LZ}, {S_K}")
        for j in range(self.zb):      # z of sample j = lz + k - j
            a(f"\tv_cvt_f32_u32 {VT[j]}, {S_T0}")
            if j + 1 < self.zb:
                a(f"\ts_sub_u32 {S_T0}, {S_T0}, 1")
        a(f"\ts_mov_b32 s96, s{m + 4 * row + 2}\n\ts_mov_b32 s97, s{m + 4 * row + 3}")
        a(f"\tv_mov_b32 {VU[0]}, {acc}")       # (an aligned pair for the packed add; its low half serves both samples)
        for k in range(self.zb // 2):
            a(f"\tv_pk_mul_f32 {self.P(VT, k)}, {self.P(VT, k)}, s[96:97] op_sel_hi:[1,0]")
        for k in range(self.zb // 2):
            a(f"\tv_pk_add_f32 {self.P(VT, k)}, {self.P(VT, k)}, {self.P(VU, 0)} op_sel_hi:[1,0]")
        proj, done = a.label("in_proj"), a.label("in_vdone")
        a(f"\ts_bitcmp1_b32 {S_PROJFLAG}, 16\n\ts_cbranch_scc1 {proj}")
        self.idx_on(self.s_out, DST)
        for k in range(self.zb // 2):
            a(f"\tv_pk_add_f32 {self.FP(k)}, {self.P(VT, k)}, s[96:97] op_sel:[0,1] op_sel_hi:[1,1]")
        a(f"\ts_branch {done}")
        # projective matrix (shape/mod.rs:906-916, nalgebra transform_point): the row value divided by
        # w = ((m[12] x + m[13] y) + m[14] z) + m[15] wherever w != 0
        a(f"{proj}:")
        for k in range(self.zb // 2):
            a(f"\tv_pk_add_f32 {self.P(VT, k)}, {self.P(VT, k)}, s[96:97] op_sel:[0,1] op_sel_hi:[1,1]")
        a(f"\ts_add_u32 {S_T0}, {S_LZ}, {S_K}")
        for j in range(self.zb):
            a(f"\tv_cvt_f32_u32 {VU[j]}, {S_T0}")
            if j + 1 < self.zb:
                a(f"\ts_sub_u32 {S_T0}, {S_T0}, 1")
        a(f"\ts_mov_b32 s96, s{m + 14}\n\ts_mov_b32 s97, s{m + 15}")
        awp = VD[6:8] if VW is VT else VW          # (an aligned pair whose low half holds m[12] x + m[13] y)
        a(f"\tv_mov_b32 {awp[0]}, {V_AW}")
        for k in range(self.zb // 2):
            a(f"\tv_pk_mul_f32 {self.P(VU, k)}, {self.P(VU, k)}, s[96:97] op_sel_hi:[1,0]")
        for k in range(self.zb // 2):
            a(f"\tv_pk_add_f32 {self.P(VU, k)}, {self.P(VU, k)}, {self.P(awp, 0)} op_sel_hi:[1,0]")
        for k in range(self.zb // 2):
            a(f"\tv_pk_add_f32 {self.P(VU, k)}, {self.P(VU, k)}, s[96:97] op_sel:[0,1] op_sel_hi:[1,1]")
        if VW is VT:        # the compact map: the quotient over the dividend - w == 0 keeps the row value: divided by 1.0 instead, which is exact
            self.mask_pass(lambda j, mk: f"v_cmp_eq_f32_e64 {mk}, 0, {VU[j]}", lambda j, mk: f"v_cndmask_b32_e64 {VU[j]}, {VU[j]}, 1.0, {mk}")
            self.f_div(VT, VU, VT)
        else:
            self.f_div(VT, VU, VW)
            self.mask_pass(lambda j, mk: f"v_cmp_neq_f32_e64 {mk}, 0, {VU[j]}", lambda j, mk: f"v_cndmask_b32_e64 {VT[j]}, {VT[j]}, {VW[j]}, {mk}")
        self.write_out(VT, done=False)
        a(f"{done}:")

    # -- the interpreter -------------------------------------------------------------------
    def emit(self):
        if self.threaded:
            return self.emit_threaded()
        a, n, lg = self.a, self.name, self.lg
        qa, qb = S_QA, S_QB
        COPY = 128
        a(f"""
; ---- interpreter {n}: in {S_TAPE} = first op, {S_LEN} = ops (bulk); returns to {S_RET} -----------
; The tape comes through the scalar cache 4 ops at a time into two SGPR batches; the decode code
; exists 8 times, copy i for the op in slot i of the batches, so that nothing is shifted or counted:
; a handler ends with a jump to {S_NEXT}, which every copy advances to the copy after it.
.L{n}_run:
	s_load_dwordx8 s[{qa}:{qa + 7}], {S_TAPE}, 0x0
.L{n}_go:                                ; entry for callers that have requested the first 4 ops themselves
	s_add_u32 s72, s44, 0x20
	s_addc_u32 s73, s45, 0
	s_getpc_b64 {S_NEXT}
.L{n}_gopc:
	s_add_u32 s70, s70, .L{n}_d0 - .L{n}_gopc
	s_addc_u32 s71, s71, 0
	s_setpc_b64 {S_NEXT}
	.p2align 7""")
        for i in range(8):
            q = (qa if i < 4 else qb) + 2 * (i % 4)
            a(f".L{n}_d{i}:")
            if i % 4 == 0:
                other = qb if i == 0 else qa
                a(f"""
	s_waitcnt lgkmcnt(0)
	s_load_dwordx8 s[{other}:{other + 7}], {S_FETCH}, 0x0
	s_add_u32 s72, s72, 0x20
	s_addc_u32 s73, s73, 0""")
            if self.kind in ("bulk", "grad"):      # (tapes of any length and number of outputs: the ops are counted)
                a(f"""
	s_sub_u32 {S_LEN}, {S_LEN}, 1
	s_cbranch_scc1 .L{n}_done""")
            a(f"""
	s_mov_b64 {S_CUR}, s[{q}:{q + 1}]
	{f"s_add_u32 s70, s70, {COPY}" if i < 7 else f"s_sub_u32 s70, s70, {7 * COPY}"}
	s_lshl_b32 {S_T0}, {S_W0}, {self.hl}
	s_lshr_b32 {S_OUT}, {S_W0}, {8 - lg}
	s_lshr_b32 {S_A}, {S_W0}, {20 - lg}
	s_and_b32 {S_T0}, {S_T0}, {hex(0xff << self.hl)}
	s_and_b32 {S_OUT}, {S_OUT}, {hex(0xfff << lg)}     ; file index = register * ZB
	s_andn2_b32 {S_A}, {S_A}, {(1 << lg) - 1}
	s_cmp_eq_u32 {S_OUT}, {S_A}
	s_cselect_b32 {S_T1}, {hex(64 << self.hl)}, 0   ; out == a: the in-place handlers
	s_or_b32 {S_T0}, {S_T0}, {S_T1}
	s_add_u32 s44, s42, {S_T0}
	s_addc_u32 s45, s43, 0
	s_setpc_b64 {S_JMP}
	.if (. - .L{n}_d{i}) > {COPY}
	.error "decode copy of {n} exceeds its slot"
	.endif
	.p2align 7""")
        if self.kind == "columns":
            a(f"""
; ---- ... and for tapes the caller has decoded into {V_DEC[0]}..{V_DEC[3]} (lane = op): one copy, four v_readlane
.L{n}_gov:
	s_getpc_b64 {S_NEXT}
.L{n}_govpc:
	s_add_u32 s70, s70, .L{n}_dv - .L{n}_govpc
	s_addc_u32 s71, s71, 0
	s_mov_b32 s45, s43
	s_mov_b32 {S_LEN}, 0
.L{n}_dv:
	s_set_gpr_idx_off                               ; (the handlers leave their index mode on)
	v_readlane_b32 s44, {V_DEC[0]}, {S_LEN}
	v_readlane_b32 {S_OUT}, {V_DEC[1]}, {S_LEN}
	v_readlane_b32 {S_W1}, {V_DEC[3]}, {S_LEN}
	s_add_u32 {S_LEN}, {S_LEN}, 1
	s_lshr_b32 {S_A}, {S_OUT}, 8
	s_setpc_b64 {S_JMP}""")
        a(f"""
.L{n}_done:
	s_waitcnt lgkmcnt(0)
	s_set_gpr_idx_off
	s_setpc_b64 {S_RET}
	.p2align {self.hl}
.L{n}_handlers:""")
        for inplace in (False, True):
            for i in range(64):
                a(f"\t.p2align {self.hl}")
                lab = f".L{n}_{'i' if inplace else 'h'}{i}"
                op = OPS[i] if i < len(OPS) else None
                a(f"{lab}:  ; {op}{' (in place)' if inplace else ''}")
                base = op.rsplit("_", 1)[0] if op and "_" in op and op not in ("COPY_REG", "COPY_IMM") else op
                if op is None or (base in UNSUPPORTED and not self.trans):
                    self.ret()             # never reached for tapes routed here (host checks)
                elif inplace and op not in self.INPLACE:
                    a(f"\ts_branch .L{n}_h{i}")
                elif EXP == "nowork" and self.kind == "columns":     # experiment: the dispatch alone (every op a no-op, the result "outside")
                    if op == "OUTPUT":
                        for j in range(self.zb):
                            a(f"\tv_mov_b32 {VRES[j]}, 1.0")
                        a(f"\ts_setpc_b64 {S_RET}")
                    else:
                        self.ret()
                else:
                    self.handler(op, inplace)
                # the next .p2align would silently grow the slot: an explicit assertion on its size
                a(f"\t.if (. - {lab}) > {1 << self.hl}\n\t.error \"handler {op} of {n} exceeds its slot\"\n\t.endif")
        a(f"\t.p2align {self.hl}")
        for lab, fn in self.ool:
            a(f"{lab}:")
            fn()



    # Delta handlers (threaded dispatch).  An op that reads one file register and writes another needs two index settings and a copy
    # through temporaries - unless the DISTANCE between the two registers is part of the instruction: with M0 = the written register,
    # the read operand is encoded as file + distance * ZB.  One handler per (op, distance) for the forms that matter (prospero's leaf
    # tapes: 24 % of the ops are register-immediate / unary forms with out != a, 28 % are in-place RR forms whose b is another register):
    # family U = unary / register-immediate ops with out != a (distance a - out), family B = in-place RR ops (distance b - a).
    U_OPS = ["COPY_REG", "NEG", "ABS", "SQUARE", "ADD_RI", "SUB_RI", "MUL_RI", "SUB_IR"]
    B_OPS = ["ADD_RR", "SUB_RR", "MUL_RR", "MIN_RR", "MAX_RR"]

    def delta_setup(self):
        # (not in the kernel for tapes with transcendental opcodes: their leaves are hundreds of ops of mostly other kinds, and the longer
        # decode cost bear.vm 1.4 % where prospero.vm's leaf kernel gained 3 %)
        # (... and the read operand's encoded number, file + distance * ZB, must not fall below v0: the compact map's file starts at v48)
        self.dmax = min(self.nr - 1, 7, FILE // self.zb) if (self.threaded and self.zb >= 4 and not self.trans and EXP != "nodelta") else 0
        self.nd = 2 * self.dmax + 1
        self.su = {8: 80, 4: 64}.get(self.zb, 0)       # slot bytes of family U / B
        self.sb = {8: 160, 4: 112}.get(self.zb, 0)

    def handler_delta(self, fam, op, d):
        a, zb = self.a, self.zb
        do = d * zb
        Fd = lambda j: f"v{FILE + do + j}"
        FdP = lambda k: f"v[{FILE + do + 2 * k}:{FILE + do + 2 * k + 1}]"
        Z, PZ = range(zb), range(zb // 2)
        self.prologue()
        if fam == "U":
            if op in ("COPY_REG", "SQUARE"):
                self.idx_on(self.s_out, SRC0 | SRC1 | DST)
                for k in PZ:
                    a(f"\tv_pk_mov_b32 {self.FP(k)}, {FdP(k)}, {FdP(k)} op_sel:[0,1]" if op == "COPY_REG" else f"\tv_pk_mul_f32 {self.FP(k)}, {FdP(k)}, {FdP(k)}")
                return self.ret(src0_on=True)
            self.idx_on(self.s_out, SRC1 | DST)
            if op in ("NEG", "ABS"):
                for j in Z:
                    a(f"\tv_xor_b32 {self.F(j)}, {S_SIGN}, {Fd(j)}" if op == "NEG" else f"\tv_and_b32 {self.F(j)}, {S_ABSM}, {Fd(j)}")
                return self.ret()
            base, form = op.rsplit("_", 1)
            ins = "v_pk_mul_f32" if base == "MUL" else "v_pk_add_f32"
            mod = {("SUB", "RI"): " neg_lo:[1,0] neg_hi:[1,0]", ("SUB", "IR"): " neg_lo:[0,1] neg_hi:[0,1]"}.get((base, form), "")
            for k in PZ:
                a(f"\t{ins} {self.FP(k)}, {S_CUR}, {FdP(k)} op_sel:[1,0] op_sel_hi:[1,1]{mod}")
            return self.ret()
        base = op.rsplit("_", 1)[0]
        if base in ("ADD", "SUB"):       # a + b = 1.0 * b + a, a - b = 1.0 * (-b) + a, exactly; b in src1, a in src2: no SRC0-relative mode
            self.idx_on(self.s_out, SRC1 | SRC2 | DST)
            neg = " neg_lo:[0,1,0] neg_hi:[0,1,0]" if base == "SUB" else ""
            for k in PZ:
                a(f"\tv_pk_fma_f32 {self.FP(k)}, {S_ONES}, {FdP(k)}, {self.FP(k)}{neg}")
            return self.ret()
        if base == "MUL":
            self.idx_on(self.s_out, SRC0 | SRC1 | DST)
            for k in PZ:
                a(f"\tv_pk_mul_f32 {self.FP(k)}, {FdP(k)}, {self.FP(k)}")
            return self.ret(src0_on=True)
        # min / max in place, b at a distance: see the in-place handler; the neutral first operand (+inf / -inf) keeps src0 plain
        slow = a.label("mmd_zero")
        self.idx_on(self.s_out, SRC1 | SRC2)
        self.zero_guard([self.F(j) for j in Z], VD[7])
        self.idx_on(self.s_out, SRC1 | SRC2 | DST)
        a(f"\ts_cbranch_vccnz {slow}")
        for j in Z:
            a(f"\t{'v_minimum3_f32' if base == 'MIN' else 'v_maximum3_f32'} {self.F(j)}, {V_PINF if base == 'MIN' else V_NINF}, {Fd(j)}, {self.F(j)}")
        self.ret()

        def slow_stub(base=base):          # a zero among a's samples: b into VU, then the in-place handler's compares and selects
            self.idx_on(self.s_out, SRC0 | SRC1)
            for k in PZ:
                self.pk_mov(self.P(VU, k), FdP(k))
            self.idx_on(self.s_out, SRC1 | SRC2 | DST)
            a(f"\ts_branch {self.mm_slow[base]}")
        self.ool.append((slow, slow_stub))

    # pseudo-opcodes of the threaded decode (slots the tape format leaves free)
    PSEUDO = {52: "INPUT_X", 53: "INPUT_Y", 54: "INPUT_Z", 55: "NEXT_CHUNK"}

    def emit_threaded(self):
        """entry + handler tables of the threaded dispatch (see __init__): `_gov` starts on the tape the caller has decoded into
        V_DEC (lane = op): handler address, out | a << 8 file indices, word 1 (b's file index for the RR forms)."""
        a, n = self.a, self.name
        a(f"""
; ---- interpreter {n} (threaded): tape decoded in {V_DEC[0]}, {V_DEC[1]}, {V_DEC[3]}, lane = op (the caller reads the first op's words and jumps
; into its handler); returns to {S_RET} from the OUTPUT op
	.p2align {self.ip_slot_log2}
.L{n}_handlers:""")
        for inplace in (False, True):
            lg = self.ip_slot_log2 if inplace else self.hl
            for i in range(64):
                a(f"\t.p2align {lg}")
                lab = f".L{n}_{'i' if inplace else 'h'}{i}"
                op = OPS[i] if i < len(OPS) else self.PSEUDO.get(i)
                a(f"{lab}:  ; {op}{' (in place)' if inplace else ''}")
                base = op.rsplit("_", 1)[0] if op and "_" in op and op not in ("COPY_REG", "COPY_IMM", "NEXT_CHUNK") and not op.startswith("INPUT_") else op
                if op is None or (base in UNSUPPORTED and not self.trans):
                    self.prologue()
                    self.ret()             # never reached for tapes routed here (host checks)
                elif inplace and op not in self.INPLACE:
                    a(f"\ts_branch .L{n}_h{i}")
                elif EXP == "nowork" and op != "NEXT_CHUNK":     # experiment: the dispatch alone (every op a no-op, the result "outside")
                    if op == "OUTPUT":
                        for j in range(self.zb):
                            a(f"\tv_mov_b32 {VRES[j]}, 1.0")
                        a(f"\ts_setpc_b64 {S_RET}")
                    else:
                        self.prologue()
                        self.ret()
                else:
                    self.handler(op, inplace)
                a(f"\t.if (. - {lab}) > {1 << lg}\n\t.error \"handler {op} of {n} exceeds its slot\"\n\t.endif")
        if self.dmax:
            for fam, ops, slot in (("U", self.U_OPS, self.su), ("B", self.B_OPS, self.sb)):
                a(f"\t.p2align 6\n.L{n}_d{fam.lower()}:")
                # (slot order: the decode counts the family's opcodes AT OR ABOVE the op's - the highest opcode comes first)
                for op in sorted(ops, key=OPS.index, reverse=True):
                    for d in range(-self.dmax, self.dmax + 1):
                        lab = a.label(f"d{fam}_{op}_{d + self.dmax}")
                        a(f"{lab}:")
                        if d == 0:
                            a("\ts_endpgm")          # (never selected: distance 0 is the in-place table's)
                        else:
                            self.handler_delta(fam, op, d)
                        a(f"\t.if (. - {lab}) > {slot}\n\t.error \"delta handler {op} {d} of {n} exceeds its slot\"\n\t.endif")
                        a(f"\t.org {lab} + {slot}")
        a(f"\t.p2align {self.hl}")
        for lab, fn in self.ool:
            a(f"{lab}:")
            fn()

def call_interp(a, it):
    """Call the interpreter `it` as a subroutine."""
    ret = a.label("ret")
    here = a.label("pc")
    a(f"""
	s_getpc_b64 {S_RET}
{here}:
	s_add_u32 s74, s74, {ret} - {here}
	s_addc_u32 s75, s75, 0
	s_branch .L{it.name}_run
{ret}:""")


def kernel_header(a, name, kernarg, next_vgpr, lds=0, wg_id=False):
    a(f"""
	.text
	.protected {name}
	.globl {name}
	.p2align 8
	.type {name},@function
{name}:""")


def kernel_footer(a, name, kernarg, next_vgpr, next_sgpr, wg_id, wg_y=False):
    a(f"""
	s_endpgm
.L{name}_end:
	.size {name}, .L{name}_end - {name}
	.rodata
	.p2align 6
	.amdhsa_kernel {name}
		.amdhsa_group_segment_fixed_size 0
		.amdhsa_private_segment_fixed_size 0
		.amdhsa_kernarg_size {kernarg}
		.amdhsa_user_sgpr_count 2
		.amdhsa_user_sgpr_kernarg_segment_ptr 1
		.amdhsa_system_sgpr_workgroup_id_x {1 if wg_id else 0}
		.amdhsa_system_sgpr_workgroup_id_y {1 if wg_y else 0}
		.amdhsa_system_sgpr_workgroup_id_z 0
		.amdhsa_system_vgpr_workitem_id 0
		.amdhsa_next_free_vgpr {next_vgpr}
		.amdhsa_next_free_sgpr {next_sgpr}
		.amdhsa_accum_offset {(next_vgpr + 3) // 4 * 4}
		.amdhsa_reserve_vcc 1
		.amdhsa_float_round_mode_32 0
		.amdhsa_float_round_mode_16_64 0
		.amdhsa_float_denorm_mode_32 3
		.amdhsa_float_denorm_mode_16_64 3
		.amdhsa_dx10_clamp 1
		.amdhsa_ieee_mode 1
	.end_amdhsa_kernel
	.text""")


def common_consts(a):
    a(f"""
	v_mov_b32 {V_QNAN}, 0x7fc00000
	v_mov_b32 {V_SQRTC}, 0xf800000
	s_mov_b32 {S_SIGN}, 0x80000000
	s_mov_b32 {S_ABSM}, 0x7fffffff""")


def handler_base(a, it):
    here = a.label("pc")
    a(f"""
	s_getpc_b64 {S_HBASE}
{here}:
	s_add_u32 s42, s42, .L{it.name}_handlers - {here}
	s_addc_u32 s43, s43, 0""")


def lut_bits(it, k):
    """LUT1[opcode k] of the threaded decode: bits 3:0 family U's index + 1 (counted from the family's highest opcode, 0: not of the family),
    7:4 family B's, 8 the op has an in-place form, 9 an RR form (word 1 names a register), 10 INPUT.  The same for every register class."""
    if k >= len(OPS):
        return 0
    op = OPS[k]
    def idx(fam):
        order = sorted(fam, key=OPS.index, reverse=True)
        return order.index(op) + 1 if op in fam else 0
    return idx(Interp.U_OPS) | (idx(Interp.B_OPS) << 4) | ((op in Interp.INPLACE) << 8) | ((op.endswith("_RR")) << 9) | ((op == "INPUT") << 10)


def emit_decode(a, it, inplace_mask):
    """threaded dispatch: the tape words in v[60:61], lane = op -> V_DEC: handler address (the in-place table when out == a and the
    op has such a form; a delta handler - Interp.handler_delta - when the two registers of the op are close enough; an INPUT of an axis
    slot has a handler of its own), file indices out | a << 8, word 1 (the RR forms: b's file index).  What depends on the opcode alone
    comes out of V_LUT1 by ONE ds_bpermute_b32 (lut_bits), an INPUT op's handler out of V_LUT2 by another (76 -> 48 instructions per leaf).
    Clobbers v18 .. v33, vcc, s[90:97]."""
    lg, hl, n = it.lg, it.hl, it.name
    D, ND = it.dmax, it.nd
    a(f"""
	v_and_b32 v18, 0xff, {V_DEC[0]}                          ; opcode
	v_lshlrev_b32 v24, 2, {V_DEC[1]}
	v_lshlrev_b32 v25, 2, v18
	ds_bpermute_b32 v29, v24, {V_LUT2}               ; an INPUT op of this slot: its handler's offset
	ds_bpermute_b32 v28, v25, {V_LUT1}               ; the opcode's bits
	v_bfe_u32 v19, {V_DEC[0]}, 8, 12                         ; out
	v_lshrrev_b32 v20, 20, {V_DEC[0]}                        ; a
	v_mov_b32 {V_DEC[3]}, {V_DEC[1]}
	v_lshlrev_b32 v23, {lg}, {V_DEC[1]}
	v_lshlrev_b32 v26, {lg}, v19
	v_cmp_eq_u32_e64 {S_M[0]}, v19, v20               ; out == a
	v_lshl_or_b32 v26, v20, {lg + 8}, v26            ; file index of out | of a << 8 (each < 256: s_set_gpr_idx_on takes bits 7:0)
	v_lshlrev_b32 v27, {hl}, v18                      ; the handler's offset: generic table ...
	v_add_u32 v30, {64 << hl}, v27                    ; ... in-place table""")
    if D:
        a(f"""
	v_sub_u32 v21, v20, v19
	v_sub_u32 v22, {V_DEC[1]}, v20
	v_add_u32 v21, {D}, v21                           ; distance a - out + D (family U), b - a + D (family B)
	v_add_u32 v22, {D}, v22
	v_cmp_gt_u32_e64 {S_M[1]}, {ND}, v21
	v_cmp_gt_u32_e64 {S_M[2]}, {ND}, v22
	v_cmp_ne_u32_e64 {S_M[3]}, {D}, v22""")
    a(f"""
	s_waitcnt lgkmcnt(0)
	v_bfe_u32 v31, v28, 8, 1                          ; has an in-place form
	v_bfe_u32 v32, v28, 9, 1                          ; RR form
	v_cmp_eq_u32 vcc, 1, v31
	v_cmp_eq_u32_e64 {S_PC}, 1, v32
	v_and_b32 v31, 15, v28                            ; family U: index + 1
	s_and_b64 {S_M[0]}, {S_M[0]}, vcc                 ; in place: out == a and the op has the form
	v_bfe_u32 v32, v28, 4, 4                          ; family B
	v_cndmask_b32_e64 {V_DEC[3]}, {V_DEC[3]}, v23, {S_PC}           ; the RR forms: word 1 = b's file index
	v_cndmask_b32_e64 v27, v27, v30, {S_M[0]}""")
    if D:
        a(f"""
	v_mad_u32_u24 v33, v31, {ND}, v21
	v_cmp_ne_u32 vcc, 0, v31
	v_mul_u32_u24 v33, {it.su}, v33
	s_and_b64 {S_M[1]}, {S_M[1]}, vcc
	s_andn2_b64 {S_M[1]}, {S_M[1]}, {S_M[0]}          ; (U: not in place - which also says distance != 0 for the ops that have the form; the others: test below)
	v_add_u32 v33, .L{n}_du - .L{n}_handlers - {ND * it.su}, v33
	v_cmp_ne_u32 vcc, {D}, v21
	v_mad_u32_u24 v31, v32, {ND}, v22
	s_and_b64 {S_M[1]}, {S_M[1]}, vcc
	v_cmp_ne_u32 vcc, 0, v32
	v_mul_u32_u24 v31, {it.sb}, v31
	v_cndmask_b32_e64 v27, v27, v33, {S_M[1]}
	s_and_b64 {S_M[2]}, {S_M[2]}, vcc
	s_and_b64 {S_M[2]}, {S_M[2]}, {S_M[3]}
	v_add_u32 v31, .L{n}_db - .L{n}_handlers - {ND * it.sb}, v31
	s_and_b64 {S_M[2]}, {S_M[2]}, {S_M[0]}            ; (B: in place)
	s_nop 0
	v_cndmask_b32_e64 v27, v27, v31, {S_M[2]}""")
    a(f"""
	v_bfe_u32 v32, v28, 10, 1                         ; INPUT
	v_cmp_eq_u32 vcc, 1, v32
	v_mov_b32 {V_DEC[1]}, v26
	s_nop 0
	v_cndmask_b32 v27, v27, v29, vcc
	v_add_u32 {V_DEC[0]}, s42, v27""")


def gen_columns(a, variants, off, trans=None):
    """fh_columns: ONE leaf (8x8x8 voxels, one pixel column per lane) per wave pass.  Waves walk the
    leaf table [layer][footprint] front layer first, 64 footprints at a time, round robin without atomics;
    hits go to the z-buffer with a 64-bit atomic max (depth << 32 | leaf), so any interleaving of the
    waves gives the same image, and leaves behind a hit usually find it and retire after one load.
    The register-file shape is chosen per leaf (variants = [(NR, ZB)], smallest NR first: 8 registers
    x 8 voxels, 16 x 4, 32 x 2 - all 64 VGPRs): 80 % of prospero's leaves take a single pass.
    kernarg: { FhRenderState* S; u32 n_waves; u32 axis slots; u32 inputs varying along a column; u32 flags; u32 pad[2] }"""
    kname = "fh_columns_t" if trans else "fh_columns"
    if trans:   # same generator into a scratch buffer, labels renamed, the routines embedded next to the handlers (s_branch range)
        b = Asm()
        b.uid = a.uid + 100000
        set_reg_map("compact")
        try:
            r = _gen_columns_body(b, variants, off, kname, trans)
        finally:
            set_reg_map("default")
        a(b.text().replace(".Lfh_columns_", ".Lfh_columns_t_"))
        return r
    set_reg_map("default" if EXP == "file64" else "compact")
    try:
        return _gen_columns_body(a, variants, off, kname, None)
    finally:
        set_reg_map("default")


def _gen_columns_body(a, variants, off, kname, trans):
    o = off
    m = S_MAT
    file_regs = max(nr * zb for nr, zb in variants)
    # the routines' register window behind the register file: 24 VGPRs - the one-sample routines, sin4 / cos4, the hand-written expf's
    # 22 and its table's two: 48 + 96 + 24 = 168 registers, THREE waves per SIMD (round 6; it was 64 + 128 + 64 = 256 and two: the
    # plain kernel at two / three / four waves takes 0.626 / 0.477 / 0.419 ms per launch on the general path)
    import gen_trans
    t_base = FILE + file_regs
    nvg = t_base + gen_trans.EXP2_WINDOW if trans else FILE + file_regs
    if EXP.startswith("vgpr") and not trans:      # experiment: the plain kernel at a lower occupancy (registers it does not use)
        nvg = int(EXP[4:])
    its = [Interp(a, f"{kname}_{nr}x{zb}", nr, zb, "columns", off, trans=bool(trans)) for nr, zb in variants]
    for it in its:
        it.t_base = t_base
        it.wide_trans = False
        it.exp2 = bool(trans)
    inplace_mask = 0
    for k, op in enumerate(OPS):
        if op in Interp.INPLACE:
            inplace_mask |= 1 << k
    S_WGID, S_NWG, S_CNT, S_I, S_L, S_NFPL = "s6", "s7", "s40", "s41", "s27", "s38"
    S_ONE, S_WGY = "s100", "s101"
    S_RCP = "s33"        # floor(2^32 / blocks per layer), kernarg word 6 (0: none); s33 is free in the leaf kernels (the bulk kernels' S_OUTP)
    import os
    BLKL = int(os.environ.get("FH_BLKL", "2"))   # footprints per work item: 4 (small enough to balance, large enough to skip empty space fast); capi.hip FH_COL_BLKL must agree
    BLK = 1 << BLKL
    # kernarg: { FhRenderState* S; u32 n_waves; u32 axis slots x | y << 8 | z << 16 (0xFF: the tape has no such input);
    #            u32 inputs that change along a pixel column (bit per input slot); u32 flags (bit 16: projective matrix; 20: column mode) }
    # - per frame constants the host works out once: 65 536 workgroups per launch each spent ~130 scalar instructions on them
    # kernarg (48 bytes): { FhRenderState* S; u32 n_waves, axis slots, inputs varying along a column, flags, reciprocal, pad;
    #                       FhLeafRef* leaf table of this slab; u32 footprints per layer; u32 layers of the slab }
    # - what a wave needs to find out whether its part of the table holds a leaf at all comes with the kernarg: two thirds of a launch's
    # waves find none, and leave after ONE dependent load (their entries) instead of two (the state's pointers first); the others have
    # the state's words on their way meanwhile
    kernel_header(a, kname, 48, nvg)
    a(f"""
	s_load_dwordx2 {S_STATE}, {S_KERNARG}, 0x0
	s_load_dwordx4 s[48:51], {S_KERNARG}, 0x8
	s_load_dword {S_RCP}, {S_KERNARG}, 0x18
	s_load_dwordx4 s[60:63], {S_KERNARG}, 0x20
	s_mov_b32 {S_WGID}, s2
	s_mov_b32 {S_WGY}, s3""")
    common_consts(a)
    if trans:       # the table registers of the hand-written expf (gen_trans.py), once per wave
        gen_trans.exp_table_init(a, t_base, lane=V_LANE)
    a(f"""
	s_mov_b32 {V_PINF}, 0x7f800000
	s_mov_b32 s58, 1.0
	s_mov_b32 s59, 1.0
	s_waitcnt lgkmcnt(0)
	s_mov_b64 {S_TABLE}, s[60:61]
	s_mov_b32 {S_NFPL}, s62
	s_mov_b32 {S_L}, s63                              ; layers of the slab
	; the state's words: on their way while the wave looks at its table entries; waited for where the first leaf starts (.Lfh_columns_leaf)
	s_load_dwordx16 s[{m}:{m + 15}], {S_STATE}, {o['P.mat']}
	s_load_dwordx2 s[24:25], {S_STATE}, {o['P.width']}
	s_load_dwordx2 {S_ARENA}, {S_STATE}, {o['arena']}
	s_load_dwordx2 {S_ZBUF}, {S_STATE}, {o['zbuf']}
	s_load_dword {S_SLABZ}, {S_STATE}, {o['slab_z']}
	s_mov_b32 {V_NINF}, 0xff800000
	s_mov_b32 {S_NWG}, s48
	s_bfe_i32 {S_SLOTX}, s49, 0x80000               ; (s0 / s1 held the kernarg pointer until here)
	s_bfe_i32 {S_SLOTY}, s49, 0x80008
	s_bfe_i32 {S_SLOTZ}, s49, 0x80010
	s_mov_b32 {S_DEPMASK}, s50
	s_and_b32 s51, s51, 0x0f1f0000                   ; flags bit 16: projective; 17 .. 19: x / y / z of the model changes along a pixel column; 20: column mode; 24 .. 27: log2 of the layers a wave of column mode takes (bit 28, set here: the lane's pixel state is valid)
	s_or_b32 {S_WGY}, {S_WGY}, s51
	; ({S_DEPMASK}, from the kernarg: the input slots whose value changes along a pixel column - the axis' matrix row has a z
	; coefficient, or the matrix is projective.  A leaf tape that reads none of them has ONE value per pixel for its 8
	; voxels and is evaluated once per pixel, below: a vertical wall, an extrusion, whatever pruning left independent of z.)
	; footprints per layer, blocks of {BLK} of them
	s_add_u32 {S_CNT}, {S_NFPL}, {BLK - 1}
	s_lshr_b32 {S_CNT}, {S_CNT}, {BLKL}
	; work items = (layer, block), front layer first.  n_waves != 0: persistent waves, wave w takes
	; items w, w + n_waves, ...; n_waves == 0: one item per workgroup of a (blocks, layers) grid -
	; short workgroups let the hardware balance them and let the tile stage of the next slab
	; (other stream) slip in between
	s_sub_u32 {S_L}, {S_L}, 1
	s_mov_b32 {S_I}, {S_WGID}
	s_mov_b32 {S_ONE}, 0
	s_cmp_eq_u32 {S_NWG}, 0
	s_cbranch_scc0 .Lfh_columns_block
	s_mov_b32 {S_ONE}, 1
	s_and_b32 {S_T0}, {S_WGY}, 0xffff
	s_bitcmp1_b32 {S_WGY}, 20
	s_cbranch_scc0 .Lfh_columns_ylayer
	s_bfe_u32 {S_T1}, {S_WGY}, 0x40018                ; column mode: the workgroup's y counts groups of 2^g layers, front group first
	s_lshl_b32 {S_T0}, {S_T0}, {S_T1}
.Lfh_columns_ylayer:
	s_sub_u32 {S_L}, {S_L}, {S_T0}
	s_cbranch_scc1 .Lfh_columns_exit
.Lfh_columns_block:
	; ---- next block of {BLK} footprints (lane = footprint) --------------------------------------
	s_cmp_ge_u32 {S_ONE}, 2
	s_cbranch_scc1 .Lfh_columns_exit
	s_add_u32 {S_ONE}, {S_ONE}, {S_ONE}
	s_bitcmp1_b32 {S_WGY}, 20
	s_cbranch_scc1 .Lfh_columns_column
	s_cmp_ge_u32 {S_I}, {S_CNT}
	s_cbranch_scc0 .Lfh_columns_haveblock
	s_sub_u32 {S_I}, {S_I}, {S_CNT}
	s_sub_u32 {S_L}, {S_L}, 1
	s_cbranch_scc1 .Lfh_columns_exit
	s_branch .Lfh_columns_block
.Lfh_columns_haveblock:
	; rotate the block index by a per-layer offset: with a round-robin stride equal to the number of
	; blocks a wave would otherwise own the same footprints in every layer (no balance at all)
	; (the rotation is what balances the launch - without it the heavy footprints' layers pile up on the same compute units: 0.58 -> 0.82 ms
	; per launch with z in every tape, measured - so it stays, but as a multiplication: (L * 1237 + I) mod CNT through the reciprocal the
	; host passes in the kernarg, floor(2^32 / CNT), one conditional subtraction behind it; the subtraction loop below, which cost the
	; average wave ten taken branches before it had looked at a leaf, only runs when no reciprocal came)
	s_mul_i32 {S_T0}, {S_L}, 1237
	s_add_u32 {S_T0}, {S_T0}, {S_I}
	s_cmp_eq_u32 {S_RCP}, 0
	s_cbranch_scc1 .Lfh_columns_rot
	s_mul_hi_u32 {S_T1}, {S_T0}, {S_RCP}
	s_mul_i32 {S_T1}, {S_T1}, {S_CNT}
	s_sub_u32 {S_T0}, {S_T0}, {S_T1}
	s_sub_u32 {S_T1}, {S_T0}, {S_CNT}
	s_cmp_ge_u32 {S_T0}, {S_CNT}
	s_cselect_b32 {S_T0}, {S_T1}, {S_T0}
.Lfh_columns_rot:
	s_cmp_ge_u32 {S_T0}, {S_CNT}
	s_cbranch_scc0 .Lfh_columns_rotated
	s_sub_u32 {S_T0}, {S_T0}, {S_CNT}
	s_branch .Lfh_columns_rot
.Lfh_columns_rotated:
	s_lshl_b32 {S_T0}, {S_T0}, {BLKL}
	s_add_u32 {S_I}, {S_I}, {S_NWG}
	v_add_u32 {V_S0}, {S_T0}, {V_LANE}
	v_cmp_gt_u32 vcc, {S_NFPL}, {V_S0}
	v_cmp_gt_u32_e64 {S_M[0]}, {BLK}, {V_LANE}
	s_nop 3
	s_and_b64 vcc, vcc, {S_M[0]}
	s_mul_i32 {S_T1}, {S_L}, {S_NFPL}
	s_add_u32 {S_T1}, {S_T1}, {S_T0}
	s_lshl_b32 {S_T1}, {S_T1}, 4                     ; 16-byte entries (FhLeafRef)
	s_add_u32 s86, s34, {S_T1}
	s_addc_u32 s87, s35, 0
	v_lshlrev_b32 {V_S0}, 4, {V_LANE}
	v_mov_b32 {V_ENT[0]}, 0
	s_and_saveexec_b64 {S_SAVE}, vcc
	global_load_dwordx4 v[{V_ENT[0][1:]}:{V_ENT[3][1:]}], {V_S0}, {S_PC}
	s_mov_b64 exec, {S_SAVE}
	s_lshl_b32 {S_LZ}, {S_L}, 3                      ; this layer's z: every leaf of the block has it
	s_waitcnt vmcnt(0) lgkmcnt(0)
	s_add_u32 {S_LZ}, {S_LZ}, {S_SLABZ}
	v_cmp_ne_u32 vcc, 0, {V_ENT[0]}
	s_mov_b32 {S_NXTV}, 0
	s_nop 2
	s_mov_b64 {S_LAYMASK}, vcc
	s_branch .Lfh_columns_leaf
.Lfh_columns_column:
	; ---- column mode (kernarg flags bit 20; grid = footprints rounded up to 64, one layer row): the wave takes ONE footprint and
	; its whole column of the slab's table, lane = layer, front layer in lane 0.  For frames whose tapes read nothing that changes
	; along a pixel column a slab's column holds at most one leaf (the nearest of a stack is the only one queued), the table is
	; nearly empty, and the (blocks, layers) grid above is 16 x as many workgroups that find nothing: 66 of the launch's 71 us
	; at 1024^3.  Workgroup ids go round the 8 XCDs: id = 64 q + 8 a + b runs on XCD b and takes footprint 64 q + 8 b + a, so that
	; the eight footprints whose entries share a 128-byte line of a layer's row are read through one L2.
	s_and_b32 {S_T0}, {S_I}, 7
	s_lshr_b32 {S_T1}, {S_I}, 6
	s_add_u32 {S_T0}, {S_T0}, {S_T1}                  ; (the XCD's eight footprints move one place per group of 64: a fixed place would
	s_and_b32 {S_T0}, {S_T0}, 7                       ; give each XCD the same vertical stripes of every image row - and their geometry)
	s_bfe_u32 {S_T1}, {S_I}, 0x30003
	s_andn2_b32 s86, {S_I}, 63
	s_lshl_b32 {S_T0}, {S_T0}, 3
	s_add_u32 {S_T0}, {S_T0}, s86
	s_add_u32 {S_T0}, {S_T0}, {S_T1}
	s_cmp_ge_u32 {S_T0}, {S_NFPL}
	s_cbranch_scc1 .Lfh_columns_exit
	s_bfe_u32 {S_T1}, {S_WGY}, 0x40018
	s_lshl_b32 {S_T1}, 1, {S_T1}                      ; the layers of this wave: 2^g from {S_L} (the group's front layer) back, lane = layer
	v_sub_u32 {V_S0}, {S_L}, {V_LANE}
	v_cmp_ge_u32 vcc, {S_L}, {V_LANE}
	v_cmp_gt_u32_e64 {S_M[0]}, {S_T1}, {V_LANE}
	v_mul_lo_u32 {V_S0}, {V_S0}, {S_NFPL}
	v_mov_b32 {V_ENT[0]}, 0
	v_add_lshl_u32 {V_S0}, {V_S0}, {S_T0}, 4            ; 16-byte entries, [layer][footprint]
	s_and_b64 vcc, vcc, {S_M[0]}
	s_and_saveexec_b64 {S_SAVE}, vcc
	global_load_dwordx4 v[{V_ENT[0][1:]}:{V_ENT[3][1:]}], {V_S0}, {S_TABLE}
	s_mov_b64 exec, {S_SAVE}
	s_mov_b32 {S_NXTV}, 0
	s_waitcnt vmcnt(0)                              ; (the entries alone: a wave that finds none leaves without the state's words)
	v_cmp_ne_u32 vcc, 0, {V_ENT[0]}
	s_nop 3
	s_mov_b64 {S_LAYMASK}, vcc
.Lfh_columns_leaf:
	; ---- next leaf of the block: everything needed to start on it is in the entries (no load before the tape's) --------
	s_cmp_eq_u64 {S_LAYMASK}, 0
	s_cbranch_scc1 .Lfh_columns_blockend
	s_waitcnt lgkmcnt(0)                            ; (the state's words, requested in the prologue)
	s_ff1_i32_b64 {S_ZL}, {S_LAYMASK}
	s_bitset0_b64 {S_LAYMASK}, {S_ZL}
	s_bitcmp1_b32 {S_WGY}, 20
	s_cbranch_scc0 .Lfh_columns_layerz
	s_sub_u32 {S_T0}, {S_L}, {S_ZL}                  ; column mode: the lane is the layer, counted from the front
	s_lshl_b32 {S_T0}, {S_T0}, 3
	s_add_u32 {S_LZ}, {S_T0}, {S_SLABZ}
.Lfh_columns_layerz:
	s_nop 0
	v_readlane_b32 {S_ID}, {V_ENT[0]}, {S_ZL}
	v_readlane_b32 s84, {V_ENT[1]}, {S_ZL}
	v_readlane_b32 {S_T0}, {V_ENT[2]}, {S_ZL}
	v_readlane_b32 {S_T1}, {V_ENT[3]}, {S_ZL}
	s_mov_b32 s85, 0
	s_nop 1
	s_and_b32 {S_LEN0}, {S_T0}, 0xffffff
	s_lshr_b32 {S_RC}, {S_T0}, 24
	s_and_b32 {S_FX}, {S_T1}, 0xffff
	s_lshr_b32 {S_FY}, {S_T1}, 16
	s_cmp_gt_u32 {S_RC}, {its[-1].nr}                          ; needs the LDS register file: left to k_leaves3d<2> (never requested ahead)
	s_cbranch_scc1 .Lfh_columns_leaf
	s_lshl_b64 {S_TBASE}, {S_TBASE}, 3
	s_add_u32 s84, s84, s30
	s_addc_u32 s85, s85, s31
	; The tape: requested while the leaf before this one was interpreted (it is in V_NXT, or about to be), or requested now -
	; up to 64 ops by one vector load, lane = op, decoded below by vector code (the scalar unit, which all waves of a CU
	; share, then only jumps); longer tapes (1 % of prospero's leaves) keep the scalar fetch.
	s_cmp_eq_u32 {S_NXTV}, 0
	s_cbranch_scc1 .Lfh_columns_request
	; (in flight: the requested tape and, if the last leaf had hits, its z-buffer atomic - loads and atomics complete in no
	; particular order with each other, so both are waited for)
	s_waitcnt vmcnt(0)
	v_mov_b32 {V_DEC[0]}, {V_NXT[0]}
	v_mov_b32 {V_DEC[1]}, {V_NXT[1]}
	s_branch .Lfh_columns_taperequested
.Lfh_columns_request:
	s_cmp_gt_u32 {S_LEN0}, 64
	s_cbranch_scc1 .Lfh_columns_longtape
	v_lshlrev_b32 {V_S3}, 3, {V_LANE}
	s_sub_u32 {S_T0}, 64, {S_LEN0}
	s_lshr_b64 exec, -1, {S_T0}
	global_load_dwordx2 v[{V_DEC[0][1:]}:{V_DEC[1][1:]}], {V_S3}, {S_TBASE}
	s_mov_b64 exec, -1
	s_branch .Lfh_columns_taperequested
.Lfh_columns_longtape:
	s_nop 0                                         ; (a long tape is fetched and decoded 63 ops at a time, per pass)
.Lfh_columns_taperequested:
	; the decode's tables, once per wave (bit 30): lane = opcode -> its bits (LUT_BITS; a table behind the kernel), lane = input slot -> the
	; handler offset of an INPUT op of that slot (the axes' slots have handlers of their own)
	s_bitcmp1_b32 {S_WGY}, 30
	s_cbranch_scc1 .Lfh_columns_luts
	s_getpc_b64 {S_PC}
.Lfh_columns_lutpc:
	s_add_u32 s86, s86, .L{kname}_lut - .Lfh_columns_lutpc
	s_addc_u32 s87, s87, 0
	v_lshlrev_b32 {V_S4}, 2, {V_LANE}
	global_load_dword {V_LUT1}, {V_S4}, {S_PC}
	v_mov_b32 {V_LUT2}, {OPS.index("INPUT") << its[0].hl}
	v_mov_b32 v18, {52 << its[0].hl}
	v_cmp_eq_u32_e64 {S_M[0]}, {S_SLOTX}, {V_LANE}
	v_cmp_eq_u32_e64 {S_M[1]}, {S_SLOTY}, {V_LANE}
	v_cmp_eq_u32_e64 {S_M[2]}, {S_SLOTZ}, {V_LANE}
	v_mov_b32 v19, {53 << its[0].hl}
	v_mov_b32 v20, {54 << its[0].hl}
	v_cndmask_b32_e64 {V_LUT2}, {V_LUT2}, v18, {S_M[0]}
	v_cndmask_b32_e64 {V_LUT2}, {V_LUT2}, v19, {S_M[1]}
	v_cndmask_b32_e64 {V_LUT2}, {V_LUT2}, v20, {S_M[2]}
	s_bitset1_b32 {S_WGY}, 30
.Lfh_columns_luts:
	; pixel of this lane, its z-buffer word.  Column mode: every leaf of the wave has the same footprint - the pixel, its matrix
	; products and its z-buffer word are set up once (bit 28), hits stay in the lane's registers from leaf to leaf (a pixel hit by a
	; nearer leaf of the wave is not pending for the next) and go to the z-buffer once, behind the wave's last leaf.
	s_mov_b32 {S_NXTV}, 0
	s_bitcmp1_b32 {S_WGY}, 28
	s_cbranch_scc1 .Lfh_columns_pixelset
	v_and_b32 {V_S0}, 7, {V_LANE}
	v_lshrrev_b32 {V_S1}, 3, {V_LANE}
	v_add_u32 {V_S0}, {S_FX}, {V_S0}
	v_add_u32 {V_S1}, {S_FY}, {V_S1}
	v_cvt_f32_u32 {V_PXF}, {V_S0}
	v_cvt_f32_u32 {V_PYF}, {V_S1}
	v_cmp_gt_u32_e64 {S_M[0]}, {S_WIDTH}, {V_S0}
	v_cmp_gt_u32_e64 {S_M[1]}, {S_HEIGHT}, {V_S1}
	v_mul_u32_u24 {V_S2}, {V_S1}, {S_WIDTH}                 ; (y, width < 2^24; the z-buffer is addressed as base + 32-bit byte offset)
	v_add_lshl_u32 {V_PIX}, {V_S2}, {V_S0}, 3
	s_and_b64 {S_M[0]}, {S_M[0]}, {S_M[1]}
	v_mov_b32 {V_DEPTH}, -1                         ; pixels outside the image never become pending
	v_mov_b32 {V_HIT}, 0
	s_mov_b64 {S_SAVE}, exec
	s_mov_b64 exec, {S_M[0]}
	global_load_dword {V_DEPTH}, {V_PIX}, {S_ZBUF} offset:4
	s_mov_b64 exec, {S_SAVE}
	; (m[4r] * x + m[4r+1] * y) per row: constant over the column (dev_ops.hpp xf_point)
	v_mul_f32 {V_AX}, s{m + 0}, {V_PXF}
	v_mul_f32 {V_S0}, s{m + 1}, {V_PYF}
	v_add_f32 {V_AX}, {V_AX}, {V_S0}
	v_mul_f32 {V_AY}, s{m + 4}, {V_PXF}
	v_mul_f32 {V_S0}, s{m + 5}, {V_PYF}
	v_add_f32 {V_AY}, {V_AY}, {V_S0}
	v_mul_f32 {V_AZ}, s{m + 8}, {V_PXF}
	v_mul_f32 {V_S0}, s{m + 9}, {V_PYF}
	v_add_f32 {V_AZ}, {V_AZ}, {V_S0}
	v_mul_f32 {V_AW}, s{m + 12}, {V_PXF}
	v_mul_f32 {V_S0}, s{m + 13}, {V_PYF}
	v_add_f32 {V_AW}, {V_AW}, {V_S0}
	s_bfe_u32 {S_T0}, {S_WGY}, 0x10014               ; column mode (bit 20) -> bit 28: the set-up stands for the wave's other leaves
	s_lshl_b32 {S_T0}, {S_T0}, 28
	s_or_b32 {S_WGY}, {S_WGY}, {S_T0}
.Lfh_columns_pixelset:
	; the next leaf of the block: its tape (if it is one of up to 64 ops for this kernel) is requested now, behind this
	; leaf's own loads, and arrives while this leaf is interpreted
	s_cmp_eq_u64 {S_LAYMASK}, 0
	s_cbranch_scc1 .Lfh_columns_noahead
	s_ff1_i32_b64 {S_T0}, {S_LAYMASK}
	s_nop 0
	v_readlane_b32 {S_T1}, {V_ENT[2]}, {S_T0}
	v_readlane_b32 s86, {V_ENT[1]}, {S_T0}
	s_mov_b32 s87, 0
	s_nop 1
	s_lshr_b32 {S_T0}, {S_T1}, 24
	s_and_b32 {S_T1}, {S_T1}, 0xffffff
	s_cmp_gt_u32 {S_T0}, {its[-1].nr}
	s_cbranch_scc1 .Lfh_columns_noahead
	s_cmp_gt_u32 {S_T1}, 64
	s_cbranch_scc1 .Lfh_columns_noahead
	s_lshl_b64 {S_PC}, {S_PC}, 3
	s_add_u32 s86, s86, s30
	s_addc_u32 s87, s87, s31
	v_lshlrev_b32 {V_S4}, 3, {V_LANE}
	s_sub_u32 {S_T0}, 64, {S_T1}
	s_lshr_b64 exec, -1, {S_T0}
	global_load_dwordx2 v[{V_NXT[0][1:]}:{V_NXT[1][1:]}], {V_S4}, {S_PC}
	s_mov_b64 exec, -1
	s_mov_b32 {S_NXTV}, 1
.Lfh_columns_noahead:
	v_mov_b32 {V_IDV}, {S_ID}
	s_cmp_eq_u32 {S_NXTV}, 0
	s_cbranch_scc1 .Lfh_columns_waitall
	s_waitcnt vmcnt(1)                              ; (the request ahead is the youngest: it may stay in flight)
	s_branch .Lfh_columns_have
.Lfh_columns_waitall:
	s_waitcnt vmcnt(0)
.Lfh_columns_have:
	; pending = depth < lz + 8  (voxel.rs:377-381)
	s_add_u32 {S_T0}, {S_LZ}, 8
	v_cmp_gt_u32 vcc, {S_T0}, {V_DEPTH}
	s_nop 3
	s_mov_b64 {S_PEND}, vcc
	s_cmp_eq_u64 {S_PEND}, 0
	s_cbranch_scc1 .Lfh_columns_nopending
	s_mov_b32 {S_K}, 7
	; does the tape read an input that changes along the column?  (tapes of up to 64 ops: the words are in v[60:61], lane = op)
	s_mov_b32 {S_INV}, 0
	s_cmp_gt_u32 {S_LEN0}, 64
	s_cbranch_scc1 .Lfh_columns_zdep
	v_and_b32 v18, 0xff, {V_DEC[0]}
	v_lshrrev_b32_e64 v19, {V_DEC[1]}, {S_DEPMASK}
	v_and_b32 v19, 1, v19
	v_cmp_eq_u32 vcc, {OPS.index("INPUT")}, v18
	v_cmp_eq_u32_e64 {S_M[1]}, 1, v19
	s_sub_u32 {S_T0}, 64, {S_LEN0}
	s_lshr_b64 {S_M[2]}, -1, {S_T0}
	s_nop 1
	s_and_b64 {S_M[1]}, {S_M[1]}, vcc
	s_and_b64 {S_M[1]}, {S_M[1]}, {S_M[2]}
	s_cmp_eq_u64 {S_M[1]}, 0
	s_cbranch_scc0 .Lfh_columns_zdep
	s_mov_b32 {S_INV}, 1                          ; one pass of the two-sample class, its first sample is the column's value
	s_branch .L{its[-1].name}_chunk
.Lfh_columns_zdep:""")
    for it in its[:-1]:
        a(f"\ts_cmp_le_u32 {S_RC}, {it.nr}\n\ts_cbranch_scc1 .L{it.name}_chunk")
    a(f"\ts_branch .L{its[-1].name}_chunk")
    for it in its:
        name, zb = it.name, it.zb
        a(f"""
.L{name}_chunk:
	s_getpc_b64 {S_HBASE}
.L{name}_hb:
	s_add_u32 s42, s42, .L{name}_handlers - .L{name}_hb
	s_addc_u32 s43, s43, 0""")
        if True:       # (threaded dispatch: the only form of the leaf kernel)
            ret, here = a.label("ret"), a.label("pc")
            a(f"""
	s_cmp_gt_u32 {S_LEN0}, 64
	s_cbranch_scc1 .L{name}_pass""")
            emit_decode(a, it, inplace_mask)
            a(f".L{name}_pass:")        # (VRES needs no initial value: a shape tape ends with its OUTPUT op, whose handler fills it)
            if EXP == "nointerp":     # experiment: the set-up alone
                for j in range(zb):
                    a(f"\tv_mov_b32 {VRES[j]}, 1.0")
                a(f"\ts_branch {ret}")
            a(f"""
	s_getpc_b64 {S_RET}
{here}:
	s_add_u32 s74, s74, {ret} - {here}
	s_addc_u32 s75, s75, 0
	s_cmp_gt_u32 {S_LEN0}, 64
	s_cbranch_scc0 .L{name}_togov
	; a tape of more than 64 ops: 63 at a time, lane 63 holding a pseudo-op whose handler (NEXT_CHUNK) comes back here
	s_mov_b64 {S_TCUR}, {S_TBASE}
	s_mov_b32 {S_REM}, {S_LEN0}
	s_getpc_b64 {S_LONG}
.L{name}_longpc:
	s_add_u32 s56, s56, .L{name}_longchunk - .L{name}_longpc
	s_addc_u32 s57, s57, 0
.L{name}_longchunk:
	s_set_gpr_idx_off
	s_min_u32 {S_T0}, {S_REM}, 64
	s_sub_u32 {S_T0}, 64, {S_T0}
	v_lshlrev_b32 {V_S3}, 3, {V_LANE}
	s_lshr_b64 exec, -1, {S_T0}
	global_load_dwordx2 v[{V_DEC[0][1:]}:{V_DEC[1][1:]}], {V_S3}, {S_TCUR}
	s_mov_b64 exec, -1
	s_waitcnt vmcnt(0)""")
            emit_decode(a, it, inplace_mask)
            a(f"""
	s_cmp_le_u32 {S_REM}, 64
	s_cbranch_scc1 .L{name}_togov
	s_add_u32 s86, s42, {55 << it.hl}                 ; NEXT_CHUNK's handler
	s_add_u32 s52, s52, {63 * 8}
	s_addc_u32 s53, s53, 0
	s_sub_u32 {S_REM}, {S_REM}, 63
	v_writelane_b32 {V_DEC[0]}, s86, 63
.L{name}_togov:
	; the first op's decoded words, and into its handler (the handlers lie beyond a branch's 128 KB: {S_JN} is a full address)
	s_set_gpr_idx_off
	s_mov_b32 {S_LEN}, 0
	s_mov_b32 {S_JN_HI}, s43
	v_readlane_b32 {S_JN_LO}, {V_DEC[0]}, {S_LEN}
	v_readlane_b32 {S_NX_LO}, {V_DEC[1]}, {S_LEN}
	v_readlane_b32 {S_NX_HI}, {V_DEC[3]}, {S_LEN}
	s_mov_b32 {S_LEN}, 1
	s_setpc_b64 {S_JN}
{ret}:""")
        # first voxel inside, front to back (sample j: depth = lz + (k - j) + 1): the samples' "value < 0" bits shifted into a mask
        # through the carry (sample 0 ends up highest), its leading bit is the hit - 2 instructions per sample instead of 8
        # ... unless no pending pixel has a sample inside at all, which is most leaves of a frame (prospero.vm's general path: 95 %): the
        # smallest of the lane's samples (v_min3_f32 drops NaNs; a -0 is not inside) against 0 first - 5 vector instructions instead of 20
        nohit = a.label("nohit")
        if zb >= 4:
            a(f"\tv_min3_f32 {V_S1}, {VRES[0]}, {VRES[1]}, {VRES[2]}")
            for j in range(3, zb - 1, 2):
                a(f"\tv_min3_f32 {V_S1}, {V_S1}, {VRES[j]}, {VRES[j + 1]}")
            a(f"\tv_min_f32 {V_S1}, {V_S1}, {VRES[zb - 1]}")
            a(f"""
	v_cmp_gt_f32 vcc, 0, {V_S1}
	s_nop 0
	s_and_b64 {S_M[0]}, vcc, {S_PEND}                ; (scc = a pending pixel has a sample inside)
	s_cbranch_scc0 {nohit}""")
        a(f"\tv_mov_b32 {V_S0}, 0")
        for j in range(zb):
            a(f"\tv_cmp_gt_f32 vcc, 0, {VRES[j]}\n\tv_addc_co_u32 {V_S0}, vcc, {V_S0}, {V_S0}, vcc")
        a(f"""
	v_ffbh_u32 {V_S1}, {V_S0}                          ; = 32 - zb + j of the first sample inside
	v_cmp_ne_u32 vcc, 0, {V_S0}
	s_add_u32 {S_T0}, {S_LZ}, {S_K}
	s_add_u32 {S_T0}, {S_T0}, {33 - zb}
	v_sub_u32 {V_S1}, {S_T0}, {V_S1}
	s_and_b64 {S_M[0]}, vcc, {S_PEND}
	s_andn2_b64 {S_PEND}, {S_PEND}, {S_M[0]}
	v_cndmask_b32_e64 {V_DEPTH}, {V_DEPTH}, {V_S1}, {S_M[0]}
	v_cndmask_b32_e64 {V_HIT}, {V_HIT}, {V_IDV}, {S_M[0]}
{nohit}:""")
        if zb < 8:
            a(f"""
	s_cmp_eq_u64 {S_PEND}, 0
	s_cbranch_scc1 .Lfh_columns_leaf_done
	s_cmp_lg_u32 {S_INV}, 0
	s_cbranch_scc1 .Lfh_columns_leaf_done
	s_sub_u32 {S_K}, {S_K}, {zb}
	s_cbranch_scc0 .L{name}_pass
	s_branch .Lfh_columns_leaf_done""")
        else:
            a("\ts_branch .Lfh_columns_leaf_done")
    a(f"""
.Lfh_columns_leaf_done:
	s_bitcmp1_b32 {S_WGY}, 20
	s_cbranch_scc1 .Lfh_columns_leaf_drain          ; (column mode: the hits go out behind the wave's last leaf)
	; z-buffer word = max(word, depth << 32 | leaf) for the lanes that were hit
	v_cmp_ne_u32 vcc, 0, {V_HIT}
	s_and_saveexec_b64 {S_SAVE}, vcc
	global_atomic_umax_x2 {V_PIX}, v[4:5], {S_ZBUF}
	s_mov_b64 exec, {S_SAVE}
.Lfh_columns_leaf_drain:
	s_waitcnt lgkmcnt(0)                            ; an unused tape-head request may still be in flight
	s_branch .Lfh_columns_leaf
.Lfh_columns_nopending:
	; no pixel of the footprint is pending for this leaf.  Column mode: nor will one be for the leaves behind it (depths only grow, the
	; leaves' z only falls) - the wave is done
	s_bitcmp1_b32 {S_WGY}, 20
	s_cbranch_scc0 .Lfh_columns_leaf_drain
	s_waitcnt vmcnt(0) lgkmcnt(0)                   ; (the next leaf's tape may be on its way)
.Lfh_columns_blockend:
	s_bitcmp1_b32 {S_WGY}, 28
	s_cbranch_scc0 .Lfh_columns_block
	v_cmp_ne_u32 vcc, 0, {V_HIT}
	s_and_saveexec_b64 {S_SAVE}, vcc
	global_atomic_umax_x2 {V_PIX}, v[4:5], {S_ZBUF}
	s_mov_b64 exec, {S_SAVE}
	s_branch .Lfh_columns_block
.Lfh_columns_exit:""")
    kernel_footer(a, kname, 48, nvg, 102, True, wg_y=True)
    a(f"\t.p2align 8\n.L{kname}_lut:")
    for k in range(64):
        a(f"\t.long {lut_bits(its[0], k)}")
    if trans:
        import gen_trans
        gen_trans.embed(a, trans, v_base=t_base, exp2=True)
    for it in its:
        it.emit()
    return kname, nvg


def gen_bulk(a, nr, zb, off, trans=None):
    """fh_float_eval: kernarg = {tape*, vars*, out*, len u32, n u32}; vars / out are [slot][n].
    trans: the variant for tapes with transcendental / modulo / rng opcodes (fh_float_eval_<nr>x<zb>_t): the handlers call the compiled
    routines, embedded behind the kernel with a register window of 26 VGPRs behind the register file."""
    name = f"fh_float_eval_{nr}x{zb}" + ("_t" if trans else "")
    it = Interp(a, name, nr, zb, "bulk", off, trans=bool(trans))
    n_vgpr = FILE + nr * zb + (26 if trans else 0)
    if trans:
        it.t_base = FILE + nr * zb
        it.t_prefix = f"fh_tb{nr}_"
        # sinf / cosf written by hand inside the handlers (gen_trans.sincos_pair: the mesher's gyroid is made of them - the compiled four-sample
        # routines were 424 instructions per call, these are 130 for four samples: the mesher's leaf stage at depth 10 0.214 -> 0.201 s)
        it.sincos2 = EXP != "compiledsincos"      # (experiment: the compiled four-sample routines, as until round 6)
        it.wide_trans = False if it.sincos2 else "sincos"
    kernel_header(a, name, 32, n_vgpr)
    a(f"""
	s_load_dwordx2 {S_TAPE}, {S_KERNARG}, 0x0
	s_load_dwordx2 {S_VARS}, {S_KERNARG}, 0x8
	s_load_dwordx2 {S_OUTP}, {S_KERNARG}, 0x10
	s_load_dwordx2 s[46:47], {S_KERNARG}, 0x18""")
    common_consts(a)
    handler_base(a, it)
    a(f"""
	s_waitcnt lgkmcnt(0)
	s_mov_b32 {S_N}, s47
	s_mul_i32 {S_T0}, {S_WG}, {64 * zb}""")
    for j in range(zb):
        # sample index of slot j; inactive lanes read sample 0 and store nothing
        a(f"""
	v_add_u32 {VOFF[j]}, {S_T0}, {V_LANE}
	v_add_u32 {VOFF[j]}, {64 * j}, {VOFF[j]}
	v_cmp_gt_u32_e64 {S_ACT[j]}, {S_N}, {VOFF[j]}
	s_nop 1
	v_cndmask_b32_e64 {VOFF[j]}, 0, {VOFF[j]}, {S_ACT[j]}
	v_lshlrev_b32 {VOFF[j]}, 2, {VOFF[j]}""")
    call_interp(a, it)
    kernel_footer(a, name, 32, n_vgpr, 100, True)
    it.emit()
    if trans:
        import gen_trans
        gen_trans.embed(a, trans, v_base=it.t_base, prefix=it.t_prefix, sincos2=it.sincos2, wide=False if it.sincos2 else "sincos")
    return name


def gen_trans_probe(a):
    """fh_trans_probe: every embedded copy of the compiled routines (gen_trans.COPIES), callable on its own.
    kernarg = {out*, first u32, copy u32, fn u32}; lane l of workgroup w evaluates routine `fn` (index into gen_trans.FUNCS + FUNCS4) of
    copy `copy` on the four floats with bit patterns first + 4 (64 w + l) + j, second argument bits x * 2654435761 + 0x9E3779B9, and
    stores the results at out[4 (64 w + l) + j].  fhip_debug_trans_probe compares them with the inlined routines of the HIP kernels
    (which tests/test_gpu_math.py holds against the host libm over all 2^32 inputs): the renamed, compacted text the hot kernels run
    is then checked itself, not only the source it came from.  An unknown (copy, fn) leaves `out` untouched."""
    import gen_trans
    name = "fh_trans_probe"
    nvg = 256
    kernel_header(a, name, 24, nvg, wg_id=True)
    a(f"""
	s_load_dwordx4 s[4:7], s[0:1], 0x0
	s_load_dword s8, s[0:1], 0x10
	v_lshl_add_u32 v1, s2, 6, v0
	v_lshlrev_b32 v1, 2, v1                         ; element index of the lane's first float
	s_mov_b32 s10, 0x9E3779B1                       ; 2654435761
	s_mov_b32 s11, 0x9E3779B9
	s_waitcnt lgkmcnt(0)""")
    for j in range(4):
        a(f"""
	v_add3_u32 v{2 + j}, v1, s6, {j}
	v_mul_lo_u32 v{6 + j}, v{2 + j}, s10
	v_add_u32 v{6 + j}, s11, v{6 + j}""")
    all_fns = gen_trans.FUNCS + gen_trans.FUNCS4
    for ci, (prefix, vb, fns) in enumerate(gen_trans.COPIES):
        nxt = a.label("copy")
        a(f"\ts_cmp_lg_u32 s7, {ci}\n\ts_cbranch_scc1 {nxt}")
        if "exp2" in fns:       # (the hand-written expf keeps its table in two registers of the window)
            gen_trans.exp_table_init(a, vb, prefix, lane="v0")
        for fn in fns:
            skip = a.label("fn")
            if fn in ("exp2", "ln2", "sin2", "cos2"):        # the two-sample routines written by hand that the EXP / LN / SIN / COS handlers hold (gen_trans.exp_pair ...), under the four-sample routines' numbers
                one = fn[:-1]
                consts, special, two = gen_trans.hand(one)
                slow = a.label("probe_special")
                a(f"\ts_cmp_lg_u32 s8, {all_fns.index(one + '4')}\n\ts_cbranch_scc1 {skip}")
                consts(a, vb)
                special(a, vb, [f"v{2 + k}" for k in range(4)], slow)
                two(a, vb, ["v2", "v3"], ["v10", "v11"])
                two(a, vb, ["v4", "v5"], ["v12", "v13"])
                a(f"\ts_branch .Lfh_trans_probe_store\n{slow}:")
                for k in range(4):
                    here, ret, h2 = a.label("call"), a.label("ret"), a.label("far")
                    a(f"""
	v_mov_b32 v{vb}, v{2 + k}
	s_getpc_b64 s[96:97]
{here}:
	s_add_u32 s96, s96, {ret} - {here}
	s_addc_u32 s97, s97, 0
	s_getpc_b64 s[98:99]
{h2}:
	s_mov_b32 s100, {prefix}{one} - {h2}
	s_ashr_i32 s101, s100, 31
	s_add_u32 s98, s98, s100
	s_addc_u32 s99, s99, s101
	s_setpc_b64 s[98:99]
{ret}:
	v_mov_b32 v{10 + k}, v{vb}""")
                a(f"\ts_branch .Lfh_trans_probe_store\n{skip}:")
                continue
            fi = all_fns.index(fn)
            a(f"\ts_cmp_lg_u32 s8, {fi}\n\ts_cbranch_scc1 {skip}")
            for j in ([0] if fn.endswith("4") else range(4)):
                if fn.endswith("4"):
                    for k in range(4):
                        a(f"\tv_mov_b32 v{vb + k}, v{2 + k}")
                else:
                    a(f"\tv_mov_b32 v{vb}, v{2 + j}\n\tv_mov_b32 v{vb + 1}, v{6 + j}")
                here, ret, h2 = a.label("call"), a.label("ret"), a.label("far")
                # (the copies lie anywhere in the code object: a computed jump, not s_branch's 128 KB)
                a(f"""
	s_getpc_b64 s[96:97]
{here}:
	s_add_u32 s96, s96, {ret} - {here}
	s_addc_u32 s97, s97, 0
	s_getpc_b64 s[98:99]
{h2}:
	s_mov_b32 s100, {prefix}{fn} - {h2}
	s_ashr_i32 s101, s100, 31
	s_add_u32 s98, s98, s100
	s_addc_u32 s99, s99, s101
	s_setpc_b64 s[98:99]
{ret}:""")
                if fn.endswith("4"):
                    for k in range(4):
                        a(f"\tv_mov_b32 v{10 + k}, v{vb + k}")
                else:
                    a(f"\tv_mov_b32 v{10 + j}, v{vb}")
            a(f"\ts_branch .Lfh_trans_probe_store\n{skip}:")
        a(f"\ts_endpgm\n{nxt}:")
    a(f"""
	s_endpgm
.Lfh_trans_probe_store:
	v_lshlrev_b32 v1, 2, v1
	global_store_dwordx4 v1, v[10:13], s[4:5]""")
    kernel_footer(a, name, 24, nvg, 102, True)
    return name, 24, nvg, [(8, "global_buffer")] + [(4, "by_value")] * 4


def gen_probe(a):
    """fh_probe: does M0-relative VGPR addressing apply to packed (VOP3P) operands?  out[0..7] (diagnostics)."""
    name = "fh_probe"
    kernel_header(a, name, 8, 40)
    a(f"""
	s_load_dwordx2 s[4:5], s[0:1], 0x0
	v_mov_b32 v10, 1.0
	v_mov_b32 v11, 2.0
	v_mov_b32 v12, 4.0
	v_mov_b32 v13, 8.0
	v_mov_b32 v20, 16.0
	v_mov_b32 v21, 32.0
	v_mov_b32 v30, 0
	v_mov_b32 v31, 0
	v_mov_b32 v32, 0
	v_mov_b32 v33, 0
	v_mov_b32 v34, 0
	v_mov_b32 v35, 0
	v_mov_b32 v36, 0
	v_mov_b32 v37, 0
	s_mov_b32 s10, 2
	s_mov_b32 s12, 0x40400000
	s_mov_b32 s13, 0x40a00000
	s_set_gpr_idx_on s10, {SRC0}
	v_pk_add_f32 v[30:31], v[10:11], v[20:21]          ; src0 relative: (4 + 16, 8 + 32) = (20, 40); not: (17, 34)
	s_set_gpr_idx_off
	s_set_gpr_idx_on s10, {DST}
	v_pk_mul_f32 v[32:33], v[10:11], v[20:21]          ; dst relative: lands in v[34:35] = (16, 64)
	s_set_gpr_idx_off
	s_set_gpr_idx_on s10, {SRC0 | DST}
	v_pk_mul_f32 v[10:11], v[10:11], s[12:13] op_sel_hi:[1,0]   ; in place on v[12:13] with a broadcast scalar: (12, 24)
	s_set_gpr_idx_off
	v_pk_mov_b32 v[36:37], v[12:13], v[12:13] op_sel:[0,1]
	; does the index mode reach v_readlane_b32?  rows 9 .. 12: (SRC0 relative) 64 = no, 128 = yes; (DST relative) value read,
	; s17 (stays 0 unless the scalar destination was displaced); v_readfirstlane under SRC0 relative
	v_mov_b32 v14, 0x42800000
	v_mov_b32 v16, 0x43000000
	s_mov_b32 s14, 0
	s_mov_b32 s15, 0
	s_mov_b32 s16, 0
	s_mov_b32 s17, 0
	s_mov_b32 s18, 0
	s_mov_b32 s19, 0
	s_mov_b32 s20, 0
	s_nop 4
	s_set_gpr_idx_on s10, {SRC0}
	v_readlane_b32 s14, v14, 0
	s_set_gpr_idx_off
	s_set_gpr_idx_on s10, {DST}
	v_readlane_b32 s15, v14, 0
	s_set_gpr_idx_off
	s_set_gpr_idx_on s10, {SRC0}
	v_readfirstlane_b32 s18, v14
	s_set_gpr_idx_off
	s_nop 4
	v_mov_b32 v22, s14
	v_mov_b32 v23, s15
	v_mov_b32 v24, s17
	v_mov_b32 v25, s18
	v_lshlrev_b32 v1, 2, v0
	s_waitcnt lgkmcnt(0)
	global_store_dword v1, v30, s[4:5]
	global_store_dword v1, v31, s[4:5] offset:256
	global_store_dword v1, v32, s[4:5] offset:512
	global_store_dword v1, v34, s[4:5] offset:768
	global_store_dword v1, v35, s[4:5] offset:1024
	global_store_dword v1, v12, s[4:5] offset:1280
	global_store_dword v1, v13, s[4:5] offset:1536
	global_store_dword v1, v36, s[4:5] offset:1792
	global_store_dword v1, v37, s[4:5] offset:2048
	global_store_dword v1, v22, s[4:5] offset:2304
	global_store_dword v1, v23, s[4:5] offset:2560
	global_store_dword v1, v24, s[4:5] offset:2816
	global_store_dword v1, v25, s[4:5] offset:3072""")
    kernel_footer(a, name, 8, 40, 24, True)
    return name, 8, 40, [(8, "global_buffer")]


def metadata(a, kernels):
    a("\t.amdgpu_metadata\n---\namdhsa.kernels:")
    for name, kernarg, vgprs, args in kernels:
        a("  - .agpr_count:     0\n    .args:")
        offp = 0
        for size, kind in args:
            if kind == "global_buffer":
                a(f"      - .address_space:  global\n        .offset:         {offp}\n        .size:           {size}\n        .value_kind:     global_buffer")
            else:
                a(f"      - .offset:         {offp}\n        .size:           {size}\n        .value_kind:     by_value")
            offp += size
        a(f"""    .group_segment_fixed_size: 0
    .kernarg_segment_align: 8
    .kernarg_segment_size: {kernarg}
    .max_flat_workgroup_size: 64
    .name:           {name}
    .private_segment_fixed_size: 0
    .sgpr_count:     106
    .sgpr_spill_count: 0
    .symbol:         {name}.kd
    .uniform_work_group_size: 1
    .uses_dynamic_stack: false
    .vgpr_count:     {vgprs}
    .vgpr_spill_count: 0
    .wavefront_size: 64""")
    a("amdhsa.target:   amdgcn-amd-amdhsa--gfx950\namdhsa.version:\n  - 1\n  - 2\n...\n\t.end_amdgpu_metadata")


def main():
    off = json.load(open(sys.argv[1]))
    a = Asm()
    a('\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"\n\t.amdhsa_code_object_version 6')
    ks = []
    if len(sys.argv) > 3:
        import gen_trans
        gen_trans.tables(a, sys.argv[3])   # the compiled routines' constant tables, once for all the kernels that embed them
    # (register-file shapes of the plain leaf kernel: 10 registers x 8 voxels, 20 x 4, 40 x 2 in the compact map's file of 80 - capi_render.hpp
    # FH_LEAF_REGS / FH_LEAF_REGS_T say what the largest shape takes: leaves beyond it are the C++ kernel's, one voxel per lane and pass)
    n, nvg = gen_columns(a, ((8, 8), (16, 4), (32, 2)) if EXP == "file64" else ((10, 8), (20, 4), (40, 2)), off)
    ks.append((n, 48, nvg, [(8, "global_buffer")] + [(4, "by_value")] * 6 + [(8, "global_buffer")] + [(4, "by_value")] * 2))
    if len(sys.argv) > 3:   # ... and the variant with the transcendental / modulo / rng opcodes (calls the compiled routines)
        # (the compact map with a register file of 96 VGPRs - 12 registers x 8 voxels, 24 x 4, 32 x 2: the tapes that carry these opcodes
        # are smooth blends that prune little - bear.vm's leaves keep 350-430 ops in 17-23 registers -, and two passes of four voxels pay
        # the per-op dispatch half as often as four passes of two; with the routines' window of 24 registers 168 VGPRs, three waves
        # per SIMD - it was 64 + 128 + 64 = 256 and two)
        n, nvg = gen_columns(a, ((22, 8), (44, 4)), off, trans=sys.argv[3])
        ks.append((n, 48, nvg, [(8, "global_buffer")] + [(4, "by_value")] * 6 + [(8, "global_buffer")] + [(4, "by_value")] * 2))
    for nr, zb, cls in ((16, 4, 0), (32, 2, 1)):
        n = gen_bulk(a, nr, zb, off)
        ks.append((n, 32, FILE + nr * zb, [(8, "global_buffer")] * 3 + [(4, "by_value")] * 2))
    from gen_tiles import gen_tiles
    ks.append(gen_tiles(a, off))
    from gen_prune import gen_prune1
    ks.append(gen_prune1(a, off))
    from gen_tilesv import gen_tilesv
    ks.append(gen_tilesv(a, off, 32, 16))
    ks.append(gen_tilesv(a, off, 64, 32))
    from gen_normals import gen_normals
    ks.append(gen_normals(a, off))
    if len(sys.argv) > 3:
        ks.append(gen_normals(a, off, trans=sys.argv[3]))
        ks.append(gen_tiles(a, off, trans=sys.argv[3]))
        ks.append(gen_tilesv(a, off, 32, 16, trans=sys.argv[3]))
        ks.append(gen_tilesv(a, off, 64, 32, trans=sys.argv[3]))
    if len(sys.argv) > 3:
        for nr, zb in ((16, 4), (32, 2)):
            n = gen_bulk(a, nr, zb, off, trans=sys.argv[3])
            ks.append((n, 32, FILE + nr * zb + 26, [(8, "global_buffer")] * 3 + [(4, "by_value")] * 2))
    ks.append(gen_probe(a))
    if len(sys.argv) > 3:
        ks.append(gen_trans_probe(a))
    from gen_ubench import gen_ubench
    ks.append(gen_ubench(a))
    metadata(a, ks)
    open(sys.argv[2], "w").write(a.text())


if __name__ == "__main__":
    main()
