"""fh_tiles_v{NR} - the evaluate + prune step of the split 3D tile stage with the interval register
file, the choice words and the prune's register map all in VGPRs (gfx950 assembly).

Same contract as fh_tiles (gen_tiles.py), which it replaces for parents whose tape needs at most NR
registers and 16 * NCW choices (the per-slab leaf level: every parent of prospero.vm; the level below
the root with NR = 64): per FhSlot, forward interval pass of the parent tape for its 64 children in
lockstep (fidget-core/src/vm/mod.rs:325-538 as restated in dev_ops.hpp), classification
(voxel.rs:310-320), then ONE reverse sweep that prunes the tape for every ambiguous child whose trace
decided something (vm/data.rs:123-318 as restated in prune_sweep, kernels.hip): same free-register
pool (lowest free first), so the child tapes are the ones fh_tiles / k_teval3d write, bit for bit.

Why: fh_tiles keeps its register file in LDS ([reg][lane], 8 B).  Every tape op then pays two dependent
LDS round trips, the 23 KB per wave cap the occupancy at < 2 waves per SIMD, and the prune's register
map ([reg][lane] bytes, LDS as well) costs four more round trips per op: ~340 cycles per op forward,
500-1000 in the prune, at 4 % VALU issue (profiles/r01i/pmc_sq_insts.txt).  Here

  * interval register r is the VGPR pair v[FILE + 2r : FILE + 2r + 1], addressed in place with
    s_set_gpr_idx_on (M0-relative operands): no memory instruction in the forward pass at all;
  * the tape is fetched 63 ops at a time with ONE vector load, lane = op, and decoded by vector
    code (handler address, file indices of out / a / b, word 1); the scalar unit, shared by all the
    waves of a CU, only does `4 x v_readlane, s_setpc`; lane 63 (or the lane after the last op)
    holds a sentinel handler that fetches the next chunk / ends the pass: nothing is counted;
  * choice words (16 choices per word) live in NCW VGPRs, M0-relative too;
  * in the prune the dead interval file becomes the register map: one VGPR per old register holding
    the new number (or DEAD) per child, the free pool is a bit mask per child; an op whose output is
    dead in all 64 children costs 7 instructions; ops are dispatched per *class* (no operand, a, a + b,
    copy, choice reg-reg, choice reg-imm, output) through a decoded jump table;
  * no LDS at all and NR = 32 fits 128 VGPRs: 4 waves per SIMD, 16 per CU.

kernarg: as fh_tiles { FhRenderState* S; u32 level; u32 big; u32 max_regs; u32 max_choices; u32 n_waves; u32 flags;
         u32 skip_regs; u32 skip_choices }: slots whose tape exceeds (max_regs, max_choices) or fits
         (skip_regs, skip_choices) are left to another launch; flags: bit 0 probes, bit 4 both lists, bit 1 (+ bits 31:16) export of
         the choices of parents that carry links (see the kernel body); the other modes of fh_tiles do not exist here.
"""
from gen_interp import OPS, UNSUPPORTED, HSTRIDE_LOG2
import gen_tiles as GT
from gen_tiles import (Tiles, S_KERNARG, S_STATE, S_LEVEL, S_BIG, S_MAXREGS, S_MAXCH, S_ARENA, S_ARENACAP, S_SLOTS, S_NSLOTS, S_SLOT, S_OFF,
                       S_LEN, S_RC, S_ACT, S_SIGN, S_ABSM, S_DECIDED, S_CI, S_NREGS, S_NCH, S_PRUNE, S_BASE, S_MA, S_MB, S_HBASE, S_TAPE,
                       S_REM, S_K, S_W0, S_W1, S_T0, S_OUT, S_A, S_T1, S_OP, S_RET, S_T64, S_LIVE, S_ALIAS, S_CIMM, S_KEEP, S_PC, S_SAVE,
                       S_M, S_T2, S_T3, S_SKIPR, S_SKIPC, S_SI, S_NWG, V_LANE, V_L4, VX, VY, VZ, AL, AH, BL, BH, RL, RH, T, V_CW, V_C,
                       V_RESL, V_RESH, V_QNAN, V_SQRTC, V_ONE, SLOT_SIZE, SL_ACT, SL_XYZ, SL_RES, SL_COFF, SL_CLEN, SL_CRC, TRANS_UNARY, T_SMAP)

SRC0, SRC1, SRC2, DST = 1, 2, 4, 8

# ---- SGPRs beyond those shared with gen_tiles ----------------------------------------------------
S_CB = "s[48:49]"          # address of the tape chunk being fetched
S_N = "s50"                # ops of the current chunk
S_SENT = "s51"             # sentinel handler code of the current chunk
S_PBASE = "s[52:53]"       # prune: jump table of the op classes
S_JMP = "s[54:55]"         # dispatch target (s55 = s43: high half of the code address)
S_CHUNK = "s3"             # prune: chunk index
S_OUTM = "s[56:57]"        # prune: lanes of the current chunk holding OUTPUT ops
S_ANYLIVE = "s[72:73]"     # prune: old registers that are mapped (map[r] != DEAD) in at least one lane
S_CBASE = "s17"            # prune: choice ops before the current chunk
S_CWI = "s18"              # prune: index of the choice word held in V_CWP
HL = 9                     # forward handler slots of 512 bytes: the hot bodies and a copy of the dispatcher sit inline
S_M8 = "s[58:59]"          # -8 as a 64-bit integer
S_BIDX = "s60"             # prune: register b (word 1 of reg,reg ops)
S_AEQB = "s61"

# ---- VGPRs -----------------------------------------------------------------------------------------
V_L8 = "v1"                # lane * 8
DEC = ["v30", "v31", "v32", "v33"]     # forward: decoded chunk, lane = op: handler address, file index of out, of a, word 1 (b: file index)
V_DEAD = "v39"
PW = ("v40", "v41")        # raw tape chunk in flight (lane = op)
# prune (the forward pass' operands are dead by then)
V_PH, V_PW0, V_PW1 = "v30", "v31", "v32"   # decoded chunk: class jump, word 0, word 1
V_PO, V_PCI = "v42", "v43"                 # ... output register, index of the op's choice (choice ops before it)
PO = ["v4", "v5"]          # free-register pool, 32 registers per word, 1 = free
V_HIGH, V_COUNT, V_KEPT, V_NO = "v6", "v7", "v8", "v9"
V_MAV, V_MBV, V_AV, V_CWP = "v10", "v11", "v12", "v13"
V_EW0, V_EW1 = "v14", "v15"             # emitted op (aligned pair: 64-bit store)
V_U = ["v16", "v17", "v18", "v19"]
V_DST = "v[20:21]"
V_RANK, V_COFF, V_CLEN, V_CRC, V_END = "v22", "v23", "v24", "v25", "v26"
N_WORK = 48

CODE_REFILL, CODE_DONE = 52, 53
CHUNK = 63
DEADV = 0xFF


class TilesV(Tiles):
    def __init__(self, a, off, nr, ncw, trans=None):
        super().__init__(a, off, trans=trans)
        self.nr, self.ncw = nr, ncw
        self.name = f"fh_tiles_v{nr}" + ("_t" if trans else "")
        self.p = f".L{self.name}"
        self.next = f"{self.p}_next"
        self.CHF = N_WORK
        self.FILE = N_WORK + ncw
        self.n_vgpr = self.FILE + 2 * nr
        if trans:       # the routines' register window behind the register file
            self.t_base, self.t_prefix = self.n_vgpr, f"fh_ti{nr}_"
            self.n_vgpr += 26
        self.W = (nr + 31) // 32
        self.uid = 0

    def lab(self, stem):
        self.uid += 1
        return f"{self.p}_{stem}_{self.uid}"

    def F(self):
        return f"v[{self.FILE}:{self.FILE + 1}]"

    # ---- helpers -------------------------------------------------------------------------------------
    def dispatch(self):
        """fetch the next op of the decoded chunk and jump to its handler with A = file[a] loaded.  (Every handler leaves the
        index mode off, destination- or src2-relative: v_readlane only honours SRC0-relative mode, tools/probe_isa.py.)
        A copy sits at the end of every handler: one taken jump per op (21-31 cycles each, profiles/r02/ubench.json)."""
        self.a(f"""
	v_readlane_b32 s54, {DEC[0]}, {S_K}
	v_readlane_b32 {S_OUT}, {DEC[1]}, {S_K}
	v_readlane_b32 {S_A}, {DEC[2]}, {S_K}
	v_readlane_b32 {S_W1}, {DEC[3]}, {S_K}
	s_add_u32 {S_K}, {S_K}, 1
	s_set_gpr_idx_on {S_A}, {SRC0 | SRC1}
	v_pk_mov_b32 v[10:11], {self.F()}, {self.F()} op_sel:[0,1]
	s_setpc_b64 {S_JMP}""")

    def done(self):
        """file[out] = R, next op"""
        self.a(f"""
	s_set_gpr_idx_on {S_OUT}, {DST}
	v_pk_mov_b32 {self.F()}, v[14:15], v[14:15] op_sel:[0,1]""")
        self.dispatch()

    def choice_done(self):
        """file[out] = R; the choice in {V_C} (2 bits) goes straight into its word of the choice file (16 to a word, the file
        is zeroed per slot); next op"""
        self.a(f"""
	v_cmp_ne_u32_e64 {S_MA}, 3, {V_C}
	s_set_gpr_idx_on {S_OUT}, {DST}
	v_pk_mov_b32 {self.F()}, v[14:15], v[14:15] op_sel:[0,1]
	s_and_b32 {S_T0}, {S_CI}, 15
	s_lshl_b32 {S_T1}, {S_T0}, 1
	s_lshr_b32 {S_T2}, {S_CI}, 4
	s_set_gpr_idx_on {S_T2}, {SRC2 | DST}
	v_lshl_or_b32 v{self.CHF}, {V_C}, {S_T1}, v{self.CHF}
	s_add_u32 {S_CI}, {S_CI}, 1
	s_or_b64 {S_DECIDED}, {S_DECIDED}, {S_MA}""")
        self.dispatch()

    def load_b(self):
        self.a(f"""
	s_set_gpr_idx_on {S_W1}, {SRC0 | SRC1}
	v_pk_mov_b32 v[12:13], {self.F()}, {self.F()} op_sel:[0,1]
	s_set_gpr_idx_off""")

    def imm_b(self):
        self.a(f"\ts_set_gpr_idx_off\n\tv_mov_b32 {BL}, {S_W1}\n\tv_mov_b32 {BH}, {S_W1}")

    def imm_a_swap(self):
        self.a(f"""
	s_set_gpr_idx_off
	v_mov_b32 {BL}, {AL}
	v_mov_b32 {BH}, {AH}
	v_mov_b32 {AL}, {S_W1}
	v_mov_b32 {AH}, {S_W1}""")

    INLINE = {"abs", "square", "sqrt", "mul", "mulimm", "min", "max", "and", "or", "input"}

    def ool_body(self, stem, fn):
        """the body of a handler: inline (hot ops: the slot is large enough) or out of line (one more taken branch)"""
        if stem in self.INLINE:
            return fn()
        lab = f"{self.p}_b_{stem}"
        if not any(l == lab for l, _ in self.ool):
            self.ool.append((lab, fn))
        self.a(f"\ts_branch {lab}")

    # ---- forward handlers (A = file[a] already in v[10:11]; the index mode is on: SRC0 | SRC1, a) --------
    def handler(self, op):
        a = self.a
        if op == "OUTPUT":
            a(f"\ts_set_gpr_idx_off\n\tv_mov_b32 {V_RESL}, {AL}\n\tv_mov_b32 {V_RESH}, {AH}")
            return self.dispatch()
        if op == "INPUT":
            a("\ts_set_gpr_idx_off")
            return self.ool_body("input", self.h_input)
        if op == "COPY_REG":
            a(f"\ts_set_gpr_idx_on {S_OUT}, {DST}\n\tv_pk_mov_b32 {self.F()}, v[10:11], v[10:11] op_sel:[0,1]")
            return self.dispatch()
        if op == "COPY_IMM":
            a(f"\ts_set_gpr_idx_off\n\tv_mov_b32 {RL}, {S_W1}\n\tv_mov_b32 {RH}, {S_W1}")
            return self.done()
        unary = {"NEG": self.b_neg, "ABS": self.b_abs, "RECIP": self.b_recip, "SQRT": self.b_sqrt, "SQUARE": self.b_square,
                 "NOT": self.b_not}
        if op in unary:
            def body(fn=unary[op]):
                fn()
                self.done()
            a("\ts_set_gpr_idx_off")
            if op == "NEG":
                return body()
            return self.ool_body(op.lower(), body)
        if op in TRANS_UNARY:
            def body(op=op):
                self.b_trans(op)
                self.done()
            a("\ts_set_gpr_idx_off")
            return self.ool_body(op.lower(), body)
        if op == "RAND":
            def body():
                self.b_rand()
                self.done()
            a("\ts_set_gpr_idx_off")
            return self.ool_body("rand", body)
        if op in ("FLOOR", "CEIL"):
            ins = "v_floor_f32" if op == "FLOOR" else "v_ceil_f32"
            a(f"\ts_set_gpr_idx_off\n\t{ins} {RL}, {AL}\n\t{ins} {RH}, {AH}")
            return self.done()
        if op == "ROUND":
            def body():
                self.round(AL, RL)
                self.round(AH, RH)
                self.done()
            a("\ts_set_gpr_idx_off")
            return self.ool_body("round", body)
        base, form = op.rsplit("_", 1)
        if form == "RR" and base in ("ADD", "SUB"):
            # R = A (+/-) file[b] straight from the file: [al + bl, ah + bh] / [al - bh, ah - bl]
            mod = "" if base == "ADD" else " op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]"
            a(f"\ts_set_gpr_idx_on {S_W1}, {SRC1}\n\tv_pk_add_f32 v[14:15], v[10:11], {self.F()}{mod}")
            return self.done()
        if form == "RR":
            self.load_b()
        elif form == "RI":
            self.imm_b()
        else:
            self.imm_a_swap()
        if base == "MUL" and form == "RI":
            base = "MULIMM"
        bodies = {"ADD": self.b_add, "SUB": self.b_sub, "MUL": self.b_mul, "MULIMM": self.b_mul_imm, "DIV": self.b_div,
                  "COMPARE": self.b_compare, "ATAN2": self.b_atan2, "MOD": self.b_mod, "MIX": self.b_mix}
        if base in bodies:
            def body(fn=bodies[base]):
                fn()
                self.done()
            if base in ("ADD", "SUB"):
                return body()
            return self.ool_body(base.lower(), body)

        def cbody(base=base):
            if base in ("MIN", "MAX"):
                self.b_minmax(base == "MIN")
            else:
                self.b_andor(base == "AND")
            self.choice_done()
        return self.ool_body(base.lower(), cbody)

    # ---- fetch of a tape chunk: lanes 0 .. n-1 load op k of the chunk at S_CB --------------------------
    def fetch(self, n_sreg):
        self.a(f"""
	s_sub_u32 {S_T2}, 64, {n_sreg}
	s_lshr_b64 exec, -1, {S_T2}
	global_load_dwordx2 v[40:41], {V_L8}, {S_CB}
	s_mov_b64 exec, -1""")

    # ---- forward interpreter ---------------------------------------------------------------------------
    def emit_forward(self):
        a, p = self.a, self.p
        a(f"""
; ---- forward interval interpreter: {S_TAPE} = first op, {S_LEN} = ops; returns to {S_RET} ------------------
{p}_run:
	s_mov_b32 s55, s43
	s_mov_b64 {S_CB}, {S_TAPE}
	s_mov_b32 {S_REM}, {S_LEN}
	s_min_u32 {S_N}, {S_REM}, {CHUNK}""")
        self.fetch(S_N)
        a(f"""
{p}_refill:
	; ---- decode the chunk that has arrived (lane = op), ask for the one after it -------------------------
	s_set_gpr_idx_off
	s_min_u32 {S_N}, {S_REM}, {CHUNK}
	s_sub_u32 {S_REM}, {S_REM}, {S_N}
	s_cmp_eq_u32 {S_REM}, 0
	s_cselect_b32 {S_SENT}, {CODE_DONE}, {CODE_REFILL}
	s_waitcnt vmcnt(0)
	v_and_b32 {T[0]}, 0xff, {PW[0]}
	v_bfe_u32 {T[1]}, {PW[0]}, 8, 12
	v_lshrrev_b32 {T[2]}, 20, {PW[0]}
	v_cmp_gt_u32 vcc, {S_N}, {V_LANE}
	v_mov_b32 {T[3]}, {S_SENT}
	v_lshlrev_b32 {T[4]}, 1, {PW[1]}
	v_cndmask_b32 {T[1]}, 0, {T[1]}, vcc            ; (sentinel lanes: register 0)
	v_cndmask_b32 {T[2]}, 0, {T[2]}, vcc
	v_cndmask_b32 {T[0]}, {T[3]}, {T[0]}, vcc
	v_lshlrev_b32 {DEC[1]}, 1, {T[1]}
	v_lshlrev_b32 {DEC[2]}, 1, {T[2]}
	v_subrev_u32 {T[3]}, 22, {T[0]}
	v_cmp_gt_u32 vcc, 12, {T[3]}
	v_lshlrev_b32 {T[0]}, {HL}, {T[0]}
	v_add_u32 {DEC[0]}, s42, {T[0]}
	v_cndmask_b32 {DEC[3]}, {PW[1]}, {T[4]}, vcc
	s_cmp_eq_u32 {S_REM}, 0
	s_cbranch_scc1 {p}_nofetch
	s_add_u32 s48, s48, {CHUNK * 8}
	s_addc_u32 s49, s49, 0
	s_min_u32 {S_T3}, {S_REM}, {CHUNK}""")
        self.fetch(S_T3)
        a(f"""
{p}_nofetch:
	s_mov_b32 {S_K}, 0
{self.next}:""")
        self.dispatch()
        a(f"""
{p}_done:
	s_set_gpr_idx_off
	s_setpc_b64 {S_RET}
	.p2align {HL}
{p}_handlers:""")
        for i in range(64):
            a(f"\t.p2align {HL}")
            a(f"{p}_h{i}:")
            if i == CODE_REFILL:
                a(f"\ts_branch {p}_refill")
            elif i == CODE_DONE:
                a(f"\ts_branch {p}_done")
            elif i >= len(OPS):
                self.dispatch()
            else:
                op = OPS[i]
                base = op.rsplit("_", 1)[0] if "_" in op and op not in ("COPY_REG", "COPY_IMM") else op
                if base in UNSUPPORTED and not self.trans:
                    self.dispatch()
                else:
                    self.handler(op)
            a(f"\t.if (. - {p}_h{i}) > {1 << HL}\n\t.error \"tile handler {i} of {self.name} exceeds its slot\"\n\t.endif")
        a(f"\t.p2align {HL}")
        for lab, fn in self.ool:
            a(f"{lab}:")
            fn()

    # ---- prune: pool of free registers (W words of 32, 1 = free), lowest first --------------------------------
    def pool_take(self, out):
        """out = lowest free register, marked used; for the lanes in exec"""
        a, u = self.a, V_U
        if self.W == 1:
            a(f"""
	v_ffbl_b32 {out}, {PO[0]}
	v_lshlrev_b32 {u[0]}, {out}, {V_ONE}
	v_add_u32 {u[1]}, 1, {out}
	v_xor_b32 {PO[0]}, {PO[0]}, {u[0]}
	v_max_u32 {V_HIGH}, {V_HIGH}, {u[1]}""")
            return
        a(f"""
	v_ffbl_b32 {u[0]}, {PO[0]}
	v_ffbl_b32 {u[1]}, {PO[1]}
	v_or_b32 {u[1]}, 32, {u[1]}
	v_min_u32 {out}, {u[0]}, {u[1]}
	v_lshlrev_b32 {u[0]}, {out}, {V_ONE}
	v_cmp_gt_u32_e64 {S_M[0]}, 32, {out}
	v_add_u32 {u[1]}, 1, {out}
	v_max_u32 {V_HIGH}, {V_HIGH}, {u[1]}
	v_cndmask_b32_e64 {u[2]}, 0, {u[0]}, {S_M[0]}
	v_cndmask_b32_e64 {u[3]}, {u[0]}, 0, {S_M[0]}
	v_xor_b32 {PO[0]}, {PO[0]}, {u[2]}
	v_xor_b32 {PO[1]}, {PO[1]}, {u[3]}""")

    def pool_give(self, reg):
        a, u = self.a, V_U
        if self.W == 1:
            a(f"\tv_lshlrev_b32 {u[0]}, {reg}, {V_ONE}\n\tv_or_b32 {PO[0]}, {PO[0]}, {u[0]}")
            return
        a(f"""
	v_lshlrev_b32 {u[0]}, {reg}, {V_ONE}
	v_cmp_gt_u32_e64 {S_M[0]}, 32, {reg}
	s_nop 1
	v_cndmask_b32_e64 {u[2]}, 0, {u[0]}, {S_M[0]}
	v_cndmask_b32_e64 {u[3]}, {u[0]}, 0, {S_M[0]}
	v_or_b32 {PO[0]}, {PO[0]}, {u[2]}
	v_or_b32 {PO[1]}, {PO[1]}, {u[3]}""")

    def map_read(self, dst, sreg):
        """dst = map[sreg] (leaves the index mode off)"""
        self.a(f"\ts_set_gpr_idx_on {sreg}, {SRC0}\n\tv_mov_b32 {dst}, v{self.FILE}\n\ts_set_gpr_idx_off")

    def map_write(self, sreg, src, live=True):
        """map[sreg] = src for the lanes in exec (non-empty).  live: src is a register number (some lane now maps sreg);
        otherwise DEAD written to every lane that mapped it (nobody maps it any more)"""
        bit = "s_bitset1_b64" if live else "s_bitset0_b64"
        self.a(f"\ts_set_gpr_idx_on {sreg}, {DST}\n\tv_mov_b32 v{self.FILE}, {src}\n\ts_set_gpr_idx_off\n\t{bit} {S_ANYLIVE}, {sreg}")

    def alloc_where_dead(self, val, sreg, within):
        """lanes of `within` (= exec) whose map value `val` is DEAD take a fresh register (written back to map[sreg])"""
        a = self.a
        lab = self.lab("alloc")
        a(f"""
	v_cmp_eq_u32 vcc, {V_DEAD}, {val}
	s_and_b64 vcc, vcc, {within}
	s_cbranch_scc0 {lab}
	s_mov_b64 exec, vcc""")
        self.pool_take(val)
        self.map_write(sreg, val)
        a(f"""
	s_mov_b64 exec, {within}
{lab}:""")

    def emit_op(self):
        """store {V_EW0, V_EW1} one op below dst, for the lanes in exec"""
        self.a(f"""
	v_lshl_add_u64 {V_DST}, {V_DST}, 0, {S_M8}
	v_add_u32 {V_COUNT}, 1, {V_COUNT}
	global_store_dwordx2 {V_DST}, v[14:15], off""")

    def head(self, has_choice):
        """common head of the op classes (exec = all lanes): the choice (if any), no = map[out]; the lanes where it is live
        -> exec (never empty: only ops whose output some lane wants are visited), map[out] = DEAD there"""
        a, p = self.a, self.p
        if has_choice:
            have = self.lab("cw_have")
            a(f"""
	v_readlane_b32 {S_CI}, {V_PCI}, {S_K}
	s_nop 0
	s_lshr_b32 {S_T1}, {S_CI}, 4
	s_and_b32 {S_T0}, {S_CI}, 15
	s_cmp_eq_u32 {S_T1}, {S_CWI}
	s_cbranch_scc1 {have}
	s_mov_b32 {S_CWI}, {S_T1}
	s_set_gpr_idx_on {S_T1}, {SRC0}
	v_mov_b32 {V_CWP}, v{self.CHF}
	s_set_gpr_idx_off
{have}:
	s_lshl_b32 {S_T0}, {S_T0}, 1
	v_bfe_u32 {V_C}, {V_CWP}, {S_T0}, 2""")
        self.map_read(V_NO, S_OUT)
        a(f"""
	v_cmp_ne_u32_e64 {S_LIVE}, {V_DEAD}, {V_NO}
	s_nop 0
	s_mov_b64 exec, {S_LIVE}""")
        self.map_write(S_OUT, V_DEAD, live=False)

    def ew0(self, opcode, a_reg=None):
        """EW0 = opcode | no << 8 [| a_reg << 20]"""
        a = self.a
        a(f"\tv_lshl_or_b32 {V_EW0}, {V_NO}, 8, {opcode}")
        if a_reg:
            a(f"\tv_lshl_or_b32 {V_EW0}, {a_reg}, 20, {V_EW0}")

    def keep_path(self, within, rr, choice):
        """the op survives in the lanes `within` (exec = within): free its register, allocate its operands, emit"""
        a = self.a
        self.pool_give(V_NO)
        self.map_read(V_MAV, S_A)
        self.alloc_where_dead(V_MAV, S_A, within)
        if rr:
            same, bdone = self.lab("b_same"), self.lab("b_done")
            a(f"""
	s_cmp_eq_u32 {S_A}, {S_W1}
	s_cbranch_scc1 {same}""")
            self.map_read(V_MBV, S_W1)
            self.alloc_where_dead(V_MBV, S_W1, within)
            a(f"""
	v_mov_b32 {V_EW1}, {V_MBV}
	s_branch {bdone}
{same}:
	v_mov_b32 {V_EW1}, {V_MAV}
{bdone}:""")
        else:
            a(f"\tv_mov_b32 {V_EW1}, {S_W1}")
        if choice:
            a(f"\tv_add_u32 {V_KEPT}, 1, {V_KEPT}")
        self.ew0(S_OP, V_MAV)
        self.emit_op()

    def alias_path(self, mask_a, mask_b):
        """Decided choices / copies: `out` is its surviving operand.  Lanes `mask_a` alias register a, lanes `mask_b` (or None)
        register b; {S_ALIAS} = their union, non-empty.  An operand that is not live yet takes the register over (no op);
        one that is gets a COPY_REG."""
        a = self.a
        done = self.lab("alias_done")
        a(f"\ts_mov_b64 exec, {S_ALIAS}")
        self.map_read(V_AV, S_A)
        if mask_b:
            self.map_read(V_MBV, S_W1)
            a(f"\tv_cndmask_b32_e64 {V_AV}, {V_AV}, {V_MBV}, {mask_b}")
        a(f"""
	v_cmp_eq_u32 vcc, {V_DEAD}, {V_AV}
	s_and_b64 {S_KEEP}, vcc, {S_ALIAS}
	s_andn2_b64 {S_CIMM}, {S_ALIAS}, vcc""")
        # hand-over lanes ({S_KEEP}), by operand
        skip_a = self.lab("ho_a")
        a(f"""
	s_and_b64 exec, {S_KEEP}, {mask_a}
	s_cbranch_scc0 {skip_a}""")
        self.map_write(S_A, V_NO)
        a(f"{skip_a}:")
        if mask_b:
            skip_b = self.lab("ho_b")
            a(f"""
	s_and_b64 exec, {S_KEEP}, {mask_b}
	s_cbranch_scc0 {skip_b}""")
            self.map_write(S_W1, V_NO)
            a(f"{skip_b}:")
        a(f"""
	s_cmp_eq_u64 {S_CIMM}, 0
	s_cbranch_scc1 {done}
	s_mov_b64 exec, {S_CIMM}""")
        self.pool_give(V_NO)
        self.ew0(2, V_AV)
        a(f"\tv_mov_b32 {V_EW1}, 0")
        self.emit_op()
        a(f"{done}:")

    def pnext(self):
        """Next op of the sweep: the highest op below {S_K} in the current chunk whose output register is mapped in some lane
        (or that is an OUTPUT).  Everything in between is dead in every lane and is never looked at: their outputs are not
        wanted now, and nothing is visited in between that could make them wanted."""
        a, p = self.a, self.p
        a(f"""
	s_mov_b64 exec, -1
	v_lshrrev_b64 v[16:17], {V_PO}, {S_ANYLIVE}
	s_bfm_b64 {S_SAVE}, {S_K}, 0
	v_and_b32 v16, 1, v16
	v_cmp_ne_u32_e64 {S_T64}, 0, v16
	s_or_b64 {S_T64}, {S_T64}, {S_OUTM}
	s_and_b64 {S_T64}, {S_T64}, {S_SAVE}
	s_cbranch_scc0 {p}_pchunk
	s_flbit_i32_b64 {S_K}, {S_T64}
	s_sub_u32 {S_K}, 63, {S_K}
	v_readlane_b32 {S_W0}, {V_PW0}, {S_K}
	v_readlane_b32 {S_W1}, {V_PW1}, {S_K}
	v_readlane_b32 s54, {V_PH}, {S_K}
	s_and_b32 {S_OP}, {S_W0}, 0xff
	s_bfe_u32 {S_OUT}, {S_W0}, 0xc0008
	s_lshr_b32 {S_A}, {S_W0}, 20
	s_setpc_b64 {S_JMP}""")

    # ---- prune sweep -------------------------------------------------------------------------------------------
    def emit_prune(self):
        a, p = self.a, self.p
        a(f"""
; ---- prune sweep: ops {S_LEN}-1 .. 0 of the tape at {S_TAPE} for the lanes {S_PRUNE}; returns to {S_RET} --------------
{p}_prune:
	s_mov_b32 s55, s53
	v_mov_b32 {V_COUNT}, 0
	v_mov_b32 {V_KEPT}, 0
	v_mov_b32 {V_HIGH}, 0
	v_mov_b32 {V_CWP}, 0
	s_mov_b32 {S_CWI}, -1
	s_mov_b64 {S_ANYLIVE}, 0
	s_mov_b32 {S_CBASE}, {S_NCH}""")
        for w in range(self.W):
            a(f"\tv_mov_b32 {PO[w]}, -1")
        for r in range(self.nr):          # the (dead) interval file becomes the register map
            a(f"\tv_mov_b32 v{self.FILE + r}, {V_DEAD}")
        off = lambda k: f"{p}_k_{k} - {p}_classes"
        a(f"""
	; last chunk first
	s_add_u32 {S_T0}, {S_LEN}, {CHUNK - 1}
	; chunks = ceil(len / {CHUNK}) by multiplication: x * 66577 >> 22 == x / 63 for x < 64512
	s_mul_i32 {S_CHUNK}, {S_T0}, 66577
	s_lshr_b32 {S_CHUNK}, {S_CHUNK}, 22
	s_sub_u32 {S_CHUNK}, {S_CHUNK}, 1
	s_mul_i32 {S_T0}, {S_CHUNK}, {CHUNK * 8}
	s_add_u32 s48, s44, {S_T0}
	s_addc_u32 s49, s45, 0
	s_mul_i32 {S_T1}, {S_CHUNK}, {CHUNK}
	s_sub_u32 {S_N}, {S_LEN}, {S_T1}""")
        self.fetch(S_N)
        a(f"""
{p}_prefill:
	; ---- decode the chunk that has arrived (lane = op): class handler, output register, choice index; ask for the chunk below
	s_waitcnt vmcnt(0)
	v_and_b32 {T[0]}, 0xff, {PW[0]}
	v_mov_b32 {V_PW0}, {PW[0]}
	v_mov_b32 {V_PW1}, {PW[1]}
	v_bfe_u32 {V_PO}, {PW[0]}, 8, 12
	v_cmp_gt_u32_e64 {S_SAVE}, {S_N}, {V_LANE}            ; lanes holding an op
	v_cmp_eq_u32_e64 {S_M[0]}, 0, {T[0]}                  ; OUTPUT
	v_mov_b32 {T[1]}, {off('a')}                          ; default class: one register operand
	v_mov_b32 v27, {off('out')}
	v_cmp_eq_u32_e64 {S_M[1]}, 2, {T[0]}                  ; COPY_REG
	v_and_b32 {T[2]}, 0xfd, {T[0]}
	v_cndmask_b32_e64 {T[1]}, {T[1]}, v27, {S_M[0]}
	s_and_b64 {S_OUTM}, {S_M[0]}, {S_SAVE}
	v_mov_b32 v27, {off('copy')}
	v_cmp_eq_u32_e64 {S_M[2]}, 1, {T[2]}                  ; INPUT (1), COPY_IMM (3)
	v_subrev_u32 {T[3]}, 22, {T[0]}
	v_cndmask_b32_e64 {T[1]}, {T[1]}, v27, {S_M[1]}
	v_mov_b32 v27, {off('none')}
	v_cmp_gt_u32_e64 {S_M[0]}, 8, {T[3]}                  ; 22 .. 29: reg,reg, no choice
	v_subrev_u32 v28, 30, {T[0]}
	v_cndmask_b32_e64 {T[1]}, {T[1]}, v27, {S_M[2]}
	v_mov_b32 v27, {off('rr')}
	v_cmp_gt_u32_e64 {S_M[1]}, 4, v28                     ; 30 .. 33: choice reg,reg
	v_subrev_u32 v28, 42, {T[0]}
	v_cndmask_b32_e64 {T[1]}, {T[1]}, v27, {S_M[0]}
	v_mov_b32 v27, {off('crr')}
	v_cmp_gt_u32_e64 {S_M[2]}, 4, v28                     ; 42 .. 45: choice reg,imm
	v_mov_b32 v28, {off('cri')}
	v_cndmask_b32_e64 {T[1]}, {T[1]}, v27, {S_M[1]}
	s_or_b64 {S_T64}, {S_M[1]}, {S_M[2]}                  ; choice ops of the chunk
	s_and_b64 {S_T64}, {S_T64}, {S_SAVE}
	v_cndmask_b32_e64 {T[1]}, {T[1]}, v28, {S_M[2]}
	v_add_u32 {V_PH}, s52, {T[1]}
	; choice index of an op = choice ops before it in the tape: those below this chunk + those below it in the chunk
	s_bcnt1_i32_b64 {S_T0}, {S_T64}
	s_sub_u32 {S_CBASE}, {S_CBASE}, {S_T0}
	v_mbcnt_lo_u32_b32 {V_PCI}, s76, 0
	v_mbcnt_hi_u32_b32 {V_PCI}, s77, {V_PCI}
	v_add_u32 {V_PCI}, {S_CBASE}, {V_PCI}
	s_mov_b32 {S_K}, {S_N}
	s_cmp_eq_u32 {S_CHUNK}, 0
	s_cbranch_scc1 {p}_pnext
	s_sub_u32 s48, s48, {CHUNK * 8}
	s_subb_u32 s49, s49, 0
	s_mov_b32 {S_T3}, {CHUNK}""")
        self.fetch(S_T3)
        a(f"{p}_pnext:")
        self.pnext()
        a(f"""
{p}_pchunk:
	s_sub_u32 {S_CHUNK}, {S_CHUNK}, 1
	s_cbranch_scc1 {p}_pdone
	s_mov_b32 {S_N}, {CHUNK}
	s_branch {p}_prefill
{p}_pdone:
	s_waitcnt vmcnt(0)
	s_setpc_b64 {S_RET}
	.p2align 6
{p}_classes:
; ---- class: OUTPUT.  every lane being pruned needs its operand -----------------------------------------------------
{p}_k_out:
	s_mov_b64 exec, {S_PRUNE}""")
        self.map_read(V_MAV, S_A)
        self.alloc_where_dead(V_MAV, S_A, S_PRUNE)
        a(f"""
	v_lshlrev_b32 {V_EW0}, 20, {V_MAV}
	v_mov_b32 {V_EW1}, {S_W1}""")
        self.emit_op()
        self.pnext()
        a(f"; ---- class: no register operand (INPUT, COPY_IMM) ------------------------------------------------------------------\n{p}_k_none:")
        self.head(False)
        self.pool_give(V_NO)
        self.ew0(S_OP)
        a(f"\tv_mov_b32 {V_EW1}, {S_W1}")
        self.emit_op()
        self.pnext()
        a(f"; ---- class: one register operand (unary, reg (op) imm, imm (op) reg) ----------------------------------------------\n{p}_k_a:")
        self.head(False)
        self.keep_path(S_LIVE, False, False)
        self.pnext()
        a(f"; ---- class: two register operands ------------------------------------------------------------------------------------\n{p}_k_rr:")
        self.head(False)
        self.keep_path(S_LIVE, True, False)
        self.pnext()
        a(f"; ---- class: COPY_REG: out is a ------------------------------------------------------------------------------------------\n{p}_k_copy:")
        self.head(False)
        a(f"\ts_mov_b64 {S_ALIAS}, {S_LIVE}")
        self.alias_path(S_LIVE, None)
        self.pnext()
        a(f"; ---- class: min / max / and / or, reg,reg ---------------------------------------------------------------------------------\n{p}_k_crr:")
        self.head(True)
        nokeep, noalias = self.lab("crr_nokeep"), self.lab("crr_noalias")
        a(f"""
	v_cmp_eq_u32_e64 {S_MA}, 1, {V_C}
	v_cmp_eq_u32_e64 {S_MB}, 2, {V_C}
	s_and_b64 {S_MA}, {S_MA}, {S_LIVE}
	s_and_b64 {S_MB}, {S_MB}, {S_LIVE}
	s_or_b64 {S_ALIAS}, {S_MA}, {S_MB}
	s_andn2_b64 {S_SAVE}, {S_LIVE}, {S_ALIAS}
	s_cmp_eq_u64 {S_ALIAS}, 0
	s_cbranch_scc1 {noalias}""")
        self.alias_path(S_MA, S_MB)
        a(f"""
{noalias}:
	s_cmp_eq_u64 {S_SAVE}, 0
	s_cbranch_scc1 {p}_pnext
	s_mov_b64 exec, {S_SAVE}""")
        self.keep_path(S_SAVE, True, True)
        self.pnext()
        a(f"; ---- class: min / max / and / or, reg,imm: Right = the immediate --------------------------------------------------------\n{p}_k_cri:")
        self.head(True)
        noalias, nocimm = self.lab("cri_noalias"), self.lab("cri_nocimm")
        a(f"""
	v_cmp_eq_u32_e64 {S_MA}, 1, {V_C}
	v_cmp_eq_u32_e64 {S_MB}, 2, {V_C}
	s_and_b64 {S_MA}, {S_MA}, {S_LIVE}
	s_and_b64 {S_MB}, {S_MB}, {S_LIVE}
	s_or_b64 {S_T64}, {S_MA}, {S_MB}
	s_andn2_b64 {S_SAVE}, {S_LIVE}, {S_T64}
	s_mov_b64 {S_ALIAS}, {S_MA}
	s_cmp_eq_u64 {S_ALIAS}, 0
	s_cbranch_scc1 {noalias}""")
        self.alias_path(S_MA, None)
        a(f"""
{noalias}:
	s_cmp_eq_u64 {S_MB}, 0
	s_cbranch_scc1 {nocimm}
	s_mov_b64 exec, {S_MB}""")
        self.pool_give(V_NO)
        self.ew0(3)
        a(f"\tv_mov_b32 {V_EW1}, {S_W1}")
        self.emit_op()
        a(f"""
{nocimm}:
	s_cmp_eq_u64 {S_SAVE}, 0
	s_cbranch_scc1 {p}_pnext
	s_mov_b64 exec, {S_SAVE}""")
        self.keep_path(S_SAVE, False, True)
        self.pnext()

    # ---- kernel ----------------------------------------------------------------------------------------------
    def emit_kernel(self):
        a, o, p, name = self.a, self.off, self.p, self.name
        a(f"""
	.text
	.protected {name}
	.globl {name}
	.p2align 8
	.type {name},@function
{name}:
	s_load_dwordx2 {S_STATE}, {S_KERNARG}, 0x0
	s_load_dwordx4 s[8:11], {S_KERNARG}, 0x8
	s_load_dwordx2 s[100:101], {S_KERNARG}, 0x18
	v_mov_b32 {V_QNAN}, 0x7fc00000
	v_mov_b32 {V_SQRTC}, 0xf800000
	v_mov_b32 {V_ONE}, 1
	v_mov_b32 {V_DEAD}, {DEADV}
	s_mov_b32 {S_SIGN}, 0x80000000
	s_mov_b32 {S_ABSM}, 0x7fffffff
	s_mov_b32 s58, -8
	s_mov_b32 s59, -1
	v_lshlrev_b32 {V_L8}, 3, {V_LANE}
	v_lshlrev_b32 {V_L4}, 2, {V_LANE}
	s_waitcnt lgkmcnt(0)
	; flags bits 9:8: this launch's waves raise their issue priority (s_setprio 3 / 1).  A level of the coarse chain is a few hundred
	; one-wave parents, each a dependent chain that sets the level's time; beside the other streams' kernels of a pipelined frame
	; their SIMDs are shared, and the arbiter otherwise hands a chain wave its turn round robin with thousands of throughput waves
	s_bitcmp1_b32 s101, 9
	s_cbranch_scc0 {p}_prio_lo
	s_setprio 3
	s_branch {p}_prio_done
{p}_prio_lo:
	s_bitcmp1_b32 s101, 8
	s_cbranch_scc0 {p}_prio_done
	s_setprio 1
{p}_prio_done:
	s_mov_b32 {S_LEVEL}, s8
	s_mov_b32 {S_BIG}, s9
	s_mov_b32 {S_MAXCH}, s11
	s_mov_b32 {S_MAXREGS}, s10
	s_min_u32 {S_MAXREGS}, {S_MAXREGS}, {self.nr}
	s_min_u32 {S_MAXCH}, {S_MAXCH}, {self.ncw * 16}
	; flags bit 4: both slot lists in this launch, `big` first then the other (a level of few parents, where two launches
	; one after the other cost more than the waves: every wave walks its share of one list, then of the other)
	; (s70, s86, s87 are the kernel's free SGPRs; {S_BIG}'s and {S_LEVEL}'s registers are reused once a list's state is loaded:
	; s70 = other list still to do | big << 1 | level << 8, s86 = this wave's index)
	s_mov_b32 s86, s2
	s_bfe_u32 s70, s101, 0x10004
	s_lshl_b32 s87, {S_BIG}, 1
	s_or_b32 s70, s70, s87
	s_lshl_b32 s87, {S_LEVEL}, 8
	s_or_b32 s70, s70, s87
{p}_list:
	; per-list state: slots[big], slot_cap[big], n_slots[big][level]
	s_lshl_b32 {S_T0}, {S_BIG}, 3
	s_add_u32 s76, s4, {S_T0}
	s_addc_u32 s77, s5, 0
	s_load_dwordx2 {S_SLOTS}, {S_T64}, {o['slots']}
	s_lshl_b32 {S_T0}, {S_BIG}, 2
	s_add_u32 s76, s4, {S_T0}
	s_addc_u32 s77, s5, 0
	s_load_dword {S_T2}, {S_T64}, {o['slot_cap']}
	s_lshl_b32 {S_T0}, {S_BIG}, 3
	s_add_u32 {S_T0}, {S_T0}, {S_LEVEL}
	s_lshl_b32 {S_T0}, {S_T0}, 2
	s_add_u32 s76, s4, {S_T0}
	s_addc_u32 s77, s5, 0
	s_load_dword {S_NSLOTS}, {S_T64}, {o['n_slots']}
	s_load_dwordx2 {S_ARENA}, {S_STATE}, {o['arena']}
	s_load_dword {S_ARENACAP}, {S_STATE}, {o['arena_cap']}
	s_getpc_b64 {S_HBASE}
{p}_pc0:
	s_add_u32 s52, s42, {p}_classes - {p}_pc0
	s_addc_u32 s53, s43, 0
	s_add_u32 s42, s42, {p}_handlers - {p}_pc0
	s_addc_u32 s43, s43, 0
	s_mov_b32 s55, s43
	s_waitcnt lgkmcnt(0)
	s_min_u32 {S_NSLOTS}, {S_NSLOTS}, {S_T2}
	s_load_dwordx2 s[76:77], {S_KERNARG}, 0x20
	s_waitcnt lgkmcnt(0)
	s_mov_b32 {S_SKIPR}, s76
	s_mov_b32 {S_SKIPC}, s77
{p}_outer:
	; ---- next slot: round robin over the waves -------------------------------------------------------------------
	s_cmp_ge_u32 {S_SI}, {S_NSLOTS}
	s_cbranch_scc1 {p}_exit
	s_mov_b32 {S_T0}, {S_SI}
	s_add_u32 {S_SI}, {S_SI}, {S_NWG}
	s_mul_i32 {S_T1}, {S_T0}, {SLOT_SIZE}
	s_mul_hi_u32 {S_T0}, {S_T0}, {SLOT_SIZE}
	s_add_u32 s20, s14, {S_T1}
	s_addc_u32 s21, s15, {S_T0}
	s_load_dwordx4 s[24:27], {S_SLOT}, 0x0
	s_load_dwordx2 {S_ACT}, {S_SLOT}, {SL_ACT}
	s_waitcnt lgkmcnt(0)
	s_cmp_eq_u64 {S_ACT}, 0
	s_cbranch_scc1 {p}_outer
	s_and_b32 {S_NREGS}, {S_RC}, 0xffff
	s_lshr_b32 {S_NCH}, {S_RC}, 16
	; is this slot for this launch?
	s_cmp_gt_u32 {S_NREGS}, {S_MAXREGS}
	s_cbranch_scc1 {p}_outer
	s_cmp_gt_u32 {S_NCH}, {S_MAXCH}
	s_cbranch_scc1 {p}_outer
	s_cmp_le_u32 {S_NREGS}, {S_SKIPR}
	s_cselect_b32 {S_T0}, 1, 0
	s_cmp_le_u32 {S_NCH}, {S_SKIPC}
	s_cselect_b32 {S_T1}, 1, 0
	s_and_b32 {S_T0}, {S_T0}, {S_T1}
	s_cmp_eq_u32 {S_T0}, 1
	s_cbranch_scc1 {p}_outer
	s_memtime s[60:61]""")
        for k, r in enumerate((VX[0], VX[1], VY[0], VY[1], VZ[0], VZ[1])):
            a(f"\tglobal_load_dword {r}, {V_L4}, {S_SLOT} offset:{SL_XYZ + 256 * k}")
        a(f"""
	s_mov_b32 s44, {S_OFF}
	s_mov_b32 s45, 0
	s_lshl_b64 {S_TAPE}, {S_TAPE}, 3
	s_add_u32 s44, s44, s10
	s_addc_u32 s45, s45, s11
	s_mov_b32 {S_CI}, 0
	s_mov_b64 {S_DECIDED}, 0
""" + "".join(f"\tv_mov_b32 v{self.CHF + w}, 0\n" for w in range(self.ncw)) + f"""	v_mov_b32 {V_RESL}, {V_QNAN}
	v_mov_b32 {V_RESH}, {V_QNAN}
	s_getpc_b64 {S_RET}
{p}_pc1:
	s_add_u32 s74, s74, {p}_ret1 - {p}_pc1
	s_addc_u32 s75, s75, 0
	s_branch {p}_run
{p}_ret1:
	s_memtime s[62:63]
	global_store_dword {V_L4}, {V_RESL}, {S_SLOT} offset:{SL_RES}
	global_store_dword {V_L4}, {V_RESH}, {S_SLOT} offset:{SL_RES + 256}
	; ---- classify: ambiguous = act && !(hi < 0) && !(lo > 0); prune those whose trace decided something ------------
	v_cmp_gt_f32_e64 {S_M[0]}, 0, {V_RESH}
	v_cmp_gt_f32_e64 {S_M[1]}, {V_RESL}, 0
	v_mov_b32 {V_COFF}, {S_OFF}
	v_mov_b32 {V_CLEN}, {S_LEN}
	v_mov_b32 {V_CRC}, {S_RC}
	s_or_b64 {S_M[0]}, {S_M[0]}, {S_M[1]}
	s_andn2_b64 {S_PRUNE}, {S_ACT}, {S_M[0]}
	s_and_b64 {S_PRUNE}, {S_PRUNE}, {S_DECIDED}
	s_cmp_eq_u64 {S_PRUNE}, 0
	s_cbranch_scc1 {p}_store
	; ---- arena: every pruned child reserves a slot as long as the parent tape -------------------------------------------
	s_bcnt1_i32_b64 {S_T0}, {S_PRUNE}
	s_mul_i32 {S_T1}, {S_T0}, {S_LEN}
	v_mbcnt_lo_u32_b32 {V_RANK}, s34, 0
	v_mbcnt_hi_u32_b32 {V_RANK}, s35, {V_RANK}
	v_cmp_eq_u32 vcc, 0, {V_LANE}
	s_and_saveexec_b64 {S_SAVE}, vcc
	v_mov_b32 {T[0]}, {S_T1}
	v_mov_b32 {T[1]}, 0
	global_atomic_add {T[2]}, {T[1]}, {T[0]}, {S_STATE} offset:{o['arena_head']} sc0
	s_waitcnt vmcnt(0)
	s_mov_b64 exec, {S_SAVE}
	s_nop 0
	v_readfirstlane_b32 {S_BASE}, {T[2]}
	s_nop 3
	s_add_u32 {S_T0}, {S_BASE}, {S_T1}
	s_cbranch_scc1 {p}_overflow
	s_cmp_le_u32 {S_T0}, {S_ARENACAP}
	s_cbranch_scc0 {p}_overflow
	; dst = arena + 8 * (base + (rank + 1) * len): one past the last op of this lane's slot
	v_add_u32 {T[0]}, 1, {V_RANK}
	v_mul_lo_u32 {T[0]}, {T[0]}, {S_LEN}
	v_add_u32 {V_END}, {S_BASE}, {T[0]}
	; flags bit 1 (level 1 behind the linked prune of level 0, prune2.hip): a parent whose tape carries this frame's links - the word
	; in front of [choice table | links | tape] is frame_stamp << 32 | len | choices << 16 - is not pruned here: its choice words go
	; to chw[list] (16 words per slot in list 0, flags[31:16] in list 1), its ambiguous children are marked (c_len = ~0, c_off = the
	; end of their arena slot) and k_prune2, launched behind this kernel, writes their tapes - one wave per child instead of the
	; lockstep sweep over the parent's ops below.
	s_bitcmp1_b32 s101, 1
	s_cbranch_scc0 {p}_sweep
	s_add_u32 {S_T0}, {S_LEN}, {S_NCH}
	s_add_u32 {S_T0}, {S_T0}, 1
	s_sub_u32 s76, {S_OFF}, {S_T0}
	s_cbranch_scc1 {p}_sweep                          ; (the tape starts too low in the arena to have anything in front)
	s_mov_b32 s77, 0
	s_lshl_b64 s[76:77], s[76:77], 3
	s_add_u32 s76, s76, s10
	s_addc_u32 s77, s77, s11
	s_load_dwordx2 s[76:77], s[76:77], 0x0
	s_load_dword {S_T2}, {S_STATE}, {o['frame_stamp']}
	s_lshl_b32 {S_T1}, {S_NCH}, 16
	s_or_b32 {S_T1}, {S_T1}, {S_LEN}
	s_waitcnt lgkmcnt(0)
	s_cmp_eq_u32 s76, {S_T1}
	s_cbranch_scc0 {p}_sweep
	s_cmp_eq_u32 s77, {S_T2}
	s_cbranch_scc0 {p}_sweep
	; the choice words: chw[list] + slot index * stride
	s_bfe_u32 {S_T0}, s70, 0x10001                    ; which list
	s_lshl_b32 {S_T1}, {S_T0}, 3
	s_add_u32 s76, s4, {S_T1}
	s_addc_u32 s77, s5, 0
	s_load_dwordx2 s[76:77], s[76:77], {o['chw']}
	s_lshr_b32 {S_T2}, s101, 16
	s_cmp_eq_u32 {S_T0}, 0
	s_cselect_b32 {S_T2}, 16, {S_T2}
	s_lshl_b32 {S_T2}, {S_T2}, 8                      ; bytes per slot
	s_sub_u32 {S_T3}, {S_SI}, {S_NWG}                 ; this slot's index (the round robin has moved on)
	s_mul_hi_u32 {S_T1}, {S_T3}, {S_T2}
	s_mul_i32 {S_T3}, {S_T3}, {S_T2}
	s_waitcnt lgkmcnt(0)
	s_add_u32 s76, s76, {S_T3}
	s_addc_u32 s77, s77, {S_T1}
""" + "".join(("\ts_add_u32 s76, s76, 0x1000\n\ts_addc_u32 s77, s77, 0\n" if w and w % 16 == 0 else "") +
              f"\ts_cmp_le_u32 {S_NCH}, {16 * w}\n\ts_cbranch_scc1 {p}_exported\n\tglobal_store_dword {V_L4}, v{self.CHF + w}, s[76:77] offset:{(w % 16) * 256}\n"
              for w in range(self.ncw)) + f"""{p}_exported:
	v_mov_b32 {T[1]}, -1
	v_cndmask_b32_e64 {V_COFF}, {V_COFF}, {V_END}, {S_PRUNE}
	v_cndmask_b32_e64 {V_CLEN}, {V_CLEN}, {T[1]}, {S_PRUNE}
	s_branch {p}_store
{p}_sweep:
	v_mov_b32 {T[0]}, {V_END}
	v_mov_b32 {T[1]}, 0
	v_lshlrev_b64 {V_DST}, 3, v[16:17]
	v_mov_b32 {T[2]}, s11
	v_add_co_u32 v20, vcc, s10, v20
	s_nop 1
	v_addc_co_u32 v21, vcc, {T[2]}, v21, vcc
	s_getpc_b64 {S_RET}
{p}_pc2:
	s_add_u32 s74, s74, {p}_ret2 - {p}_pc2
	s_addc_u32 s75, s75, 0
	s_branch {p}_prune
{p}_ret2:
	s_mov_b64 exec, -1
	; child = {{ base + (rank + 1) * len - count, count, high | kept << 16 }} for the pruned lanes
	v_sub_u32 {V_END}, {V_END}, {V_COUNT}
	v_lshl_or_b32 {T[5]}, {V_KEPT}, 16, {V_HIGH}
	v_cndmask_b32_e64 {V_COFF}, {V_COFF}, {V_END}, {S_PRUNE}
	v_cndmask_b32_e64 {V_CLEN}, {V_CLEN}, {V_COUNT}, {S_PRUNE}
	v_cndmask_b32_e64 {V_CRC}, {V_CRC}, {T[5]}, {S_PRUNE}
	s_branch {p}_store
{p}_overflow:
	; arena full: the children keep the parent tape; the bump pointer is clamped back to the capacity (it must not creep
	; towards 2^32 under a long run of failures, and it must never fall below a range that was granted - which giving the
	; reservation back by subtraction could do)
	v_cmp_eq_u32 vcc, 0, {V_LANE}
	s_and_saveexec_b64 {S_SAVE}, vcc
	v_mov_b32 {T[0]}, 1
	v_mov_b32 {T[1]}, 0
	v_mov_b32 {T[2]}, {S_ARENACAP}
	global_atomic_add {T[1]}, {T[0]}, {S_STATE} offset:{o['arena_overflow']}
	global_atomic_umin {T[1]}, {T[2]}, {S_STATE} offset:{o['arena_head']}
	s_mov_b64 exec, {S_SAVE}
{p}_store:
	; diagnostics (flags bit 0; the atomics serialise, never in production runs): shader clocks of the forward pass and
	; of classify + prune summed per level (stat[16 + l], stat[24 + l]), slots (stat[8 + l]), slowest slot (stat[l])
	s_bitcmp1_b32 s101, 0
	s_cbranch_scc0 {p}_noprobe
	s_memtime s[56:57]
	s_waitcnt lgkmcnt(0)
	s_sub_u32 s56, s56, s62
	s_subb_u32 s57, s57, s63
	s_sub_u32 s62, s62, s60
	s_subb_u32 s63, s63, s61
	s_lshl_b32 {S_T0}, {S_LEVEL}, 3
	v_cmp_eq_u32 vcc, 0, {V_LANE}
	s_and_saveexec_b64 {S_SAVE}, vcc
	v_mov_b32 {T[2]}, {S_T0}
	v_mov_b32 {T[0]}, s62
	v_mov_b32 {T[1]}, s63
	global_atomic_add_x2 {T[2]}, v[16:17], {S_STATE} offset:{o['stat'] + 8 * 16}
	v_mov_b32 {T[0]}, s56
	v_mov_b32 {T[1]}, s57
	global_atomic_add_x2 {T[2]}, v[16:17], {S_STATE} offset:{o['stat'] + 8 * 24}
	s_add_u32 s56, s56, s62
	s_addc_u32 s57, s57, s63
	s_nop 0
	v_mov_b32 {T[0]}, s56
	v_mov_b32 {T[1]}, s57
	global_atomic_umax_x2 {T[2]}, v[16:17], {S_STATE} offset:{o['stat']}
	v_mov_b32 {T[0]}, 1
	v_mov_b32 {T[1]}, 0
	global_atomic_add_x2 {T[2]}, v[16:17], {S_STATE} offset:{o['stat'] + 8 * 8}
	s_mov_b64 exec, {S_SAVE}
{p}_noprobe:
	global_store_dword {V_L4}, {V_COFF}, {S_SLOT} offset:{SL_COFF}
	global_store_dword {V_L4}, {V_CLEN}, {S_SLOT} offset:{SL_CLEN}
	global_store_dword {V_L4}, {V_CRC}, {S_SLOT} offset:{SL_CRC}
	s_branch {p}_outer
{p}_exit:
	s_bitcmp1_b32 s70, 0
	s_cbranch_scc0 {p}_quit
	s_and_b32 s70, s70, 0xfffffffe
	s_bfe_u32 {S_BIG}, s70, 0x10001
	s_xor_b32 {S_BIG}, {S_BIG}, 1
	s_xor_b32 s70, s70, 2                            ; (bit 1 stays the list being walked: the export mode reads it)
	s_lshr_b32 {S_LEVEL}, s70, 8
	s_mov_b32 {S_SI}, s86
	s_branch {p}_list
{p}_quit:
	s_endpgm
{p}_end:
	.size {name}, {p}_end - {name}
	.rodata
	.p2align 6
	.amdhsa_kernel {name}
		.amdhsa_group_segment_fixed_size 0
		.amdhsa_private_segment_fixed_size 0
		.amdhsa_kernarg_size 40
		.amdhsa_user_sgpr_count 2
		.amdhsa_user_sgpr_kernarg_segment_ptr 1
		.amdhsa_system_sgpr_workgroup_id_x 1
		.amdhsa_system_sgpr_workgroup_id_y 0
		.amdhsa_system_sgpr_workgroup_id_z 0
		.amdhsa_system_vgpr_workitem_id 0
		.amdhsa_next_free_vgpr {self.n_vgpr}
		.amdhsa_next_free_sgpr 102
		.amdhsa_accum_offset {(self.n_vgpr + 3) // 4 * 4}
		.amdhsa_reserve_vcc 1
		.amdhsa_float_round_mode_32 0
		.amdhsa_float_round_mode_16_64 0
		.amdhsa_float_denorm_mode_32 3
		.amdhsa_float_denorm_mode_16_64 3
		.amdhsa_dx10_clamp 1
		.amdhsa_ieee_mode 1
	.end_amdhsa_kernel
	.text""")
        if self.trans:
            import gen_trans
            gen_trans.embed(a, self.trans, v_base=self.t_base, prefix=self.t_prefix, s_map=T_SMAP)
        self.emit_forward()
        self.emit_prune()


def gen_tilesv(a, off, nr, ncw, trans=None):
    t = TilesV(a, off, nr, ncw, trans=trans)
    t.emit_kernel()
    return t.name, 40, t.n_vgpr, [(8, "global_buffer")] + [(4, "by_value")] * 8
