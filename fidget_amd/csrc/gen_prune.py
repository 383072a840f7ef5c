"""fh_prune1 — the prune sweep of ONE child tile per wavefront, in scalar code (gfx950).

The reverse sweep of VmData::simplify (fidget-core/src/vm/data.rs:123-318, restated as
prune_sweep in kernels.hip) is data dependent per child: in the 64-children-in-lockstep form
(fh_tiles) every op costs the wave the full path as soon as one lane is live.  For the long
tapes of the pre-pass levels (the root tape: 6363 ops for prospero, of which a child keeps
~9 %) it is far cheaper to give every child its own wave and run the sweep on the scalar unit:
dead ops are skipped by a real branch after ~15 instructions.

State: old register -> new register map in two VGPRs (lane = old register, v_readlane /
v_writelane with a scalar index), free-register pool as two 64-bit SGPR masks, the child's
choices from S->chw (written by fh_tiles in export mode), the tape through the scalar cache
(8 ops per load, double buffered, walked backwards).

kernarg: { FhRenderState* S; u32 level; u32 big; u32 max_choices; u32 mode }
         mode 2 (tape groups, level 0): 64 workgroups per block of root tiles, slot = block * n_tgroups; the choice
         words come from S->chwr (k_tscatter3d) instead of chw[big]
grid   : 64 workgroups (of one wave) per slot; workgroup = slot * 64 + child lane
Limits : <= 128 registers
"""
from gen_tiles import SLOT_SIZE, SL_COFF, SL_CLEN, SL_CRC, SL_XYZ

S_KERNARG = "s[0:1]"
S_WG = "s2"
S_STATE = "s[4:5]"
S_LEVEL, S_BIG, S_MAXCH = "s6", "s7", "s3"
S_SLOT = "s[8:9]"
S_OFF, S_LEN, S_RC = "s12", "s13", "s14"     # slot header s[12:15]
S_NCH = "s10"
S_C = "s11"                                    # child lane
S_TAPE = "s[16:17]"
S_CHWP = "s[18:19]"                            # address of this child's word 0
S_DST = "s[20:21]"
S_B, S_CITOP, S_NNB, S_EL = "s22", "s23", "s24", "s7"   # current batch, lowest choice index decoded so far, batch in flight, emit lane
S_MODE = "s87"
S_POOLA, S_POOLB = "s[26:27]", "s[28:29]"
S_HIGH, S_COUNT, S_KEPT = "s30", "s31", "s32"
S_END = "s33"                                  # end of the arena slot, in ops
S_LIVEA, S_LIVEB = "s[36:37]", "s[38:39]"      # old registers with a mapping (= map[r] != DEAD)
S_CAND, S_FORCE, S_LT64 = "s[40:41]", "s[42:43]", "s[44:45]"   # batch lanes still to visit; OUTPUT ops; out < 64
S_M, S_M2 = "s[48:49]", "s[62:63]"
S_VCUR, S_VNXT, S_VNN = "s[50:51]", "s[52:53]", "s[54:55]"     # valid lanes of the three batches in the pipeline
S_LOADP, S_CHUNKP = "s[58:59]", "s[60:61]"
S_W0, S_W1, S_W = "s68", "s69", "s[68:69]"
S_OP, S_OUT, S_A = "s70", "s71", "s72"
S_NO, S_MA, S_MB = "s73", "s74", "s75"
S_CH, S_CLS = "s76", "s77"                     # choice of this op; class: 0 none, 1 reg,reg, 2 reg,imm
S_T0, S_T1, S_T2, S_T3 = "s78", "s79", "s80", "s81"
S_T64 = "s[82:83]"
S_E0, S_E1 = "s84", "s85"
S_ALIAS = "s86"
S_RET = "s[88:89]"
S_MRR, S_MRI, S_MNOA = "s[90:91]", "s[92:93]", "s[94:95]"   # opcode-set bit masks
V_LANE, V_MAPA, V_MAPB, V_L8, V_E0, V_E1, V_ZERO, V_T = "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7"
V_CW0, V_CW1, V_CCH, V_COUT = "v8", "v9", "v10", "v11"         # current batch: op words, choice, out & 63
V_NW0, V_NW1, V_NCI, V_NCHW = "v12", "v13", "v14", "v15"       # next batch: op words, choice index, choice word (in flight)
V_NN = "v[16:17]"                                              # the batch below, in flight
DEAD = 0xFF


class Prune1:
    def __init__(self, a, off):
        self.a, self.off = a, off
        self.n = 0
        self.ool = []     # rare paths (registers >= 64), kept out of the fall-through code: a taken branch costs far more than an instruction

    def lab(self, stem):
        self.n += 1
        return f".Lfh_prune1_{stem}_{self.n}"

    def map_read(self, dst, reg):
        """dst = map[reg] (both halves read, the right one selected: no branch)"""
        self.a(f"""
	v_readlane_b32 {dst}, {V_MAPA}, {reg}
	v_readlane_b32 {S_T3}, {V_MAPB}, {reg}
	s_cmp_lt_u32 {reg}, 64
	s_cselect_b32 {dst}, {dst}, {S_T3}""")

    def map_write(self, reg, val, live):
        hi, done = self.lab("mw_hi"), self.lab("mw_done")
        bit = "s_bitset1_b64" if live else "s_bitset0_b64"
        self.a(f"""
	s_mov_b32 m0, {reg}
	s_cmp_lt_u32 {reg}, 64
	s_cbranch_scc0 {hi}
	v_writelane_b32 {V_MAPA}, {val}, m0
	{bit} {S_LIVEA}, {reg}
{done}:""")
        self.ool.append(f"""
{hi}:
	v_writelane_b32 {V_MAPB}, {val}, m0
	{bit} {S_LIVEB}, {reg}
	s_branch {done}""")

    def give(self, reg):
        hi, done = self.lab("gv_hi"), self.lab("gv_done")
        self.a(f"""
	s_cmp_lt_u32 {reg}, 64
	s_cbranch_scc0 {hi}
	s_bitset1_b64 {S_POOLA}, {reg}
{done}:""")
        self.ool.append(f"""
{hi}:
	s_bitset1_b64 {S_POOLB}, {reg}
	s_branch {done}""")

    def take(self, dst):
        hi, done = self.lab("tk_hi"), self.lab("tk_done")
        self.a(f"""
	s_ff1_i32_b64 {dst}, {S_POOLA}
	s_cmp_eq_i32 {dst}, -1
	s_cbranch_scc1 {hi}
	s_bitset0_b64 {S_POOLA}, {dst}
{done}:
	s_add_u32 {S_T3}, {dst}, 1
	s_max_u32 {S_HIGH}, {S_HIGH}, {S_T3}""")
        self.ool.append(f"""
{hi}:
	s_ff1_i32_b64 {dst}, {S_POOLB}
	s_bitset0_b64 {S_POOLB}, {dst}
	s_add_u32 {dst}, {dst}, 64
	s_branch {done}""")

    def use(self, dst, reg):
        """dst = new register of old value `reg`, allocated on first (i.e. last) use"""
        ok = self.lab("use_ok")
        self.map_read(dst, reg)
        self.a(f"\ts_cmp_eq_u32 {dst}, {DEAD}\n\ts_cbranch_scc0 {ok}")
        self.take(dst)
        self.map_write(reg, dst, True)
        self.a(f"{ok}:")

    def emit_op(self):
        """append {S_E0, S_E1} below what was written so far: op n of the child's tape (counted from its
        end) sits in lane 63 - n % 64 of v[4:5]; a full buffer goes out as one 512-byte store"""
        skip = self.lab("emit_skip")
        self.a(f"""
	s_mov_b32 m0, {S_EL}
	s_add_u32 {S_COUNT}, {S_COUNT}, 1
	v_writelane_b32 {V_E0}, {S_E0}, m0
	v_writelane_b32 {V_E1}, {S_E1}, m0
	s_sub_u32 {S_EL}, {S_EL}, 1
	s_cbranch_scc0 {skip}
	s_lshr_b32 {S_T3}, {S_COUNT}, 6
	s_lshl_b32 {S_T3}, {S_T3}, 9
	s_sub_u32 s60, s20, {S_T3}
	s_subb_u32 s61, s21, 0
	s_mov_b64 exec, -1
	global_store_dwordx2 {V_L8}, v[4:5], {S_CHUNKP}
	s_mov_b64 exec, 1
	s_mov_b32 {S_EL}, 63
{skip}:""")

    def advance(self):
        """one step of the three-deep batch pipeline: current <- next <- in flight; decode the new `next`
        (choice indices per lane, its choice words requested); request the batch below"""
        a = self.a
        noload, loaded = self.lab("adv_noload"), self.lab("adv_loaded")
        a(f"""
	s_waitcnt vmcnt(0)
	s_mov_b64 exec, -1
	v_mov_b32 {V_CW0}, {V_NW0}
	v_mov_b32 {V_CW1}, {V_NW1}
	v_and_b32 {V_T}, 15, {V_NCI}
	v_lshlrev_b32 {V_T}, 1, {V_T}
	v_lshrrev_b32 {V_CCH}, {V_T}, {V_NCHW}
	v_and_b32 {V_CCH}, 3, {V_CCH}
	s_mov_b64 {S_VCUR}, {S_VNXT}
	s_mov_b64 {S_VNXT}, {S_VNN}
	v_and_b32 {V_T}, 0xff, {V_CW0}
	v_cmp_eq_u32_e64 {S_FORCE}, 0, {V_T}
	v_bfe_u32 {V_COUT}, {V_CW0}, 8, 12
	v_cmp_gt_u32_e64 {S_LT64}, 64, {V_COUT}
	v_and_b32 {V_COUT}, 63, {V_COUT}
	s_and_b64 {S_FORCE}, {S_FORCE}, {S_VCUR}
	; ops that hand their own register on (a decided choice or a copy whose surviving operand IS the
	; output register - the accumulator of a min / max chain) change nothing: never candidates
	v_bfe_u32 v18, {V_CW0}, 8, 12
	v_lshrrev_b32 v19, 20, {V_CW0}
	v_subrev_u32 v20, 30, {V_T}
	v_cmp_gt_u32_e64 {S_M}, 4, v20                    ; choice, reg,reg
	v_subrev_u32 v20, 42, {V_T}
	v_cmp_gt_u32_e64 {S_M2}, 4, v20                   ; choice, reg,imm
	v_cmp_eq_u32_e64 s[64:65], v19, v18               ; a == out
	v_cmp_eq_u32_e64 s[66:67], 1, {V_CCH}             ; Left
	s_or_b64 {S_M2}, {S_M2}, {S_M}
	s_and_b64 {S_M2}, {S_M2}, s[66:67]
	v_cmp_eq_u32_e64 s[66:67], 2, {V_T}               ; COPY_REG
	s_or_b64 {S_M2}, {S_M2}, s[66:67]
	s_and_b64 {S_M2}, {S_M2}, s[64:65]
	v_cmp_eq_u32_e64 s[64:65], {V_CW1}, v18           ; b == out
	v_cmp_eq_u32_e64 s[66:67], 2, {V_CCH}             ; Right
	s_and_b64 {S_M}, {S_M}, s[64:65]
	s_and_b64 {S_M}, {S_M}, s[66:67]
	s_or_b64 {S_M2}, {S_M2}, {S_M}
	s_andn2_b64 {S_CAND}, {S_VCUR}, {S_M2}
	v_mov_b32 {V_NW0}, v16
	v_mov_b32 {V_NW1}, v17
	v_and_b32 {V_T}, 0xff, {V_NW0}
	v_subrev_u32 v18, 30, {V_T}
	v_cmp_gt_u32_e64 {S_M}, 4, v18
	v_subrev_u32 v18, 42, {V_T}
	v_cmp_gt_u32_e64 {S_M2}, 4, v18
	s_or_b64 {S_M}, {S_M}, {S_M2}
	s_and_b64 {S_M}, {S_M}, {S_VNXT}
	s_bcnt1_i32_b64 {S_T0}, {S_M}
	s_sub_u32 {S_CITOP}, {S_CITOP}, {S_T0}
	v_mbcnt_lo_u32_b32 {V_NCI}, s48, 0
	v_mbcnt_hi_u32_b32 {V_NCI}, s49, {V_NCI}
	v_add_u32 {V_NCI}, {S_CITOP}, {V_NCI}
	v_lshrrev_b32 v22, 4, {V_NCI}
	v_lshlrev_b32 v22, 8, v22
	s_mov_b64 exec, {S_M}
	global_load_dword {V_NCHW}, v22, {S_CHWP}
	s_sub_u32 {S_NNB}, {S_NNB}, 1
	s_cmp_lt_i32 {S_NNB}, 0
	s_cbranch_scc1 {noload}
	s_mov_b64 {S_VNN}, -1
	s_sub_u32 s58, s58, 0x200
	s_subb_u32 s59, s59, 0
	s_mov_b64 exec, -1
	global_load_dwordx2 {V_NN}, {V_L8}, {S_LOADP}
	s_branch {loaded}
{noload}:
	s_mov_b64 {S_VNN}, 0
{loaded}:
	s_mov_b64 exec, 1""")

    def emit(self):
        a, o = self.a, self.off
        name = "fh_prune1"
        nxt = ".Lfh_prune1_next"
        a(f"""
	.text
	.protected {name}
	.globl {name}
	.p2align 8
	.type {name},@function
{name}:
	s_load_dwordx2 {S_STATE}, {S_KERNARG}, 0x0
	s_load_dwordx4 s[8:11], {S_KERNARG}, 0x8
	v_mov_b32 {V_MAPA}, {DEAD}
	v_mov_b32 {V_MAPB}, {DEAD}
	v_mov_b32 {V_ZERO}, 0
	v_lshlrev_b32 {V_L8}, 3, {V_LANE}
	s_mov_b64 exec, 1
	s_waitcnt lgkmcnt(0)
	s_mov_b32 {S_LEVEL}, s8
	s_mov_b32 {S_BIG}, s9
	s_mov_b32 {S_MAXCH}, s10
	s_mov_b32 {S_MODE}, s11
	s_lshr_b32 {S_T0}, {S_WG}, 6                  ; slot (mode 1: block of root tiles)
	s_and_b32 {S_C}, {S_WG}, 63                   ; child lane
	; n_slots[big][level], slots[big], chw[big]
	s_lshl_b32 {S_T1}, {S_BIG}, 3
	s_add_u32 {S_T1}, {S_T1}, {S_LEVEL}
	s_lshl_b32 {S_T1}, {S_T1}, 2
	s_add_u32 s82, s4, {S_T1}
	s_addc_u32 s83, s5, 0
	s_load_dword {S_T2}, {S_T64}, {o['n_slots']}
	s_lshl_b32 {S_T1}, {S_BIG}, 3
	s_add_u32 s82, s4, {S_T1}
	s_addc_u32 s83, s5, 0
	s_load_dwordx2 {S_SLOT}, {S_T64}, {o['slots']}
	s_load_dwordx2 {S_CHWP}, {S_T64}, {o['chw']}
	s_load_dwordx2 {S_TAPE}, {S_STATE}, {o['arena']}
	s_cmp_eq_u32 {S_MODE}, 2
	s_cbranch_scc0 .Lfh_prune1_chwok
	s_waitcnt lgkmcnt(0)
	s_load_dwordx2 {S_CHWP}, {S_STATE}, {o['chwr']}
	s_load_dword {S_T1}, {S_STATE}, {o['n_tgroups']}
	s_waitcnt lgkmcnt(0)
	s_mul_i32 {S_T0}, {S_T0}, {S_T1}                ; the block's primary slot
.Lfh_prune1_chwok:
	s_waitcnt lgkmcnt(0)
	s_cmp_ge_u32 {S_T0}, {S_T2}
	s_cbranch_scc1 .Lfh_prune1_exit
	; slot = slots + si * sizeof(FhSlot); chw column of this child
	s_mul_i32 {S_T1}, {S_T0}, {SLOT_SIZE}
	s_mul_hi_u32 {S_T2}, {S_T0}, {SLOT_SIZE}
	s_add_u32 s8, s8, {S_T1}
	s_addc_u32 s9, s9, {S_T2}
	s_add_u32 {S_T1}, {S_MAXCH}, 15
	s_lshr_b32 {S_T1}, {S_T1}, 4
	s_lshl_b32 {S_T1}, {S_T1}, 8                  ; bytes of choice words per slot
	s_mul_hi_u32 {S_T2}, {S_T0}, {S_T1}
	s_mul_i32 {S_T1}, {S_T0}, {S_T1}
	s_add_u32 s18, s18, {S_T1}
	s_addc_u32 s19, s19, {S_T2}
	s_lshl_b32 {S_T1}, {S_C}, 2
	s_add_u32 s18, s18, {S_T1}
	s_addc_u32 s19, s19, 0
	; is this child marked (c_len == ~0)?  c_off = end of its arena slot
	s_add_u32 s82, s8, {S_T1}
	s_addc_u32 s83, s9, 0
	s_load_dword {S_T2}, {S_T64}, {SL_CLEN}
	s_load_dword {S_END}, {S_T64}, {SL_COFF}
	s_load_dwordx4 s[12:15], {S_SLOT}, 0x0
	s_waitcnt lgkmcnt(0)
	s_cmp_eq_u32 {S_T2}, -1
	s_cbranch_scc0 .Lfh_prune1_exit
	; dst = arena + 8 * end
	s_mov_b64 {S_T64}, {S_TAPE}
	s_mov_b32 {S_T0}, {S_END}
	s_mov_b32 {S_T1}, 0
	s_lshl_b64 s[78:79], s[78:79], 3
	s_add_u32 s20, s82, {S_T0}
	s_addc_u32 s21, s83, {S_T1}
	; opcode sets: choice reg,reg 30..33 ; choice reg,imm 42..45 ; no operand a: INPUT (1), COPY_IMM (3)
	s_mov_b32 s90, 0xc0000000
	s_mov_b32 s91, 0x3
	s_mov_b32 s92, 0
	s_mov_b32 s93, 0x3c00
	s_mov_b32 s94, 0xa
	s_mov_b32 s95, 0
	s_mov_b32 {S_HIGH}, 0
	s_mov_b32 {S_COUNT}, 0
	s_mov_b32 {S_KEPT}, 0
	s_mov_b32 {S_EL}, 63
	s_mov_b64 {S_LIVEA}, 0
	s_mov_b64 {S_LIVEB}, 0
	s_mov_b64 {S_POOLA}, -1
	s_mov_b64 {S_POOLB}, -1
.Lfh_prune1_tape:
	; tape = arena + 8 * off
	s_lshr_b32 {S_NCH}, {S_RC}, 16
	s_mov_b32 {S_T0}, {S_OFF}
	s_mov_b32 {S_T1}, 0
	s_lshl_b64 s[78:79], s[78:79], 3
	s_add_u32 s16, s82, {S_T0}
	s_addc_u32 s17, s83, {S_T1}
	s_cmp_eq_u32 {S_LEN}, 0
	s_cbranch_scc1 .Lfh_prune1_done
	; batches of 64 ops, last first: lane = op index % 64.  Top batch: lanes 0 .. (len-1) % 64
	s_sub_u32 {S_T0}, {S_LEN}, 1
	s_lshr_b32 {S_B}, {S_T0}, 6
	s_and_b32 {S_T0}, {S_T0}, 63
	s_sub_u32 {S_T0}, 63, {S_T0}
	s_lshr_b64 {S_VNN}, -1, {S_T0}
	s_mov_b64 {S_VNXT}, 0
	s_mov_b64 {S_VCUR}, 0
	s_mov_b32 {S_NNB}, {S_B}
	s_mov_b32 {S_CITOP}, {S_NCH}
	s_lshl_b32 {S_T0}, {S_B}, 9
	s_add_u32 s58, s16, {S_T0}
	s_addc_u32 s59, s17, 0
	s_mov_b64 exec, {S_VNN}
	global_load_dwordx2 {V_NN}, {V_L8}, {S_LOADP}
	s_mov_b64 exec, 1""")
        self.advance()
        self.advance()
        a(f"""
{nxt}:
	; which ops of the batch, below the last one handled, write a register that is wanted now?
	s_mov_b64 exec, -1
	v_lshrrev_b64 v[18:19], {V_COUT}, {S_LIVEA}
	v_lshrrev_b64 v[20:21], {V_COUT}, {S_LIVEB}
	v_cndmask_b32_e64 v18, v20, v18, {S_LT64}
	v_and_b32 v18, 1, v18
	v_cmp_ne_u32_e64 {S_M}, 0, v18
	s_mov_b64 exec, 1
	{"s_mov_b64 s[48:49], 0" if __import__("os").environ.get("FH_EXP") == "prune_nolive" else ""}
	s_or_b64 {S_M}, {S_M}, {S_FORCE}
	s_and_b64 {S_M}, {S_M}, {S_CAND}
	s_cbranch_scc0 .Lfh_prune1_bdone
	; the highest such op: everything above it is dead and leaves the state alone
	s_flbit_i32_b64 {S_T0}, {S_M}
	s_sub_u32 {S_T0}, 63, {S_T0}
	s_bfm_b64 {S_M2}, {S_T0}, 0
	s_and_b64 {S_CAND}, {S_CAND}, {S_M2}
	v_readlane_b32 {S_W0}, {V_CW0}, {S_T0}
	v_readlane_b32 {S_W1}, {V_CW1}, {S_T0}
	v_readlane_b32 {S_CH}, {V_CCH}, {S_T0}
	s_and_b32 {S_OP}, {S_W0}, 0xff
	s_bfe_u32 {S_OUT}, {S_W0}, 0xc0008
	s_lshr_b32 {S_A}, {S_W0}, 20
	s_mov_b32 {S_CLS}, 0
	s_bitcmp1_b64 {S_MRR}, {S_OP}
	s_cselect_b32 {S_CLS}, 1, 0
	s_bitcmp1_b64 {S_MRI}, {S_OP}
	s_cselect_b32 {S_CLS}, 2, {S_CLS}
	s_cmp_eq_u32 {S_OP}, 0
	s_cbranch_scc1 .Lfh_prune1_output""")
        self.map_read(S_NO, S_OUT)
        a(f"""
	s_cmp_eq_u32 {S_NO}, {DEAD}
	s_cbranch_scc1 {nxt}                          ; value never used
	s_mov_b32 {S_T0}, {DEAD}""")
        self.map_write(S_OUT, S_T0, False)
        a(f"""
	; ---- decided choices / copies alias `out` with the surviving operand ---------------------
	s_mov_b32 {S_ALIAS}, -1
	s_cmp_eq_u32 {S_OP}, 2
	s_cselect_b32 {S_ALIAS}, {S_A}, {S_ALIAS}
	s_cmp_eq_u32 {S_CLS}, 0
	s_cbranch_scc1 .Lfh_prune1_aliased
	s_cmp_eq_u32 {S_CH}, 1
	s_cselect_b32 {S_ALIAS}, {S_A}, {S_ALIAS}
	s_cmp_eq_u32 {S_CH}, 2
	s_cbranch_scc0 .Lfh_prune1_aliased
	s_cmp_eq_u32 {S_CLS}, 1
	s_cbranch_scc0 .Lfh_prune1_copyimm
	s_mov_b32 {S_ALIAS}, {S_W1}
.Lfh_prune1_aliased:
	s_cmp_eq_i32 {S_ALIAS}, -1
	s_cbranch_scc1 .Lfh_prune1_keep""")
        self.map_read(S_MA, S_ALIAS)
        a(f"""
	s_cmp_eq_u32 {S_MA}, {DEAD}
	s_cbranch_scc0 .Lfh_prune1_copyreg""")
        self.map_write(S_ALIAS, S_NO, True)       # the operand takes the register over, nothing is emitted
        a(f"""
	s_branch {nxt}
.Lfh_prune1_copyreg:""")
        self.give(S_NO)
        a(f"""
	s_lshl_b32 {S_E0}, {S_NO}, 8
	s_lshl_b32 {S_T0}, {S_MA}, 20
	s_or_b32 {S_E0}, {S_E0}, {S_T0}
	s_or_b32 {S_E0}, {S_E0}, 2
	s_mov_b32 {S_E1}, 0""")
        self.emit_op()
        a(f"""
	s_branch {nxt}
.Lfh_prune1_copyimm:""")
        self.give(S_NO)
        a(f"""
	s_lshl_b32 {S_E0}, {S_NO}, 8
	s_or_b32 {S_E0}, {S_E0}, 3
	s_mov_b32 {S_E1}, {S_W1}""")
        self.emit_op()
        a(f"""
	s_branch {nxt}
.Lfh_prune1_keep:""")
        self.give(S_NO)
        a(f"""
	s_mov_b32 {S_MA}, 0
	s_mov_b32 {S_E1}, {S_W1}
	s_bitcmp1_b64 {S_MNOA}, {S_OP}
	s_cbranch_scc1 .Lfh_prune1_noa""")
        self.use(S_MA, S_A)
        a(f"""
.Lfh_prune1_noa:
	; operand b: reg,reg forms 22..33
	s_sub_u32 {S_T0}, {S_OP}, 22
	s_cmp_lt_u32 {S_T0}, 12
	s_cbranch_scc0 .Lfh_prune1_nob""")
        self.use(S_MB, S_W1)
        a(f"""
	s_mov_b32 {S_E1}, {S_MB}
.Lfh_prune1_nob:
	s_cmp_lg_u32 {S_CLS}, 0
	s_addc_u32 {S_KEPT}, {S_KEPT}, 0
	s_lshl_b32 {S_E0}, {S_NO}, 8
	s_lshl_b32 {S_T0}, {S_MA}, 20
	s_or_b32 {S_E0}, {S_E0}, {S_T0}
	s_or_b32 {S_E0}, {S_E0}, {S_OP}""")
        self.emit_op()
        a(f"""
	s_branch {nxt}
.Lfh_prune1_output:""")
        self.use(S_MA, S_A)
        a(f"""
	s_lshl_b32 {S_E0}, {S_MA}, 20
	s_mov_b32 {S_E1}, {S_W1}
""")
        self.emit_op()
        a(f"""
	s_branch {nxt}
.Lfh_prune1_bdone:
	; nothing else in this batch: the next one down
	s_cmp_eq_u32 {S_B}, 0
	s_cbranch_scc1 .Lfh_prune1_done
	s_sub_u32 {S_B}, {S_B}, 1""")
        self.advance()
        a(f"""
	s_branch {nxt}""")
        for code in self.ool:
            a(code)
        a(f"""
.Lfh_prune1_done:
	; the ops still in the buffer: lanes 64 - n .. 63
	s_and_b32 {S_T0}, {S_COUNT}, 63
	s_cbranch_scc0 .Lfh_prune1_flushed
	s_sub_u32 {S_T1}, 64, {S_T0}
	s_bfm_b64 {S_M2}, {S_T0}, {S_T1}
	s_lshr_b32 {S_T3}, {S_COUNT}, 6
	s_add_u32 {S_T3}, {S_T3}, 1
	s_lshl_b32 {S_T3}, {S_T3}, 9
	s_sub_u32 s60, s20, {S_T3}
	s_subb_u32 s61, s21, 0
	s_mov_b64 exec, {S_M2}
	global_store_dwordx2 {V_L8}, v[4:5], {S_CHUNKP}
	s_mov_b64 exec, 1
.Lfh_prune1_flushed:
	; child = {{ end - count, count, high | kept << 16 }}
	s_sub_u32 {S_T0}, {S_END}, {S_COUNT}
	s_lshl_b32 {S_T1}, {S_KEPT}, 16
	s_or_b32 {S_T1}, {S_T1}, {S_HIGH}
	s_lshl_b32 {S_T2}, {S_C}, 2
	s_add_u32 s82, s8, {S_T2}
	s_addc_u32 s83, s9, 0
	v_mov_b32 v18, {S_T0}
	v_mov_b32 v19, {S_COUNT}
	v_mov_b32 {V_T}, {S_T1}
	global_store_dword {V_ZERO}, v18, {S_T64} offset:{SL_COFF}
	global_store_dword {V_ZERO}, v19, {S_T64} offset:{SL_CLEN}
	global_store_dword {V_ZERO}, {V_T}, {S_T64} offset:{SL_CRC}
.Lfh_prune1_exit:
	s_endpgm
.Lfh_prune1_end:
	.size {name}, .Lfh_prune1_end - {name}
	.rodata
	.p2align 6
	.amdhsa_kernel {name}
		.amdhsa_group_segment_fixed_size 0
		.amdhsa_private_segment_fixed_size 0
		.amdhsa_kernarg_size 24
		.amdhsa_user_sgpr_count 2
		.amdhsa_user_sgpr_kernarg_segment_ptr 1
		.amdhsa_system_sgpr_workgroup_id_x 1
		.amdhsa_system_sgpr_workgroup_id_y 0
		.amdhsa_system_sgpr_workgroup_id_z 0
		.amdhsa_system_vgpr_workitem_id 0
		.amdhsa_next_free_vgpr 24
		.amdhsa_next_free_sgpr 102
		.amdhsa_accum_offset 24
		.amdhsa_reserve_vcc 1
		.amdhsa_float_round_mode_32 0
		.amdhsa_float_round_mode_16_64 0
		.amdhsa_float_denorm_mode_32 3
		.amdhsa_float_denorm_mode_16_64 3
		.amdhsa_dx10_clamp 1
		.amdhsa_ieee_mode 1
	.end_amdhsa_kernel
	.text""")


def gen_prune1(a, off):
    Prune1(a, off).emit()
    return "fh_prune1", 24, 24, [(8, "global_buffer")] + [(4, "by_value")] * 4
