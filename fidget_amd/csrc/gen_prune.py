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
         mode 1 (tape groups, level 0): the wave walks the child's needed groups last to first and writes
         ONE tape: group after group, each result folded into r0 right away (r0 = op(r0, r1))
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
S_K, S_CI, S_CW = "s22", "s23", "s24"
S_HAVEW = "s7"                                  # (s7 = `big` is dead after the prologue)
S_MODE, S_LIVEG, S_RANK, S_SLOTI, S_NLIVE, S_NG, S_GOP, S_GI = "s87", "s97", "s100", "s101", "s88", "s89", "s15", "s25"
S_CWN, S_PREF = "s34", "s35"                  # prefetched choice word, prefetch in flight
S_POOLA, S_POOLB = "s[26:27]", "s[28:29]"
S_HIGH, S_COUNT, S_KEPT = "s30", "s31", "s32"
S_END = "s33"                                  # end of the arena slot, in ops
S_QA, S_QB = 36, 52                            # s[36:51] current 8 ops, s[52:67] next 8 ops
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
S_QBASE = "s96"                                # index of the first op of the current batch
S_FETCH = "s[98:99]"
V_LANE, V_MAPA, V_MAPB, V_E0, V_E1, V_ZERO, V_T = "v0", "v1", "v2", "v4", "v5", "v6", "v7"
DEAD = 0xFF


class Prune1:
    def __init__(self, a, off):
        self.a, self.off = a, off
        self.n = 0

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

    def map_write(self, reg, val):
        hi, done = self.lab("mw_hi"), self.lab("mw_done")
        self.a(f"""
	s_mov_b32 m0, {reg}
	s_cmp_lt_u32 {reg}, 64
	s_cbranch_scc0 {hi}
	v_writelane_b32 {V_MAPA}, {val}, m0
	s_branch {done}
{hi}:
	v_writelane_b32 {V_MAPB}, {val}, m0
{done}:
	s_nop 0""")

    def give(self, reg):
        hi, done = self.lab("gv_hi"), self.lab("gv_done")
        self.a(f"""
	s_cmp_lt_u32 {reg}, 64
	s_cbranch_scc0 {hi}
	s_bitset1_b64 {S_POOLA}, {reg}
	s_branch {done}
{hi}:
	s_bitset1_b64 {S_POOLB}, {reg}
{done}:""")

    def take(self, dst):
        hi, done = self.lab("tk_hi"), self.lab("tk_done")
        self.a(f"""
	s_ff1_i32_b64 {dst}, {S_POOLA}
	s_cmp_eq_i32 {dst}, -1
	s_cbranch_scc1 {hi}
	s_bitset0_b64 {S_POOLA}, {dst}
	s_branch {done}
{hi}:
	s_ff1_i32_b64 {dst}, {S_POOLB}
	s_bitset0_b64 {S_POOLB}, {dst}
	s_add_u32 {dst}, {dst}, 64
{done}:
	s_add_u32 {S_T3}, {dst}, 1
	s_max_u32 {S_HIGH}, {S_HIGH}, {S_T3}""")

    def use(self, dst, reg):
        """dst = new register of old value `reg`, allocated on first (i.e. last) use"""
        ok = self.lab("use_ok")
        self.map_read(dst, reg)
        self.a(f"\ts_cmp_eq_u32 {dst}, {DEAD}\n\ts_cbranch_scc0 {ok}")
        self.take(dst)
        self.map_write(reg, dst)
        self.a(f"{ok}:")

    def emit_op(self):
        """append {S_E0, S_E1} below dst (lane 0 stores)"""
        self.a(f"""
	v_mov_b32 {V_E0}, {S_E0}
	v_mov_b32 {V_E1}, {S_E1}
	s_add_u32 s20, s20, -8
	s_addc_u32 s21, s21, -1
	s_add_u32 {S_COUNT}, {S_COUNT}, 1
	global_store_dwordx2 {V_ZERO}, v[4:5], {S_DST}""")

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
	s_mov_b64 exec, 1
	s_waitcnt lgkmcnt(0)
	s_mov_b32 {S_LEVEL}, s8
	s_mov_b32 {S_BIG}, s9
	s_mov_b32 {S_MAXCH}, s10
	s_mov_b32 {S_MODE}, s11
	s_lshr_b32 {S_T0}, {S_WG}, 6                  ; slot (mode 1: block of root tiles)
	s_and_b32 {S_C}, {S_WG}, 63                   ; child lane
	s_load_dword {S_NG}, {S_STATE}, {o['n_tgroups']}
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
	s_waitcnt lgkmcnt(0)
	s_cmp_eq_u32 {S_MODE}, 0
	s_cbranch_scc1 .Lfh_prune1_slotok
	s_mul_i32 {S_T0}, {S_T0}, {S_NG}               ; the block's primary slot
.Lfh_prune1_slotok:
	s_mov_b32 {S_SLOTI}, {S_T0}
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
	s_mov_b32 {S_NLIVE}, 0
	s_cmp_eq_u32 {S_MODE}, 0
	s_cbranch_scc0 .Lfh_prune1_ginit
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
	s_mov_b32 {S_K}, {S_LEN}
	s_mov_b32 {S_CI}, {S_NCH}
	s_mov_b32 {S_HAVEW}, 0
	s_mov_b32 {S_PREF}, 0
	; first batch: the 8-op block holding op len-1, and the one below it
	s_sub_u32 {S_T0}, {S_LEN}, 1
	s_and_b32 {S_QBASE}, {S_T0}, -8
	s_lshl_b32 {S_T0}, {S_QBASE}, 3
	s_add_u32 s98, s16, {S_T0}
	s_addc_u32 s99, s17, 0
	s_load_dwordx16 s[{S_QA}:{S_QA + 15}], {S_FETCH}, 0x0
	s_sub_u32 s98, s98, 0x40
	s_subb_u32 s99, s99, 0
	s_cmp_eq_u32 {S_QBASE}, 0
	s_cbranch_scc1 .Lfh_prune1_first
	s_load_dwordx16 s[{S_QB}:{S_QB + 15}], {S_FETCH}, 0x0
.Lfh_prune1_first:
	s_waitcnt lgkmcnt(0)
{nxt}:
	s_sub_u32 {S_K}, {S_K}, 1
	s_cbranch_scc1 .Lfh_prune1_done
	s_cmp_ge_u32 {S_K}, {S_QBASE}
	s_cbranch_scc1 .Lfh_prune1_haveq
	; next lower block of 8 ops; prefetch the one below it
	s_waitcnt lgkmcnt(0)""")
        for i in range(0, 16, 2):
            a(f"\ts_mov_b64 s[{S_QA + i}:{S_QA + i + 1}], s[{S_QB + i}:{S_QB + i + 1}]")
        a(f"""
	s_sub_u32 {S_QBASE}, {S_QBASE}, 8
	s_sub_u32 s98, s98, 0x40
	s_subb_u32 s99, s99, 0
	s_cmp_eq_u32 {S_QBASE}, 0
	s_cbranch_scc1 .Lfh_prune1_haveq
	s_load_dwordx16 s[{S_QB}:{S_QB + 15}], {S_FETCH}, 0x0
.Lfh_prune1_haveq:
	s_sub_u32 {S_T0}, {S_K}, {S_QBASE}
	s_lshl_b32 {S_T0}, {S_T0}, 1
	s_mov_b32 m0, {S_T0}
	s_nop 0
	s_movrels_b64 {S_W}, s[{S_QA}:{S_QA + 1}]
	s_and_b32 {S_OP}, {S_W0}, 0xff
	s_bfe_u32 {S_OUT}, {S_W0}, 0xc0008
	s_lshr_b32 {S_A}, {S_W0}, 20
	; ---- choice of this op (min / max / and / or), consumed back to front ---------------------
	s_mov_b32 {S_CLS}, 0
	s_bitcmp1_b64 {S_MRR}, {S_OP}
	s_cselect_b32 {S_CLS}, 1, 0
	s_bitcmp1_b64 {S_MRI}, {S_OP}
	s_cselect_b32 {S_CLS}, 2, {S_CLS}
	s_cmp_eq_u32 {S_CLS}, 0
	s_cbranch_scc1 .Lfh_prune1_nochoice
	s_sub_u32 {S_CI}, {S_CI}, 1
	s_and_b32 {S_T0}, {S_CI}, 15
	s_cmp_eq_u32 {S_T0}, 15
	s_cselect_b32 {S_T1}, 0, {S_HAVEW}
	s_cmp_eq_u32 {S_T1}, 0
	s_cbranch_scc0 .Lfh_prune1_haveword
	; a new word of 16 choices: the prefetched one, or (first time) a direct load
	s_lshr_b32 {S_T1}, {S_CI}, 4
	s_lshl_b32 {S_T2}, {S_T1}, 8
	s_add_u32 s82, s18, {S_T2}
	s_addc_u32 s83, s19, 0
	s_cmp_eq_u32 {S_PREF}, 0
	s_cbranch_scc1 .Lfh_prune1_wdirect
	s_waitcnt lgkmcnt(0)
	s_mov_b32 {S_CW}, {S_CWN}
	s_branch .Lfh_prune1_wnext
.Lfh_prune1_wdirect:
	s_load_dword {S_CW}, {S_T64}, 0x0
	s_waitcnt lgkmcnt(0)
.Lfh_prune1_wnext:
	s_mov_b32 {S_HAVEW}, 1
	s_mov_b32 {S_PREF}, 0
	s_cmp_eq_u32 {S_T1}, 0
	s_cbranch_scc1 .Lfh_prune1_haveword
	s_sub_u32 s82, s82, 0x100
	s_subb_u32 s83, s83, 0
	s_load_dword {S_CWN}, {S_T64}, 0x0             ; the word below, needed 16 choices from now
	s_mov_b32 {S_PREF}, 1
.Lfh_prune1_haveword:
	s_lshl_b32 {S_T0}, {S_T0}, 1
	s_lshr_b32 {S_CH}, {S_CW}, {S_T0}
	s_and_b32 {S_CH}, {S_CH}, 3
.Lfh_prune1_nochoice:
	s_cmp_eq_u32 {S_OP}, 0
	s_cbranch_scc1 .Lfh_prune1_output""")
        self.map_read(S_NO, S_OUT)
        a(f"""
	s_cmp_eq_u32 {S_NO}, {DEAD}
	s_cbranch_scc1 {nxt}                          ; value never used
	s_mov_b32 {S_T0}, {DEAD}""")
        self.map_write(S_OUT, S_T0)
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
        self.map_write(S_ALIAS, S_NO)       # the operand takes the register over, nothing is emitted
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
	s_cmp_eq_u32 {S_MODE}, 0
	s_cbranch_scc1 .Lfh_prune1_outemit
	; tape groups: the group's result goes to register `rank` (COPY_REG) instead of the output
	s_min_u32 {S_T0}, {S_RANK}, 1
	s_lshl_b32 {S_T0}, {S_T0}, 8
	s_or_b32 {S_E0}, {S_E0}, {S_T0}
	s_or_b32 {S_E0}, {S_E0}, 2
	s_mov_b32 {S_E1}, 0
.Lfh_prune1_outemit:""")
        self.emit_op()
        a(f"""
	s_branch {nxt}
; ---- tape groups (mode 1) ------------------------------------------------------------------
.Lfh_prune1_ginit:
	s_lshl_b32 {S_T1}, {S_C}, 2
	s_add_u32 s82, s8, {S_T1}
	s_addc_u32 s83, s9, 0
	s_load_dword {S_LIVEG}, {S_T64}, {SL_XYZ}        ; groups this child needs (k_tcombine3d)
	s_load_dword {S_GOP}, {S_STATE}, {o['tgroup_op']}
	s_waitcnt lgkmcnt(0)
	s_bcnt1_i32_b32 {S_NLIVE}, {S_LIVEG}
	s_min_u32 {S_HIGH}, {S_NLIVE}, 2
	; Written back to front: OUTPUT r0; then, last needed group first: [r0 = op(r0, r1)] after the
	; group's ops, whose result is copied to r1 (to r0 for the first group).  Only r0 and r1 are
	; reserved, so the tape needs no more registers than its largest group + 2.
	s_mov_b32 {S_E0}, 0
	s_mov_b32 {S_E1}, 0""")
        self.emit_op()
        a(f"""
	s_mov_b32 {S_RANK}, {S_NLIVE}
.Lfh_prune1_gnext:
	s_waitcnt lgkmcnt(0)                            ; prefetches of the group just finished
	s_cmp_eq_u32 {S_LIVEG}, 0
	s_cbranch_scc1 .Lfh_prune1_finish
	s_flbit_i32_b32 {S_T0}, {S_LIVEG}
	s_sub_u32 {S_GI}, 31, {S_T0}
	s_bitset0_b32 {S_LIVEG}, {S_GI}
	s_sub_u32 {S_RANK}, {S_RANK}, 1
	s_cmp_eq_u32 {S_RANK}, 0
	s_cbranch_scc1 .Lfh_prune1_gfirst
	s_lshl_b32 {S_E0}, {S_GOP}, 0                   ; r0 = op(r0, r1): out 0, a 0, b 1
	s_mov_b32 {S_E1}, 1""")
        self.emit_op()
        a(f"""
	s_add_u32 {S_KEPT}, {S_KEPT}, 1
.Lfh_prune1_gfirst:
	s_mul_i32 {S_T0}, {S_GI}, 12
	s_add_u32 s82, s4, {S_T0}
	s_addc_u32 s83, s5, 0
	s_load_dwordx2 s[12:13], {S_T64}, {o['tgroup']}
	s_load_dword s14, {S_T64}, {o['tgroup'] + 8}
	s_load_dwordx2 {S_CHWP}, {S_STATE}, {o['chw'] + 8}
	s_load_dwordx2 {S_T64}, {S_STATE}, {o['arena']}
	; this group's choice words: chw[1] + (primary slot + g) * bytes per slot + lane * 4
	s_add_u32 {S_T1}, {S_MAXCH}, 15
	s_lshr_b32 {S_T1}, {S_T1}, 4
	s_lshl_b32 {S_T1}, {S_T1}, 8
	s_add_u32 {S_T0}, {S_SLOTI}, {S_GI}
	s_mul_hi_u32 {S_T2}, {S_T0}, {S_T1}
	s_mul_i32 {S_T1}, {S_T0}, {S_T1}
	s_waitcnt lgkmcnt(0)
	s_add_u32 s18, s18, {S_T1}
	s_addc_u32 s19, s19, {S_T2}
	s_lshl_b32 {S_T1}, {S_C}, 2
	s_add_u32 s18, s18, {S_T1}
	s_addc_u32 s19, s19, 0
	; fresh register map; registers 0 .. n_live-1 hold the groups' results and are never handed out
	s_mov_b64 exec, -1
	v_mov_b32 {V_MAPA}, {DEAD}
	v_mov_b32 {V_MAPB}, {DEAD}
	s_mov_b64 exec, 1
	s_mov_b64 {S_POOLA}, -4                        ; r0, r1 are taken
	s_mov_b64 {S_POOLB}, -1
	s_branch .Lfh_prune1_tape
.Lfh_prune1_done:
	s_cmp_eq_u32 {S_MODE}, 0
	s_cbranch_scc0 .Lfh_prune1_gnext
.Lfh_prune1_finish:
	; child = {{ end - count, count, high | kept << 16 }}
	s_sub_u32 {S_T0}, {S_END}, {S_COUNT}
	s_lshl_b32 {S_T1}, {S_KEPT}, 16
	s_or_b32 {S_T1}, {S_T1}, {S_HIGH}
	s_lshl_b32 {S_T2}, {S_C}, 2
	s_add_u32 s82, s8, {S_T2}
	s_addc_u32 s83, s9, 0
	v_mov_b32 {V_E0}, {S_T0}
	v_mov_b32 {V_E1}, {S_COUNT}
	v_mov_b32 {V_T}, {S_T1}
	global_store_dword {V_ZERO}, {V_E0}, {S_T64} offset:{SL_COFF}
	global_store_dword {V_ZERO}, {V_E1}, {S_T64} offset:{SL_CLEN}
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
		.amdhsa_next_free_vgpr 8
		.amdhsa_next_free_sgpr 102
		.amdhsa_accum_offset 8
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
    return "fh_prune1", 24, 8, [(8, "global_buffer")] + [(4, "by_value")] * 4
