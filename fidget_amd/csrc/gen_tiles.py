"""fh_tiles — the evaluate + prune step of the split 3D tile stage in gfx950 assembly.

Per FhSlot (one parent tile, one child per lane), exactly what k_teval3d (kernels.hip) does:

  1. forward interval pass over the parent tape (fidget-core/src/vm/mod.rs:325-538 semantics, as
     restated in dev_ops.hpp): interval register file in LDS [reg][lane] (8 B), one choice
     (2 bits) per min/max/and/or per lane packed 16 to a word in LDS;
  2. classification of the result (voxel.rs:310-320) and, for the ambiguous children whose trace
     decided something, ONE reverse sweep that prunes the tape (vm/data.rs:123-318): dead ops are
     dropped, decided choices alias or copy their surviving operand, registers are renumbered
     densely; every child writes its tape back to front into a slot of the parent's length.

Dispatch is threaded (`s_setpc_b64` into 128-byte handler slots); the tape is fetched through
the scalar cache, 4 ops per load, double buffered.  Emitted by gen_interp.py.

kernarg: { FhRenderState* S; u32 level; u32 big; u32 max_regs; u32 max_choices; u32 n_waves; u32 flags;
           u32 skip_regs; u32 skip_choices }   slots whose tape exceeds (max_regs, max_choices) or fits
         (skip_regs, skip_choices) are left to the launch with the matching LDS layout
         flags bit 0: phase probes; bit 2: forward pass only (results + exported choices, nothing else);
         bits 31:16: export mode: choice words per slot in S->chw (0: as in the LDS layout);
         bit 3: every OUTPUT stores its interval to FhSlot::tvals[output index][lane] (tape groups);
         bit 1: export mode for the long tapes of the pre-pass levels (choice words go
         to S->chw[big][slot][word][lane], pruned lanes are only marked, fh_prune1 sweeps them one per wave)
LDS    : regs [max_regs][64] x 8 B | choice words [(max_choices+15)/16][64] x 4 B | map [max_regs][64] x 1 B
Limits : <= 128 registers (pool of 4 x 32 bits), opcodes of the assembly set (no transcendental / modulo / rng)
"""
from gen_interp import Asm, OPS, UNSUPPORTED, HSTRIDE_LOG2  # noqa: F401

# ---- SGPRs -----------------------------------------------------------------------------------
S_KERNARG = "s[0:1]"
S_STATE = "s[4:5]"
S_LEVEL, S_BIG, S_MAXREGS, S_MAXCH = "s6", "s7", "s8", "s9"
S_ARENA = "s[10:11]"
S_ARENACAP = "s12"
S_SLOTS = "s[14:15]"
S_NSLOTS = "s16"
S_CHBASE, S_MAPBASE = "s17", "s18"
S_CUROFF = "s19"          # byte offset of eval_cur[big][level] in the state
S_SLOT = "s[20:21]"
S_OFF, S_LEN, S_RC, S_SLEVEL = "s24", "s25", "s26", "s27"   # slot header words 0..3
S_ACT = "s[22:23]"
S_SIGN, S_ABSM = "s28", "s29"
S_DECIDED = "s[30:31]"
S_CI = "s32"
S_NREGS, S_NCH = "s33", "s36"
S_PRUNE = "s[34:35]"
S_BASE = "s37"
S_MA, S_MB = "s[38:39]", "s[40:41]"
S_HBASE = "s[42:43]"
S_TAPE = "s[44:45]"
S_REM = "s46"
S_K = "s47"               # prune: index of the op being visited
S_QA, S_QB = 48, 56
S_W0, S_W1, S_CUR = "s64", "s65", "s[64:65]"
S_T0, S_OUT, S_A, S_T1 = "s66", "s67", "s68", "s69"
S_BATCH = "s70"
S_OP = "s71"
S_FETCH = "s[72:73]"
S_RET = "s[74:75]"
S_T64 = "s[76:77]"
S_LIVE, S_ALIAS, S_CIMM, S_KEEP = "s[78:79]", "s[80:81]", "s[82:83]", "s[84:85]"
S_PC = "s[86:87]"
S_SAVE = "s[88:89]"
S_M = [f"s[{90 + 2 * j}:{91 + 2 * j}]" for j in range(4)]
S_T2, S_T3 = "s98", "s99"
S_STAGED = "s13"
S_SKIPR, S_SKIPC = "s7", "s19"   # (s7 = `big` is dead after the prologue)
S_RR = "s3"
S_FLAGS = "s101"           # kernarg `flags`: bit 0 = probes, bit 1 = export (choices to HBM, no prune here)
S_CHW, S_CHWSLOT, S_CHWTMP = "s[78:79]", "s[80:81]", "s[82:83]"   # export mode (the prune registers are free then)
S_TV = "s[84:85]"          # tape groups (forward only): FhSlot::tvals
S_SI, S_NWG = "s2", "s100"   # slot index of this wave, waves in the launch

# ---- VGPRs -----------------------------------------------------------------------------------
V_LANE, V_L8, V_L4 = "v0", "v1", "v2"
VX, VY, VZ = ("v4", "v5"), ("v6", "v7"), ("v8", "v9")
AL, AH, BL, BH, RL, RH = "v10", "v11", "v12", "v13", "v14", "v15"
T = [f"v{16 + i}" for i in range(12)]      # v16..v27 scratch (also the div / sqrt sequences)
V_CW = "v28"
V_C = "v29"
V_AADDR, V_OADDR, V_TADDR = "v30", "v31", "v32"
V_RESL, V_RESH = "v34", "v35"
V_QNAN, V_SQRTC = "v36", "v37"
V_ONE, V_MONE = "v38", "v39"
P = ["v40", "v41", "v42", "v43"]            # free-register pool, 1 = free
V_HIGH = "v44"
V_DST = "v[46:47]"
V_COUNT, V_KEPT = "v48", "v49"
V_NO, V_MAV, V_NA, V_NB = "v50", "v51", "v52", "v53"
V_CWP = "v54"
V_EW0, V_EW1 = "v56", "v57"                 # emitted op (consecutive, 64-bit store)
V_U = [f"v{58 + i}" for i in range(6)]      # v58..v63 prune scratch
V_RANK = "v33"
V_COFF, V_CLEN, V_CRC = "v64", "v65", "v66"
V_DEAD = "v67"
V_ZERO = "v76"                              # v[68:75]: tape batch in flight
V_PW = "v[78:79]"                           # prune: prefetched op
V_AOUT, V_AA, V_AB, V_AAL = "v80", "v81", "v82", "v83"   # prune: LDS addresses of map[out], map[a], map[b], alias
V_MBV, V_AV = "v55", "v45"                  # prune: map[b], value of the aliased entry
N_VGPR = 84

SLOT_SIZE = 40 + 64 * 4 * 14
SL_ACT, SL_XYZ, SL_CORNER, SL_RES, SL_COFF, SL_CLEN, SL_CRC = 16, 40, 40 + 6 * 256, 40 + 9 * 256, 40 + 11 * 256, 40 + 12 * 256, 40 + 13 * 256


TRANS_UNARY = ("SIN", "COS", "TAN", "ASIN", "ACOS", "ATAN", "EXP", "LN")
T_SMAP = [78, 79, 80, 81, 82, 83, 84, 85, 88, 89]      # scalar registers of the compiled routines in the tile kernels: the prune's masks
                                                        # (dead in the forward pass) and S_SAVE; return address s[96:97] (= S_M[3])


class Tiles:
    def __init__(self, a, off, trans=None):
        self.a, self.off = a, off
        self.name = "fh_tiles"
        self.next = ".Lfh_tiles_next"
        self.ool = []
        # `trans` (the compiled routines' assembly): handlers for the transcendental opcodes (the *_t kernels).  The routines'
        # vector registers go behind the kernel's own (t_base), their labels get t_prefix.
        self.trans, self.t_base, self.t_prefix = trans, N_VGPR, "fh_til_"
        self.n_vgpr = N_VGPR + (26 if trans else 0)

    # ---- transcendental opcodes: interval rules of types/interval.rs:136-302 (dev_ops.hpp iv_sincos .. iv_ln) around the compiled
    # f32 routines (gen_trans.py), which are what the HIP kernels inline: same bounds, bit for bit --------------------------------
    def tcall(self, fn, arg, res):
        """res = <fn>(arg); clobbers the routines' window, s78..s85, s88, s89, s96, s97 and vcc"""
        here, ret = self.a.label("tcall"), self.a.label("tret")
        self.a(f"""
	v_mov_b32 v{self.t_base}, {arg}
	s_getpc_b64 s[96:97]
{here}:
	s_add_u32 s96, s96, {ret} - {here}
	s_addc_u32 s97, s97, 0
	s_branch {self.t_prefix}{fn}
{ret}:
	v_mov_b32 {res}, v{self.t_base}""")

    def quadrant(self, x, q):
        """q = iv_quadrant(x) as a float 0..3: rem_euclid(floorf(x * 2 / PI), 4), NaN -> 0 (dev_ops.hpp; interval.rs:143-147)"""
        a = self.a
        a(f"\tv_add_f32 {q}, {x}, {x}\n\tv_mov_b32 {T[10]}, 0x40490fdb")
        self.div(q, T[10], q, d=T[5:10])
        a(f"""
	v_floor_f32 {q}, {q}
	v_mul_f32 {T[10]}, 0.25, {q}
	v_trunc_f32 {T[10]}, {T[10]}
	v_fma_f32 {q}, -4.0, {T[10]}, {q}                     ; fmodf(q, 4): exact (q is an integer, 4 a power of two)
	v_add_f32 {T[10]}, 4.0, {q}
	v_cmp_gt_f32 vcc, 0, {q}
	s_nop 1
	v_cndmask_b32 {q}, {q}, {T[10]}, vcc
	v_cmp_gt_f32 vcc, {q}, 0                              ; !(q > 0): NaN and 0 -> 0
	s_nop 1
	v_cndmask_b32 {q}, 0, {q}, vcc""")

    def b_trans(self, op):
        a = self.a
        fn = op.lower()
        fl, fu, lq, uq, d = T[0], T[1], T[2], T[3], T[4]
        if op in ("EXP", "ATAN", "LN", "ASIN", "ACOS", "TAN"):
            self.tcall(fn, AL, fl)
            self.tcall(fn, AH, fu)
            if op == "ACOS":                      # decreasing
                a(f"\tv_mov_b32 {RL}, {fu}\n\tv_mov_b32 {RH}, {fl}")
            else:
                a(f"\tv_mov_b32 {RL}, {fl}\n\tv_mov_b32 {RH}, {fu}")
            if op == "LN":                        # a.lo <= 0 -> NaN
                a(f"\tv_cmp_ge_f32_e64 {S_M[0]}, 0, {AL}")
                self.nan_out(S_M[0])
            elif op in ("ASIN", "ACOS"):          # a.lo < -1 || a.hi > 1 -> NaN
                a(f"\tv_cmp_gt_f32_e64 {S_M[0]}, -1.0, {AL}\n\tv_cmp_gt_f32_e64 {S_M[1]}, {AH}, 1.0\n\ts_nop 0\n\ts_or_b64 {S_M[0]}, {S_M[0]}, {S_M[1]}")
                self.nan_out(S_M[0])
            elif op == "TAN":                     # hi - lo >= PI -> NaN; tan(hi) >= tan(lo) ? [l, u] : NaN
                a(f"""
	v_sub_f32 {d}, {AH}, {AL}
	v_mov_b32 {T[10]}, 0x40490fdb
	v_cmp_ge_f32_e64 {S_M[0]}, {d}, {T[10]}
	v_cmp_nge_f32_e64 {S_M[1]}, {fu}, {fl}
	s_nop 0
	s_or_b64 {S_M[0]}, {S_M[0]}, {S_M[1]}""")
                self.nan_out(S_M[0])
            return
        # SIN / COS (iv_sincos): the function at both ends, the quadrants of both ends, then the table
        self.tcall(fn, AL, fl)
        self.tcall(fn, AH, fu)
        self.quadrant(AL, lq)
        self.quadrant(AH, uq)
        if op == "COS":                           # cos(x) = sin(x + pi/2): quadrants rotated by one
            for q in (lq, uq):
                a(f"\tv_add_f32 {q}, 1.0, {q}\n\tv_cmp_eq_f32 vcc, 4.0, {q}\n\ts_nop 1\n\tv_cndmask_b32 {q}, {q}, 0, vcc")
        lqd, uqd, same, c30, c12, big = S_M[0], S_M[1], S_M[2], S_M[3], S_MA, S_MB
        a(f"""
	v_sub_f32 {d}, {AH}, {AL}
	v_min_f32 {T[5]}, {fl}, {fu}
	v_max_f32 {T[6]}, {fl}, {fu}
	v_mov_b32 {T[10]}, 0x40490fdb
	; decreasing quadrants: 1, 2  <=>  |q - 1.5| < 1
	v_add_f32 {T[7]}, -1.5, {lq}
	v_add_f32 {T[8]}, -1.5, {uq}
	v_cmp_lt_f32_e64 {lqd}, |{T[7]}|, 1.0
	v_cmp_lt_f32_e64 {uqd}, |{T[8]}|, 1.0
	v_cmp_eq_f32_e64 {same}, {lq}, {uq}
	v_cmp_gt_f32_e64 {c30}, {T[7]}, 1.0                  ; lq == 3  (lq - 1.5 = 1.5; 3.0 is no inline constant)
	v_cmp_eq_f32_e64 {S_T64}, 0, {uq}
	v_cmp_eq_f32_e64 {c12}, 1.0, {lq}
	v_cmp_eq_f32_e64 vcc, 2.0, {uq}
	v_cmp_ge_f32_e64 {big}, {d}, {T[10]}
	s_and_b64 {c30}, {c30}, {S_T64}
	s_and_b64 {c12}, {c12}, vcc
	; increasing: (same && !lq_dec) || (3 -> 0); decreasing: (same && lq_dec) || (1 -> 2); both only while hi - lo < PI
	s_andn2_b64 {S_T64}, {same}, {lqd}
	s_or_b64 {c30}, {c30}, {S_T64}
	s_and_b64 {S_T64}, {same}, {lqd}
	s_or_b64 {c12}, {c12}, {S_T64}
	s_andn2_b64 {c30}, {c30}, {big}
	s_andn2_b64 {c12}, {c12}, {big}
	; (0 | 3) -> (1 | 2): [min, 1];  (1 | 2) -> (3 | 0): [-1, max]
	s_andn2_b64 {same}, {uqd}, {lqd}
	s_andn2_b64 {big}, {lqd}, {uqd}
	v_mov_b32 {RL}, -1.0
	v_mov_b32 {RH}, 1.0""")
        self.sel(RL, RL, T[5], same)
        self.sel(RH, RH, T[6], big)
        self.sel(RL, RL, fl, c30)
        self.sel(RH, RH, fu, c30)
        self.sel(RL, RL, fu, c12)
        self.sel(RH, RH, fl, c12)
        a(f"""
	v_mov_b32 {T[10]}, 0x40c90fdb
	v_cmp_ge_f32_e64 {S_M[0]}, {d}, {T[10]}            ; hi - lo >= TAU: [-1, 1]
	v_cmp_u_f32_e64 {S_M[1]}, {AL}, {AH}
	s_nop 0""")
        self.sel(RL, RL, "-1.0", S_M[0])
        self.sel(RH, RH, "1.0", S_M[0])
        self.nan_out(S_M[1])



    # ---- rand, mix, modulo, atan2: interval rules of types/interval.rs:485-503, 541-627 (dev_ops.hpp iv_rand, iv_mix, iv_rem_euclid,
    # iv_atan2) - the *_t kernels only: modulo and atan2 CALL the compiled f32 routines (rem_euclid, atan2f), as the HIP kernels inline them ----
    def tcall2(self, fn, arg0, arg1, res):
        """res = <fn>(arg0, arg1); clobbers what tcall clobbers"""
        here, ret = self.a.label("tcall"), self.a.label("tret")
        self.a(f"""
	v_mov_b32 v{self.t_base}, {arg0}
	v_mov_b32 v{self.t_base + 1}, {arg1}
	s_getpc_b64 s[96:97]
{here}:
	s_add_u32 s96, s96, {ret} - {here}
	s_addc_u32 s97, s97, 0
	s_branch {self.t_prefix}{fn}
{ret}:
	v_mov_b32 {res}, v{self.t_base}""")

    def pcg(self, x, r, k):
        """r = rng::hash(x) (rng/mod.rs:8-13: the PCG output permutation); k = three scratch registers (r may be x)"""
        self.a(f"""
	v_mov_b32 {k[2]}, 747796405
	v_mul_lo_u32 {k[0]}, {x}, {k[2]}
	v_add_u32 {k[0]}, 0xac564b05, {k[0]}
	v_lshrrev_b32 {k[1]}, 28, {k[0]}
	v_add_u32 {k[1]}, 4, {k[1]}
	v_lshrrev_b32 {k[1]}, {k[1]}, {k[0]}
	v_xor_b32 {k[1]}, {k[1]}, {k[0]}
	v_mov_b32 {k[2]}, 277803737
	v_mul_lo_u32 {k[1]}, {k[1]}, {k[2]}
	v_lshrrev_b32 {k[0]}, 22, {k[1]}
	v_xor_b32 {r}, {k[0]}, {k[1]}""")

    def b_rand(self):
        """NaN or not one bit pattern -> [0, 1]; else the point rng::rand (rng/mod.rs:19-23): bits (hash >> 9) | 1.0, minus 1"""
        a = self.a
        a(f"\tv_cmp_u_f32_e64 {S_M[0]}, {AL}, {AH}\n\tv_cmp_ne_u32_e64 {S_M[1]}, {AL}, {AH}")
        self.pcg(AL, T[0], T[1:4])
        a(f"""
	v_lshrrev_b32 {T[0]}, 9, {T[0]}
	v_or_b32 {T[0]}, 0x3f800000, {T[0]}
	v_add_f32 {T[0]}, -1.0, {T[0]}
	s_or_b64 {S_M[0]}, {S_M[0]}, {S_M[1]}
	s_nop 1""")
        self.sel(RL, T[0], "0", S_M[0])
        self.sel(RH, T[0], "1.0", S_M[0])

    def b_mix(self):
        """both operands one non-NaN bit pattern -> the point rng::mix (rng/mod.rs:30-33): hash(a + hash(b)); else NaN"""
        a = self.a
        a(f"""
	v_cmp_u_f32_e64 {S_M[0]}, {AL}, {AH}
	v_cmp_u_f32_e64 {S_M[1]}, {BL}, {BH}
	v_cmp_ne_u32_e64 {S_M[2]}, {AL}, {AH}
	v_cmp_ne_u32_e64 {S_T64}, {BL}, {BH}
	s_or_b64 {S_M[0]}, {S_M[0]}, {S_M[1]}
	s_or_b64 {S_M[2]}, {S_M[2]}, {S_T64}
	s_or_b64 {S_M[0]}, {S_M[0]}, {S_M[2]}""")
        self.pcg(BL, T[0], T[1:4])
        a(f"\tv_add_u32 {T[0]}, {AL}, {T[0]}")
        self.pcg(T[0], T[0], T[1:4])
        a(f"\tv_mov_b32 {RL}, {T[0]}\n\tv_mov_b32 {RH}, {T[0]}")
        self.nan_out(S_M[0])

    def b_mod(self):
        """NaN or a divisor that contains 0 -> NaN; a point divisor > 0 and a.lo / d not an integer with the same floor as a.hi / d ->
        [rem_euclid(a.lo, d), rem_euclid(a.hi, d)]; else [0, |divisor|.hi]"""
        a = self.a
        x, y, fx, fy, r1, r2 = T[0], T[1], T[2], T[3], T[10], T[11]
        self.div(AL, BL, x, d=T[4:10])
        self.div(AH, BL, y, d=T[4:10])
        a(f"""
	v_floor_f32 {fx}, {x}
	v_floor_f32 {fy}, {y}
	v_cmp_eq_f32_e64 {S_M[0]}, {BL}, {BH}
	v_cmp_lt_f32_e64 {S_M[1]}, 0, {BL}
	v_cmp_neq_f32_e64 {S_M[2]}, {x}, {fx}
	v_cmp_eq_f32_e64 {S_T64}, {fx}, {fy}
	s_and_b64 {S_M[0]}, {S_M[0]}, {S_M[1]}
	s_and_b64 {S_M[2]}, {S_M[2]}, {S_T64}
	s_and_b64 {S_M[2]}, {S_M[2]}, {S_M[0]}                    ; same: the remainders of both ends bound the result""")
        self.tcall2("mod", AL, BL, r1)
        self.tcall2("mod", AH, BL, r2)
        a(f"""
	v_cmp_gt_f32_e64 {S_M[0]}, 0, {BL}
	s_nop 1
	v_cndmask_b32_e64 {T[4]}, {BH}, -{BL}, {S_M[0]}          ; |divisor|.hi for a divisor that does not contain 0
	v_cmp_u_f32_e64 {S_M[0]}, {AL}, {AH}
	v_cmp_u_f32_e64 {S_M[1]}, {BL}, {BH}
	v_cmp_ge_f32_e64 {S_T64}, 0, {BL}
	s_or_b64 {S_M[0]}, {S_M[0]}, {S_M[1]}
	v_cmp_le_f32_e64 {S_M[1]}, 0, {BH}
	s_nop 0
	s_and_b64 {S_M[1]}, {S_M[1]}, {S_T64}
	s_or_b64 {S_M[0]}, {S_M[0]}, {S_M[1]}""")
        self.sel(RL, "0", r1, S_M[2])
        self.sel(RH, T[4], r2, S_M[2])
        self.nan_out(S_M[0])

    def b_atan2(self):
        """A = y, B = x.  NaN -> NaN; y contains 0 and x.lo < 0 -> [-pi, pi]; else atan2f at two corners chosen by the signs of y and x
        (interval.rs:541-597), the smaller and the larger of the two"""
        a = self.a
        y0, y1, x1, v0, v1 = T[0], T[1], T[2], T[3], T[4]
        ypos, yneg, xpos, xneg = S_M[0], S_M[1], S_M[2], S_T64
        a(f"""
	v_cmp_le_f32_e64 {ypos}, 0, {AL}                        ; y.lo >= 0
	v_cmp_ge_f32_e64 {yneg}, 0, {AH}                        ; y.hi <= 0
	v_cmp_le_f32_e64 {xpos}, 0, {BL}                        ; x.lo >= 0
	v_cmp_ge_f32_e64 {xneg}, 0, {BH}                        ; x.hi <= 0
	s_andn2_b64 {yneg}, {yneg}, {ypos}
	s_andn2_b64 {xneg}, {xneg}, {xpos}
	s_nop 1
	; y0: y >= 0: x >= 0 ? y.hi : y.lo;  y <= 0: x >= 0 ? y.lo : y.hi;  y mixed: y.lo
	v_cndmask_b32_e64 {T[5]}, {AL}, {AH}, {xpos}            ; x >= 0 ? y.hi : y.lo
	v_cndmask_b32_e64 {T[6]}, {AH}, {AL}, {xpos}            ; x >= 0 ? y.lo : y.hi
	v_mov_b32 {y0}, {AL}
	v_cndmask_b32_e64 {y0}, {y0}, {T[6]}, {yneg}
	v_cndmask_b32_e64 {y0}, {y0}, {T[5]}, {ypos}
	; y1: y >= 0: x <= 0 (strictly mixed excluded) ? y.hi : y.lo;  y <= 0: x <= 0 ? y.lo : y.hi;  y mixed: y.hi
	v_cndmask_b32_e64 {T[5]}, {AL}, {AH}, {xneg}            ; x <= 0 ? y.hi : y.lo
	v_cndmask_b32_e64 {T[6]}, {AH}, {AL}, {xneg}            ; x <= 0 ? y.lo : y.hi
	v_mov_b32 {y1}, {AH}
	v_cndmask_b32_e64 {y1}, {y1}, {T[6]}, {yneg}
	v_cndmask_b32_e64 {y1}, {y1}, {T[5]}, {ypos}
	; x1: x.hi, but x.lo when y is mixed
	s_or_b64 {xpos}, {ypos}, {yneg}
	s_nop 1
	v_cndmask_b32_e64 {x1}, {BL}, {BH}, {xpos}""")
        self.tcall2("atan2", y0, BL, v0)
        self.tcall2("atan2", y1, x1, v1)
        a(f"""
	v_min_f32 {RL}, {v0}, {v1}
	v_max_f32 {RH}, {v0}, {v1}
	v_cmp_ge_f32_e64 {S_M[0]}, 0, {AL}                      ; y.lo <= 0 && y.hi >= 0 && x.lo < 0: the whole circle
	v_cmp_le_f32_e64 {S_M[1]}, 0, {AH}
	v_cmp_gt_f32_e64 {S_M[2]}, 0, {BL}
	v_mov_b32 {T[5]}, 0x40490fdb
	s_and_b64 {S_M[0]}, {S_M[0]}, {S_M[1]}
	s_and_b64 {S_M[0]}, {S_M[0]}, {S_M[2]}
	v_cmp_u_f32_e64 {S_M[1]}, {AL}, {AH}
	v_cmp_u_f32_e64 {S_M[2]}, {BL}, {BH}
	s_nop 0""")
        self.sel(RL, RL, f"-{T[5]}", S_M[0])
        self.sel(RH, RH, T[5], S_M[0])
        a(f"\ts_or_b64 {S_M[1]}, {S_M[1]}, {S_M[2]}")
        self.nan_out(S_M[1])

    # ---- helpers -------------------------------------------------------------------------
    def done(self):
        """store R to the out register, next op"""
        self.a(f"\tds_write_b64 {V_OADDR}, v[14:15]\n\ts_branch {self.next}")

    def load_b(self):
        self.a(f"""
	s_lshl_b32 {S_T1}, {S_W1}, 9
	v_add_u32 {V_TADDR}, {S_T1}, {V_L8}
	ds_read_b64 v[12:13], {V_TADDR}
	s_waitcnt lgkmcnt(0)""")

    def imm_b(self):
        self.a(f"\tv_mov_b32 {BL}, {S_W1}\n\tv_mov_b32 {BH}, {S_W1}\n\ts_waitcnt lgkmcnt(0)")

    def imm_a_swap(self):
        """imm,reg forms: lhs = imm, rhs = the register operand (loaded into A by the dispatcher)"""
        self.a(f"""
	s_waitcnt lgkmcnt(0)
	v_mov_b32 {BL}, {AL}
	v_mov_b32 {BH}, {AH}
	v_mov_b32 {AL}, {S_W1}
	v_mov_b32 {AH}, {S_W1}""")

    def nan_mask(self, dst, lo, hi):
        self.a(f"\tv_cmp_u_f32_e64 {dst}, {lo}, {hi}")

    def div(self, num, den, out, d=T):
        """IEEE-correct num / den (the compiler's own expansion)"""
        self.a(f"""
	v_div_scale_f32 {d[0]}, {S_T64}, {den}, {den}, {num}
	v_rcp_f32 {d[1]}, {d[0]}
	v_div_scale_f32 {d[2]}, vcc, {num}, {den}, {num}
	v_fma_f32 {d[3]}, -{d[0]}, {d[1]}, 1.0
	v_fmac_f32 {d[1]}, {d[3]}, {d[1]}
	v_mul_f32 {d[3]}, {d[2]}, {d[1]}
	v_fma_f32 {d[4]}, -{d[0]}, {d[3]}, {d[2]}
	v_fmac_f32 {d[3]}, {d[4]}, {d[1]}
	v_fma_f32 {d[0]}, -{d[0]}, {d[3]}, {d[2]}
	v_div_fmas_f32 {d[0]}, {d[0]}, {d[1]}, {d[3]}
	v_div_fixup_f32 {out}, {d[0]}, {den}, {num}""")

    def sqrt(self, x, out, d=T):
        self.a(f"""
	v_mul_f32 {d[0]}, 0x4f800000, {x}
	v_cmp_gt_f32 vcc, {V_SQRTC}, {x}
	s_nop 1
	v_cndmask_b32 {d[1]}, {x}, {d[0]}, vcc
	v_sqrt_f32 {d[0]}, {d[1]}
	s_nop 0
	v_add_u32 {d[2]}, -1, {d[0]}
	v_add_u32 {d[3]}, 1, {d[0]}
	v_fma_f32 {d[4]}, -{d[2]}, {d[0]}, {d[1]}
	v_fma_f32 {d[5]}, -{d[3]}, {d[0]}, {d[1]}
	v_cmp_ge_f32_e64 {S_M[3]}, 0, {d[4]}
	s_nop 1
	v_cndmask_b32_e64 {d[0]}, {d[0]}, {d[2]}, {S_M[3]}
	v_cmp_lt_f32_e64 {S_M[3]}, 0, {d[5]}
	s_nop 1
	v_cndmask_b32_e64 {d[0]}, {d[0]}, {d[3]}, {S_M[3]}
	v_mul_f32 {d[2]}, 0x37800000, {d[0]}
	v_cndmask_b32 {d[0]}, {d[0]}, {d[2]}, vcc
	v_mov_b32 {d[3]}, 0x260
	v_cmp_class_f32 vcc, {d[1]}, {d[3]}
	s_nop 1
	v_cndmask_b32 {out}, {d[0]}, {d[1]}, vcc""")

    def round(self, x, out, d=T):
        self.a(f"""
	v_trunc_f32 {d[0]}, {x}
	v_sub_f32 {d[1]}, {x}, {d[0]}
	v_cmp_ge_f32_e64 {S_M[3]}, |{d[1]}|, 0.5
	s_nop 1
	v_cndmask_b32_e64 {d[1]}, 0, 1.0, {S_M[3]}
	v_bfi_b32 {d[1]}, {S_ABSM}, {d[1]}, {x}
	v_add_f32 {out}, {d[0]}, {d[1]}""")

    def sel(self, dst, if_false, if_true, mask):
        """dst = mask ? if_true : if_false; the mask must have been written >= 2 wait states ago"""
        self.a(f"\tv_cndmask_b32_e64 {dst}, {if_false}, {if_true}, {mask}")

    def nan_out(self, mask):
        self.a(f"\ts_nop 1")
        self.sel(RL, RL, V_QNAN, mask)
        self.sel(RH, RH, V_QNAN, mask)

    # ---- bodies (A, B plain; result in R) ------------------------------------------------------
    def b_neg(self):
        self.a(f"\tv_xor_b32 {RL}, {S_SIGN}, {AH}\n\tv_xor_b32 {RH}, {S_SIGN}, {AL}")

    def b_abs(self):
        a = self.a
        a(f"""
	v_cmp_gt_f32_e64 {S_M[0]}, 0, {AL}
	v_cmp_gt_f32_e64 {S_M[1]}, {AH}, 0
	v_xor_b32 {T[0]}, {S_SIGN}, {AL}
	v_xor_b32 {T[1]}, {S_SIGN}, {AH}
	v_max_f32 {T[2]}, {AH}, {T[0]}""")
        self.sel(T[3], T[1], 0, S_M[1])
        self.sel(T[4], T[0], T[2], S_M[1])
        self.sel(RL, AL, T[3], S_M[0])
        self.sel(RH, AH, T[4], S_M[0])

    def b_recip(self):
        a = self.a
        self.div("1.0", AH, RL)
        self.div("1.0", AL, RH)
        a(f"""
	v_cmp_gt_f32_e64 {S_M[0]}, {AL}, 0
	v_cmp_gt_f32_e64 {S_M[1]}, 0, {AH}
	s_or_b64 {S_M[0]}, {S_M[0]}, {S_M[1]}
	s_nop 1""")
        self.sel(RL, V_QNAN, RL, S_M[0])
        self.sel(RH, V_QNAN, RH, S_M[0])

    def b_sqrt(self):
        self.sqrt(AL, RL)
        self.sqrt(AH, RH)
        self.a(f"\tv_cmp_gt_f32_e64 {S_M[0]}, 0, {AL}")
        self.nan_out(S_M[0])

    def b_square(self):
        a = self.a
        a(f"""
	v_mul_f32 {T[0]}, {AL}, {AL}
	v_mul_f32 {T[1]}, {AH}, {AH}
	v_max_f32_e64 {T[2]}, |{AL}|, |{AH}|
	v_mul_f32 {T[2]}, {T[2]}, {T[2]}
	v_cmp_gt_f32_e64 {S_M[0]}, 0, {AH}
	v_cmp_gt_f32_e64 {S_M[1]}, {AL}, 0
	v_cmp_u_f32_e64 {S_M[2]}, {AL}, {AH}
	v_mov_b32 {RL}, 0
	v_mov_b32 {RH}, {T[2]}""")
        self.sel(RL, RL, V_QNAN, S_M[2])
        self.sel(RH, RH, V_QNAN, S_M[2])
        self.sel(RL, RL, T[0], S_M[1])
        self.sel(RH, RH, T[1], S_M[1])
        self.sel(RL, RL, T[1], S_M[0])
        self.sel(RH, RH, T[0], S_M[0])

    def b_not(self):
        a = self.a
        a(f"""
	v_cmp_ge_f32_e64 {S_M[0]}, 0, {AL}
	v_cmp_ge_f32_e64 {S_M[1]}, {AH}, 0
	v_cmp_u_f32_e64 {S_M[2]}, {AL}, {AH}
	v_cmp_eq_f32_e64 {S_M[3]}, 0, {AL}
	v_cmp_eq_f32_e64 {S_MA}, 0, {AH}
	s_and_b64 {S_M[0]}, {S_M[0]}, {S_M[1]}
	s_or_b64 {S_M[0]}, {S_M[0]}, {S_M[2]}
	s_and_b64 {S_M[3]}, {S_M[3]}, {S_MA}
	s_nop 1""")
        # M0 = contains zero or NaN -> upper bound 1 ; M3 = exactly zero -> lower bound 1
        self.sel(RL, 0, 1.0, S_M[3])
        self.sel(RH, 0, 1.0, S_M[0])

    def b_add(self):
        self.a(f"\tv_add_f32 {RL}, {AL}, {BL}\n\tv_add_f32 {RH}, {AH}, {BH}")

    def b_sub(self):
        self.a(f"\tv_sub_f32 {RL}, {AL}, {BH}\n\tv_sub_f32 {RH}, {AH}, {BL}")

    def minmax4(self, p):
        a = self.a
        a(f"""
	v_min_f32 {RL}, {p[0]}, {p[1]}
	v_max_f32 {RH}, {p[0]}, {p[1]}
	v_min_f32 {RL}, {RL}, {p[2]}
	v_max_f32 {RH}, {RH}, {p[2]}
	v_min_f32 {RL}, {RL}, {p[3]}
	v_max_f32 {RH}, {RH}, {p[3]}""")

    def b_mul(self):
        a = self.a
        a(f"""
	v_cmp_u_f32_e64 {S_M[0]}, {AL}, {AH}
	v_cmp_u_f32_e64 {S_M[1]}, {BL}, {BH}
	v_mul_f32 {T[0]}, {AL}, {BL}
	v_mul_f32 {T[1]}, {AL}, {BH}
	v_mul_f32 {T[2]}, {AH}, {BL}
	v_mul_f32 {T[3]}, {AH}, {BH}
	s_or_b64 {S_M[0]}, {S_M[0]}, {S_M[1]}""")
        self.minmax4(T)
        self.nan_out(S_M[0])

    def b_mul_imm(self):
        a = self.a
        a(f"""
	v_cmp_u_f32_e64 {S_M[0]}, {AL}, {AH}
	v_cmp_u_f32_e64 {S_M[1]}, {BL}, {BL}
	v_cmp_gt_f32_e64 {S_M[2]}, 0, {BL}
	v_mul_f32 {T[0]}, {AL}, {BL}
	v_mul_f32 {T[1]}, {AH}, {BL}
	s_or_b64 {S_M[0]}, {S_M[0]}, {S_M[1]}""")
        self.sel(RL, T[0], T[1], S_M[2])
        self.sel(RH, T[1], T[0], S_M[2])
        self.nan_out(S_M[0])

    def b_div(self):
        a = self.a
        q = [T[8], T[9], T[10], T[11]]
        self.div(AL, BL, q[0])
        self.div(AL, BH, q[1])
        self.div(AH, BL, q[2])
        self.div(AH, BH, q[3])
        self.minmax4(q)
        # NaN unless the divisor is strictly signed and the dividend is a number
        a(f"""
	v_cmp_gt_f32_e64 {S_M[0]}, {BL}, 0
	v_cmp_gt_f32_e64 {S_M[1]}, 0, {BH}
	v_cmp_u_f32_e64 {S_M[2]}, {AL}, {AH}
	s_or_b64 {S_M[0]}, {S_M[0]}, {S_M[1]}
	s_andn2_b64 {S_M[0]}, {S_M[0]}, {S_M[2]}
	s_nop 1""")
        self.sel(RL, V_QNAN, RL, S_M[0])
        self.sel(RH, V_QNAN, RH, S_M[0])

    def b_compare(self):
        a = self.a
        a(f"""
	v_cmp_u_f32_e64 {S_M[0]}, {AL}, {AH}
	v_cmp_u_f32_e64 {S_M[1]}, {BL}, {BH}
	v_cmp_lt_f32_e64 {S_M[2]}, {AH}, {BL}
	v_cmp_gt_f32_e64 {S_M[3]}, {AL}, {BH}
	s_or_b64 {S_M[0]}, {S_M[0]}, {S_M[1]}
	v_cmp_eq_f32_e64 {S_M[1]}, {AL}, {AH}
	v_cmp_eq_f32_e64 {S_MA}, {BL}, {BH}
	v_cmp_eq_f32_e64 {S_MB}, {AL}, {BL}
	v_mov_b32 {RL}, -1.0
	v_mov_b32 {RH}, 1.0
	s_and_b64 {S_M[1]}, {S_M[1]}, {S_MA}
	s_and_b64 {S_M[1]}, {S_M[1]}, {S_MB}
	s_nop 0""")
        self.sel(RL, RL, 0, S_M[1])       # both points and equal -> [0, 0]
        self.sel(RH, RH, 0, S_M[1])
        self.sel(RL, RL, 1.0, S_M[3])     # strictly greater -> [1, 1]
        self.sel(RH, RH, 1.0, S_M[3])
        self.sel(RL, RL, -1.0, S_M[2])    # strictly less -> [-1, -1]
        self.sel(RH, RH, -1.0, S_M[2])
        self.sel(RL, RL, V_QNAN, S_M[0])
        self.sel(RH, RH, V_QNAN, S_M[0])

    def nan_ab(self):
        self.a(f"""
	v_cmp_u_f32_e64 {S_M[0]}, {AL}, {AH}
	v_cmp_u_f32_e64 {S_M[1]}, {BL}, {BH}
	s_or_b64 {S_M[0]}, {S_M[0]}, {S_M[1]}""")

    def b_minmax(self, is_min):
        a = self.a
        self.nan_ab()
        if is_min:   # Left: a.hi < b.lo ; Right: b.hi < a.lo
            a(f"\tv_cmp_lt_f32_e64 {S_M[1]}, {AH}, {BL}\n\tv_cmp_lt_f32_e64 {S_M[2]}, {BH}, {AL}")
            a(f"\tv_min_f32 {RL}, {AL}, {BL}\n\tv_min_f32 {RH}, {AH}, {BH}")
        else:        # Left: a.lo > b.hi ; Right: b.lo > a.hi
            a(f"\tv_cmp_gt_f32_e64 {S_M[1]}, {AL}, {BH}\n\tv_cmp_gt_f32_e64 {S_M[2]}, {BL}, {AH}")
            a(f"\tv_max_f32 {RL}, {AL}, {BL}\n\tv_max_f32 {RH}, {AH}, {BH}")
        a(f"\tv_mov_b32 {V_C}, 3")
        self.sel(V_C, V_C, 2, S_M[2])
        self.sel(V_C, V_C, 1, S_M[1])
        self.sel(V_C, V_C, 3, S_M[0])
        self.sel(RL, RL, V_QNAN, S_M[0])
        self.sel(RH, RH, V_QNAN, S_M[0])

    def b_andor(self, is_and):
        a = self.a
        self.nan_ab()
        # Z = a is exactly [0,0]; N = a does not contain 0
        a(f"""
	v_cmp_eq_f32_e64 {S_M[1]}, 0, {AL}
	v_cmp_eq_f32_e64 {S_M[2]}, 0, {AH}
	v_cmp_ge_f32_e64 {S_M[3]}, 0, {AL}
	v_cmp_ge_f32_e64 {S_MA}, {AH}, 0
	s_and_b64 {S_M[1]}, {S_M[1]}, {S_M[2]}
	s_and_b64 {S_M[3]}, {S_M[3]}, {S_MA}
	v_mov_b32 {V_C}, 3""")
        # M1 = Z, M3 = contains zero
        if is_and:
            a(f"\tv_min_f32 {RL}, {BL}, 0\n\tv_max_f32 {RH}, {BH}, 0")
            a(f"\ts_nop 0")
            self.sel(RL, BL, RL, S_M[3])      # !contains -> b, Right
            self.sel(RH, BH, RH, S_M[3])
            self.sel(V_C, 2, V_C, S_M[3])
            self.sel(RL, RL, 0, S_M[1])       # zero -> [0,0], Left
            self.sel(RH, RH, 0, S_M[1])
            self.sel(V_C, V_C, 1, S_M[1])
        else:
            a(f"\tv_min_f32 {RL}, {AL}, {BL}\n\tv_max_f32 {RH}, {AH}, {BH}")
            a(f"\ts_nop 0")
            self.sel(RL, RL, BL, S_M[1])      # zero -> b, Right
            self.sel(RH, RH, BH, S_M[1])
            self.sel(V_C, V_C, 2, S_M[1])
            self.sel(RL, AL, RL, S_M[3])      # !contains -> a, Left
            self.sel(RH, AH, RH, S_M[3])
            self.sel(V_C, 1, V_C, S_M[3])
        self.sel(V_C, V_C, 3, S_M[0])
        self.sel(RL, RL, V_QNAN, S_M[0])
        self.sel(RH, RH, V_QNAN, S_M[0])

    # ---- handler table ---------------------------------------------------------------------
    def ool_body(self, stem, fn):
        lab = f".Lfh_tiles_b_{stem}"
        if not any(l == lab for l, _ in self.ool):
            self.ool.append((lab, fn))
        self.a(f"\ts_branch {lab}")

    def handler(self, op):
        a = self.a
        if op == "OUTPUT":
            # flags bit 3 (tape groups): the tape's outputs are terms of the root tree, word 1 = term index;
            # their intervals go to the block's tvals[term][lane]
            a(f"""
	s_waitcnt lgkmcnt(0)
	v_mov_b32 {V_RESL}, {AL}
	v_mov_b32 {V_RESH}, {AH}
	s_bitcmp1_b32 {S_FLAGS}, 3
	s_cbranch_scc0 {self.next}
	s_lshl_b32 {S_T1}, {S_W1}, 9
	v_add_u32 {V_TADDR}, {S_T1}, {V_L8}
	global_store_dwordx2 {V_TADDR}, v[10:11], {S_TV}
	s_branch {self.next}""")
            return
        if op == "INPUT":
            return self.ool_body("input", self.h_input)
        if op == "COPY_REG":
            a(f"\ts_waitcnt lgkmcnt(0)\n\tds_write_b64 {V_OADDR}, v[10:11]\n\ts_branch {self.next}")
            return
        if op == "COPY_IMM":
            a(f"\tv_mov_b32 {RL}, {S_W1}\n\tv_mov_b32 {RH}, {S_W1}")
            return self.done()
        unary = {"NEG": self.b_neg, "ABS": self.b_abs, "RECIP": self.b_recip, "SQRT": self.b_sqrt, "SQUARE": self.b_square,
                 "NOT": self.b_not}
        if op in unary:
            def body(fn=unary[op]):
                fn()
                self.done()
            a("\ts_waitcnt lgkmcnt(0)")
            return self.ool_body(op.lower(), body)
        if op in TRANS_UNARY:
            def body(op=op):
                self.b_trans(op)
                self.done()
            a("\ts_waitcnt lgkmcnt(0)")
            return self.ool_body(op.lower(), body)
        if op == "RAND":
            def body():
                self.b_rand()
                self.done()
            a("\ts_waitcnt lgkmcnt(0)")
            return self.ool_body("rand", body)
        if op in ("FLOOR", "CEIL"):
            ins = "v_floor_f32" if op == "FLOOR" else "v_ceil_f32"
            a(f"\ts_waitcnt lgkmcnt(0)\n\t{ins} {RL}, {AL}\n\t{ins} {RH}, {AH}")
            return self.done()
        if op == "ROUND":
            def body():
                self.round(AL, RL)
                self.round(AH, RH)
                self.done()
            a("\ts_waitcnt lgkmcnt(0)")
            return self.ool_body("round", body)
        base, form = op.rsplit("_", 1)
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
            a(f"\tds_write_b64 {V_OADDR}, v[14:15]\n\ts_branch .Lfh_tiles_choice")
        return self.ool_body(base.lower(), cbody)

    def h_input(self):
        a = self.a
        lx, ly, lz = a.label("tin_x"), a.label("tin_y"), a.label("tin_z")
        a(f"""
	s_lshl_b32 {S_T0}, {S_W1}, 2
	s_add_u32 s76, s4, {S_T0}
	s_addc_u32 s77, s5, 0
	s_load_dword {S_T0}, {S_T64}, {self.off['P.in_kind']}
	s_load_dword {S_T1}, {S_T64}, {self.off['P.in_value']}
	s_waitcnt lgkmcnt(0)
	s_cmp_eq_u32 {S_T0}, 0
	s_cbranch_scc1 {lx}
	s_cmp_eq_u32 {S_T0}, 1
	s_cbranch_scc1 {ly}
	s_cmp_eq_u32 {S_T0}, 2
	s_cbranch_scc1 {lz}
	v_mov_b32 {RL}, {S_T1}
	v_mov_b32 {RH}, {S_T1}""")
        self.done()
        for lab, (lo, hi) in ((lx, VX), (ly, VY), (lz, VZ)):
            a(f"{lab}:\n\tv_mov_b32 {RL}, {lo}\n\tv_mov_b32 {RH}, {hi}")
            self.done()

    # ---- forward interpreter ---------------------------------------------------------------------
    def emit_forward(self):
        a = self.a
        qa, qb = S_QA, S_QB
        a(f"""
; ---- forward interval interpreter: {S_TAPE} = first op, {S_REM} = ops; returns to {S_RET} ----
.Lfh_tiles_run:
	; the tape comes through the vector memory path (vmcnt): the scalar path shares lgkmcnt with
	; the LDS register file, and every handler's LDS wait would also wait for the prefetch
	global_load_dwordx4 v[68:71], {V_ZERO}, {S_TAPE}
	global_load_dwordx4 v[72:75], {V_ZERO}, {S_TAPE} offset:16
	s_add_u32 s72, s44, 0x20
	s_addc_u32 s73, s45, 0
	s_mov_b32 {S_BATCH}, 4
	s_waitcnt vmcnt(0)
""" + "".join(f"\tv_readfirstlane_b32 s{qa + i}, v{68 + i}\n" for i in range(8)) + f"""
	global_load_dwordx4 v[68:71], {V_ZERO}, {S_FETCH}
	global_load_dwordx4 v[72:75], {V_ZERO}, {S_FETCH} offset:16
	s_add_u32 s72, s72, 0x20
	s_addc_u32 s73, s73, 0
{self.next}:
	s_sub_u32 {S_REM}, {S_REM}, 1
	s_cbranch_scc1 .Lfh_tiles_done
	s_sub_u32 {S_BATCH}, {S_BATCH}, 1
	s_cbranch_scc1 .Lfh_tiles_refill
.Lfh_tiles_decode:
	s_mov_b64 {S_CUR}, s[{qa}:{qa + 1}]
	s_mov_b64 s[{qa}:{qa + 1}], s[{qa + 2}:{qa + 3}]
	s_mov_b64 s[{qa + 2}:{qa + 3}], s[{qa + 4}:{qa + 5}]
	s_mov_b64 s[{qa + 4}:{qa + 5}], s[{qa + 6}:{qa + 7}]
	s_lshr_b32 {S_A}, {S_W0}, 20
	s_lshl_b32 {S_T1}, {S_A}, 9
	v_add_u32 {V_AADDR}, {S_T1}, {V_L8}
	ds_read_b64 v[10:11], {V_AADDR}
	s_lshl_b32 {S_T0}, {S_W0}, {HSTRIDE_LOG2}
	s_and_b32 {S_T0}, {S_T0}, {hex(0xff << HSTRIDE_LOG2)}
	s_add_u32 s86, s42, {S_T0}
	s_addc_u32 s87, s43, 0
	s_bfe_u32 {S_OUT}, {S_W0}, 0xc0008
	s_lshl_b32 {S_T1}, {S_OUT}, 9
	v_add_u32 {V_OADDR}, {S_T1}, {V_L8}
	s_setpc_b64 {S_PC}
.Lfh_tiles_refill:
	s_waitcnt vmcnt(0)
""" + "".join(f"\tv_readfirstlane_b32 s{qa + i}, v{68 + i}\n" for i in range(8)) + f"""
	global_load_dwordx4 v[68:71], {V_ZERO}, {S_FETCH}
	global_load_dwordx4 v[72:75], {V_ZERO}, {S_FETCH} offset:16
	s_add_u32 s72, s72, 0x20
	s_addc_u32 s73, s73, 0
	s_mov_b32 {S_BATCH}, 3
	s_branch .Lfh_tiles_decode
.Lfh_tiles_done:
	s_waitcnt vmcnt(0) lgkmcnt(0)
	s_setpc_b64 {S_RET}
; ---- record the choice in {V_C} (2 bits, 16 to a word, word to LDS when full) -------------
.Lfh_tiles_choice:
	v_cmp_ne_u32_e64 {S_MA}, 3, {V_C}
	s_and_b32 {S_T0}, {S_CI}, 15
	s_lshl_b32 {S_T1}, {S_T0}, 1
	v_lshl_or_b32 {V_CW}, {V_C}, {S_T1}, {V_CW}
	s_add_u32 {S_CI}, {S_CI}, 1
	s_or_b64 {S_DECIDED}, {S_DECIDED}, {S_MA}
	s_cmp_eq_u32 {S_T0}, 15
	s_cbranch_scc0 {self.next}
	s_lshr_b32 {S_T0}, {S_CI}, 4
	s_sub_u32 {S_T0}, {S_T0}, 1
	s_lshl_b32 {S_T0}, {S_T0}, 8
	s_bitcmp1_b32 {S_FLAGS}, 1
	s_cbranch_scc1 .Lfh_tiles_choice_export
	s_add_u32 {S_T0}, {S_T0}, {S_CHBASE}
	v_add_u32 {V_TADDR}, {S_T0}, {V_L4}
	ds_write_b32 {V_TADDR}, {V_CW}
	v_mov_b32 {V_CW}, 0
	s_branch {self.next}
.Lfh_tiles_choice_export:
	s_add_u32 s82, s80, {S_T0}
	s_addc_u32 s83, s81, 0
	global_store_dword {V_L4}, {V_CW}, {S_CHWTMP}
	s_nop 0
	v_mov_b32 {V_CW}, 0
	s_branch {self.next}
	.p2align {HSTRIDE_LOG2}
.Lfh_tiles_handlers:""")
        for i, op in enumerate(OPS):
            a(f"\t.p2align {HSTRIDE_LOG2}")
            a(f".Lfh_tiles_h{i}:  ; {op}")
            base = op.rsplit("_", 1)[0] if "_" in op and op not in ("COPY_REG", "COPY_IMM") else op
            if base in UNSUPPORTED and not self.trans:
                a(f"\ts_branch {self.next}")
            else:
                self.handler(op)
            a(f"\t.if (. - .Lfh_tiles_h{i}) > {1 << HSTRIDE_LOG2}\n\t.error \"tile handler {op} exceeds its slot\"\n\t.endif")
        a(f"\t.p2align {HSTRIDE_LOG2}")
        for lab, fn in self.ool:
            a(f"{lab}:")
            fn()

    # ---- pool of free registers (W x 32 bits, 1 = free) ------------------------------------------
    def pool_take(self, out, W):
        """out = lowest free register, marked used; for the lanes in exec"""
        a = self.a
        u = V_U
        if W == 1:
            a(f"""
	v_ffbl_b32 {out}, {P[0]}
	v_lshlrev_b32 {u[0]}, {out}, {V_ONE}
	v_add_u32 {u[1]}, 1, {out}
	v_xor_b32 {P[0]}, {P[0]}, {u[0]}
	v_max_u32 {V_HIGH}, {V_HIGH}, {u[1]}""")
            return
        a(f"""
	v_ffbl_b32 {u[0]}, {P[0]}
	v_ffbl_b32 {u[1]}, {P[1]}
	v_ffbl_b32 {u[2]}, {P[2]}
	v_ffbl_b32 {u[3]}, {P[3]}
	v_or_b32 {u[1]}, 32, {u[1]}
	v_or_b32 {u[2]}, 64, {u[2]}
	v_or_b32 {u[3]}, 96, {u[3]}
	v_min3_u32 {out}, {u[0]}, {u[1]}, {u[2]}
	v_min_u32 {out}, {out}, {u[3]}
	v_lshlrev_b32 {u[0]}, {out}, {V_ONE}
	v_lshrrev_b32 {u[1]}, 5, {out}
	v_add_u32 {u[2]}, 1, {out}
	v_max_u32 {V_HIGH}, {V_HIGH}, {u[2]}""")
        for w in range(4):
            a(f"\tv_cmp_eq_u32_e64 {S_M[w]}, {w}, {u[1]}")
        a("\ts_nop 0")
        for w in range(4):
            a(f"\tv_cndmask_b32_e64 {u[2]}, 0, {u[0]}, {S_M[w]}\n\tv_xor_b32 {P[w]}, {P[w]}, {u[2]}")

    def pool_give(self, reg, W):
        a = self.a
        u = V_U
        if W == 1:
            a(f"\tv_lshlrev_b32 {u[0]}, {reg}, {V_ONE}\n\tv_or_b32 {P[0]}, {P[0]}, {u[0]}")
            return
        a(f"\tv_lshlrev_b32 {u[0]}, {reg}, {V_ONE}\n\tv_lshrrev_b32 {u[1]}, 5, {reg}")
        for w in range(4):
            a(f"\tv_cmp_eq_u32_e64 {S_M[w]}, {w}, {u[1]}")
        a("\ts_nop 0")
        for w in range(4):
            a(f"\tv_cndmask_b32_e64 {u[2]}, 0, {u[0]}, {S_M[w]}\n\tv_or_b32 {P[w]}, {P[w]}, {u[2]}")

    def map_addr(self, dst, sreg):
        """LDS address of map[sreg][lane]"""
        self.a(f"""
	s_lshl_b32 {S_T2}, {sreg}, 6
	s_add_u32 {S_T2}, {S_T2}, {S_MAPBASE}
	v_add_u32 {dst}, {S_T2}, {V_LANE}""")

    def alloc_where_dead(self, val, addr, within, W, lab):
        """lanes of `within` whose map value `val` is DEAD get a fresh register (written back)"""
        a = self.a
        a(f"""
	v_cmp_eq_u32 vcc, {V_DEAD}, {val}
	s_and_b64 vcc, vcc, {within}
	s_cbranch_scc0 {lab}
	s_mov_b64 exec, vcc""")
        self.pool_take(val, W)
        a(f"""
	ds_write_b8 {addr}, {val}
	s_mov_b64 exec, {within}
{lab}:""")

    def emit_op(self, mask):
        """store {V_EW0, V_EW1} one op below dst for the lanes in `mask` (exec is left = mask)"""
        self.a(f"""
	s_mov_b64 exec, {mask}
	v_add_co_u32 v46, vcc, -8, v46
	v_addc_co_u32 v47, vcc, -1, v47, vcc
	v_add_u32 {V_COUNT}, 1, {V_COUNT}
	global_store_dwordx2 {V_DST}, v[56:57], off""")

    # ---- prune sweep -------------------------------------------------------------------------
    def emit_prune(self, W):
        """Reverse sweep for the lanes in S_PRUNE (vm/data.rs:123-318 as in prune_sweep, kernels.hip);
        W = words of the free-register pool (1: tapes of <= 32 registers, 4: <= 128)."""
        a = self.a
        L = lambda n: f".Lfh_tiles_p{W}_{n}"
        a(f"""
; ---- prune sweep ({32 * W} registers): ops {S_LEN}-1 .. 0 of the tape at {S_TAPE}; lanes {S_PRUNE} --------
.Lfh_tiles_prune{W}:
	s_mov_b32 {S_K}, {S_LEN}
	s_mov_b32 {S_CI}, {S_NCH}
	v_mov_b32 {V_COUNT}, 0
	v_mov_b32 {V_KEPT}, 0
	v_mov_b32 {V_HIGH}, 0""")
        for w in range(W):
            a(f"\tv_mov_b32 {P[w]}, -1")
        a(f"""
	; stage the tape in LDS, in the (now dead) interval register file, when it fits: a scalar
	; load per op would cost its full latency every step
	s_lshl_b32 {S_T0}, {S_LEN}, 3
	s_mov_b32 {S_STAGED}, 0
	s_cmp_le_u32 {S_T0}, {S_CHBASE}
	s_cbranch_scc0 {L('next')}
	s_mov_b32 {S_STAGED}, 1
	v_lshlrev_b32 {V_U[0]}, 4, {V_LANE}
{L('stage')}:
	v_cmp_gt_u32 vcc, {S_T0}, {V_U[0]}
	s_and_saveexec_b64 {S_SAVE}, vcc
	s_cbranch_execz {L('staged')}
	global_load_dwordx4 v[68:71], {V_U[0]}, {S_TAPE}
	s_waitcnt vmcnt(0)
	ds_write_b128 {V_U[0]}, v[68:71]
	s_mov_b64 exec, {S_SAVE}
	v_add_u32 {V_U[0]}, 0x400, {V_U[0]}
	s_branch {L('stage')}
{L('staged')}:
	s_mov_b64 exec, -1
	s_waitcnt lgkmcnt(0)
	s_sub_u32 {S_T0}, {S_T0}, 8
	v_mov_b32 {V_U[0]}, {S_T0}
	ds_read_b64 {V_PW}, {V_U[0]}
{L('next')}:
	s_mov_b64 exec, -1
	s_sub_u32 {S_K}, {S_K}, 1
	s_cbranch_scc1 {L('done')}
	s_cmp_eq_u32 {S_STAGED}, 0
	s_cbranch_scc1 {L('slow')}
	s_waitcnt lgkmcnt(0)
	v_readfirstlane_b32 {S_W0}, v78
	v_readfirstlane_b32 {S_W1}, v79
	; prefetch op k-1 (a read below LDS address 0 at k = 0 is harmless and never used)
	s_lshl_b32 {S_T0}, {S_K}, 3
	s_sub_u32 {S_T0}, {S_T0}, 8
	v_mov_b32 {V_U[0]}, {S_T0}
	ds_read_b64 {V_PW}, {V_U[0]}
	s_branch {L('haveop')}
{L('slow')}:
	s_lshl_b32 {S_T0}, {S_K}, 3
	s_add_u32 s76, s44, {S_T0}
	s_addc_u32 s77, s45, 0
	s_load_dwordx2 {S_CUR}, {S_T64}, 0x0
	s_waitcnt lgkmcnt(0)
{L('haveop')}:
	s_and_b32 {S_OP}, {S_W0}, 0xff
	s_bfe_u32 {S_OUT}, {S_W0}, 0xc0008
	s_lshr_b32 {S_A}, {S_W0}, 20
	; choice of this op per lane (choice ops only): c in {V_C}
	s_mov_b32 {S_T3}, 0
	s_cmp_ge_u32 {S_OP}, 30
	s_cbranch_scc0 {L('nochoice')}
	s_cmp_lt_u32 {S_OP}, 34
	s_cselect_b32 {S_T3}, 1, 0
	s_cmp_ge_u32 {S_OP}, 42
	s_cselect_b32 {S_T0}, 1, 0
	s_cmp_lt_u32 {S_OP}, 46
	s_cselect_b32 {S_T1}, 1, 0
	s_and_b32 {S_T0}, {S_T0}, {S_T1}
	s_lshl_b32 {S_T0}, {S_T0}, 1
	s_or_b32 {S_T3}, {S_T3}, {S_T0}
	s_cmp_eq_u32 {S_T3}, 0
	s_cbranch_scc1 {L('nochoice')}
	; {S_T3}: 1 = choice op reg,reg  2 = choice op reg,imm
	s_sub_u32 {S_CI}, {S_CI}, 1
	s_and_b32 {S_T0}, {S_CI}, 15
	s_cmp_eq_u32 {S_T0}, 15
	s_cselect_b32 {S_T1}, 1, 0
	s_add_u32 {S_T2}, {S_CI}, 1
	s_cmp_eq_u32 {S_T2}, {S_NCH}
	s_cselect_b32 {S_T2}, 1, 0
	s_or_b32 {S_T1}, {S_T1}, {S_T2}
	s_cmp_eq_u32 {S_T1}, 0
	s_cbranch_scc1 {L('haveword')}
	s_lshr_b32 {S_T1}, {S_CI}, 4
	s_lshl_b32 {S_T1}, {S_T1}, 8
	s_add_u32 {S_T1}, {S_T1}, {S_CHBASE}
	v_add_u32 {V_TADDR}, {S_T1}, {V_L4}
	ds_read_b32 {V_CWP}, {V_TADDR}
	s_waitcnt lgkmcnt(0)
{L('haveword')}:
	s_lshl_b32 {S_T0}, {S_T0}, 1
	v_bfe_u32 {V_C}, {V_CWP}, {S_T0}, 2
{L('nochoice')}:
	s_cmp_eq_u32 {S_OP}, 0
	s_cbranch_scc1 {L('output')}
	; ---- all map traffic of this op in ONE LDS round trip: read map[out], kill it (a no-op for
	; lanes where it is dead already), then read the operands' entries (LDS keeps the order)""")
        self.map_addr(V_AOUT, S_OUT)
        a(f"""
	ds_read_u8 {V_NO}, {V_AOUT}
	s_mov_b64 exec, {S_PRUNE}
	ds_write_b8 {V_AOUT}, {V_DEAD}
	s_mov_b64 exec, -1
	v_mov_b32 {V_MAV}, 0
	; operand a: every op except INPUT (1) and COPY_IMM (3)
	s_cmp_eq_u32 {S_OP}, 1
	s_cbranch_scc1 {L('noa')}
	s_cmp_eq_u32 {S_OP}, 3
	s_cbranch_scc1 {L('noa')}""")
        self.map_addr(V_AA, S_A)
        a(f"""
	ds_read_u8 {V_MAV}, {V_AA}
{L('noa')}:
	; operand b: reg,reg forms (22..33)
	s_mov_b32 {S_RR}, 0
	s_cmp_ge_u32 {S_OP}, 22
	s_cbranch_scc0 {L('nob')}
	s_cmp_lt_u32 {S_OP}, 34
	s_cbranch_scc0 {L('nob')}
	s_mov_b32 {S_RR}, 1""")
        self.map_addr(V_AB, S_W1)
        a(f"""
	ds_read_u8 {V_MBV}, {V_AB}
{L('nob')}:
	s_waitcnt lgkmcnt(0)
	v_cmp_ne_u32 vcc, {V_DEAD}, {V_NO}
	s_and_b64 {S_LIVE}, vcc, {S_PRUNE}
	s_cbranch_scc0 {L('next')}
	; ---- decided choices / copies alias `out` with the surviving operand ---------------------
	s_mov_b64 {S_ALIAS}, 0
	s_mov_b64 {S_CIMM}, 0
	s_cmp_eq_u32 {S_OP}, 2
	s_cbranch_scc0 {L('notcopy')}
	s_mov_b64 {S_ALIAS}, {S_LIVE}
	v_mov_b32 {V_AV}, {V_MAV}
	v_mov_b32 {V_AAL}, {V_AA}
	s_branch {L('alias')}
{L('notcopy')}:
	s_cmp_eq_u32 {S_T3}, 0
	s_cbranch_scc1 {L('keep')}
	v_cmp_eq_u32_e64 {S_MA}, 1, {V_C}
	v_cmp_eq_u32_e64 {S_MB}, 2, {V_C}
	v_mov_b32 {V_AV}, {V_MAV}
	v_mov_b32 {V_AAL}, {V_AA}
	s_and_b64 {S_MA}, {S_MA}, {S_LIVE}
	s_and_b64 {S_MB}, {S_MB}, {S_LIVE}
	s_cmp_eq_u32 {S_T3}, 1
	s_cbranch_scc0 {L('rimm')}
	v_cndmask_b32_e64 {V_AV}, {V_AV}, {V_MBV}, {S_MB}
	v_cndmask_b32_e64 {V_AAL}, {V_AAL}, {V_AB}, {S_MB}
	s_or_b64 {S_ALIAS}, {S_MA}, {S_MB}
	s_branch {L('alias')}
{L('rimm')}:
	s_mov_b64 {S_ALIAS}, {S_MA}
	s_mov_b64 {S_CIMM}, {S_MB}
{L('alias')}:
	s_cmp_eq_u64 {S_ALIAS}, 0
	s_cbranch_scc1 {L('cimm')}
	v_cmp_eq_u32 vcc, {V_DEAD}, {V_AV}
	s_and_b64 {S_MA}, vcc, {S_ALIAS}
	s_andn2_b64 {S_MB}, {S_ALIAS}, vcc
	; operand not live yet: it takes over the register, nothing is emitted
	s_mov_b64 exec, {S_MA}
	ds_write_b8 {V_AAL}, {V_NO}
	s_cmp_eq_u64 {S_MB}, 0
	s_cbranch_scc1 {L('cimm')}
	; operand already live: COPY_REG no <- map[alias]; the register of `out` is free before it
	s_mov_b64 exec, {S_MB}""")
        self.pool_give(V_NO, W)
        a(f"""
	v_lshlrev_b32 {V_EW0}, 8, {V_NO}
	v_lshl_or_b32 {V_EW0}, {V_AV}, 20, {V_EW0}
	v_or_b32 {V_EW0}, 2, {V_EW0}
	v_mov_b32 {V_EW1}, 0""")
        self.emit_op(S_MB)
        a(f"""
{L('cimm')}:
	s_cmp_eq_u64 {S_CIMM}, 0
	s_cbranch_scc1 {L('keep')}
	s_mov_b64 exec, {S_CIMM}""")
        self.pool_give(V_NO, W)
        a(f"""
	v_lshlrev_b32 {V_EW0}, 8, {V_NO}
	v_or_b32 {V_EW0}, 3, {V_EW0}
	v_mov_b32 {V_EW1}, {S_W1}""")
        self.emit_op(S_CIMM)
        a(f"""
{L('keep')}:
	s_andn2_b64 {S_KEEP}, {S_LIVE}, {S_ALIAS}
	s_andn2_b64 {S_KEEP}, {S_KEEP}, {S_CIMM}
	s_cmp_eq_u64 {S_KEEP}, 0
	s_cbranch_scc1 {L('next')}
	s_mov_b64 exec, {S_KEEP}""")
        self.pool_give(V_NO, W)
        a(f"\tv_mov_b32 {V_EW1}, {S_W1}")
        # operand a (V_MAV = 0 and never DEAD when the op has none)
        self.alloc_where_dead(V_MAV, V_AA, S_KEEP, W, L('a_ok'))
        a(f"""
	s_cmp_eq_u32 {S_RR}, 0
	s_cbranch_scc1 {L('b_ok')}
	v_mov_b32 {V_EW1}, {V_MAV}
	s_cmp_eq_u32 {S_A}, {S_W1}
	s_cbranch_scc1 {L('b_ok')}""")
        self.alloc_where_dead(V_MBV, V_AB, S_KEEP, W, L('b_alloc'))
        a(f"""
	v_mov_b32 {V_EW1}, {V_MBV}
{L('b_ok')}:
	s_cmp_eq_u32 {S_T3}, 0
	s_cbranch_scc1 {L('nokept')}
	v_add_u32 {V_KEPT}, 1, {V_KEPT}
{L('nokept')}:
	v_lshlrev_b32 {V_EW0}, 8, {V_NO}
	v_lshl_or_b32 {V_EW0}, {V_MAV}, 20, {V_EW0}
	v_or_b32 {V_EW0}, {S_OP}, {V_EW0}""")
        self.emit_op(S_KEEP)
        a(f"""
	s_branch {L('next')}
{L('output')}:""")
        self.map_addr(V_AA, S_A)
        a(f"""
	ds_read_u8 {V_MAV}, {V_AA}
	s_waitcnt lgkmcnt(0)
	s_mov_b64 exec, {S_PRUNE}""")
        self.alloc_where_dead(V_MAV, V_AA, S_PRUNE, W, L('o_ok'))
        a(f"""
	v_lshlrev_b32 {V_EW0}, 20, {V_MAV}
	v_mov_b32 {V_EW1}, {S_W1}""")
        self.emit_op(S_PRUNE)
        a(f"""
	s_branch {L('next')}
{L('done')}:
	s_mov_b64 exec, -1
	s_waitcnt vmcnt(0) lgkmcnt(0)
	s_setpc_b64 {S_RET}""")

    # ---- kernel ---------------------------------------------------------------------------------
    def emit_kernel(self):
        a = self.a
        o = self.off
        name = self.name
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
	v_mov_b32 {V_DEAD}, 0xff
	v_mov_b32 {V_ZERO}, 0
	s_mov_b32 {S_SIGN}, 0x80000000
	s_mov_b32 {S_ABSM}, 0x7fffffff
	v_lshlrev_b32 {V_L8}, 3, {V_LANE}
	v_lshlrev_b32 {V_L4}, 2, {V_LANE}
	s_waitcnt lgkmcnt(0)
	s_mov_b32 {S_LEVEL}, s8
	s_mov_b32 {S_BIG}, s9
	s_mov_b32 {S_MAXCH}, s11
	s_mov_b32 {S_MAXREGS}, s10
	; LDS layout
	s_lshl_b32 {S_CHBASE}, {S_MAXREGS}, 9
	s_add_u32 {S_T0}, {S_MAXCH}, 15
	s_lshr_b32 {S_T0}, {S_T0}, 4
	s_lshl_b32 {S_T0}, {S_T0}, 8
	s_add_u32 {S_MAPBASE}, {S_CHBASE}, {S_T0}
	; per-list state: slots[big], n_slots[big][level], eval_cur[big][level]
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
	s_add_u32 {S_CUROFF}, {S_T0}, {o['eval_cur']}
	s_add_u32 s76, s4, {S_T0}
	s_addc_u32 s77, s5, 0
	s_load_dword {S_NSLOTS}, {S_T64}, {o['n_slots']}
	s_lshl_b32 {S_T0}, {S_BIG}, 3
	s_add_u32 s76, s4, {S_T0}
	s_addc_u32 s77, s5, 0
	s_load_dwordx2 {S_CHW}, {S_T64}, {o['chw']}
	s_load_dwordx2 {S_ARENA}, {S_STATE}, {o['arena']}
	s_load_dword {S_ARENACAP}, {S_STATE}, {o['arena_cap']}
	s_getpc_b64 {S_HBASE}
.Lfh_tiles_pc0:
	s_add_u32 s42, s42, .Lfh_tiles_handlers - .Lfh_tiles_pc0
	s_addc_u32 s43, s43, 0
	s_waitcnt lgkmcnt(0)
	s_min_u32 {S_NSLOTS}, {S_NSLOTS}, {S_T2}
	s_load_dwordx2 s[76:77], {S_KERNARG}, 0x20
	s_waitcnt lgkmcnt(0)
	s_mov_b32 {S_SKIPR}, s76
	s_mov_b32 {S_SKIPC}, s77
.Lfh_tiles_outer:
	; ---- next slot: round robin over the waves ----------------------------------------------------
	s_cmp_ge_u32 {S_SI}, {S_NSLOTS}
	s_cbranch_scc1 .Lfh_tiles_exit
	s_mov_b32 {S_T0}, {S_SI}
	s_add_u32 {S_SI}, {S_SI}, {S_NWG}
	; export mode: this slot's choice words at chw[big] + si * (words * 256 B); words = flags[31:16], or
	; (0) those of this launch's LDS layout
	s_sub_u32 {S_T2}, {S_MAPBASE}, {S_CHBASE}
	s_lshr_b32 {S_T3}, {S_FLAGS}, 16
	s_lshl_b32 {S_T3}, {S_T3}, 8
	s_cmp_eq_u32 {S_T3}, 0
	s_cselect_b32 {S_T2}, {S_T2}, {S_T3}
	s_mul_hi_u32 {S_T3}, {S_T0}, {S_T2}
	s_mul_i32 {S_T2}, {S_T0}, {S_T2}
	s_add_u32 s80, s78, {S_T2}
	s_addc_u32 s81, s79, {S_T3}
	s_mul_i32 {S_T1}, {S_T0}, {SLOT_SIZE}
	s_mul_hi_u32 {S_T0}, {S_T0}, {SLOT_SIZE}
	s_add_u32 s20, s14, {S_T1}
	s_addc_u32 s21, s15, {S_T0}
	s_load_dwordx4 s[24:27], {S_SLOT}, 0x0
	s_load_dwordx2 {S_ACT}, {S_SLOT}, {SL_ACT}
	s_load_dwordx2 {S_TV}, {S_SLOT}, 0x20
	s_waitcnt lgkmcnt(0)
	s_cmp_eq_u64 {S_ACT}, 0
	s_cbranch_scc1 .Lfh_tiles_outer
	s_and_b32 {S_NREGS}, {S_RC}, 0xffff
	s_lshr_b32 {S_NCH}, {S_RC}, 16
	; is this slot for this launch's LDS layout?
	s_cmp_gt_u32 {S_NREGS}, {S_MAXREGS}
	s_cbranch_scc1 .Lfh_tiles_outer
	s_cmp_gt_u32 {S_NCH}, {S_MAXCH}
	s_cbranch_scc1 .Lfh_tiles_outer
	s_cmp_le_u32 {S_NREGS}, {S_SKIPR}
	s_cselect_b32 {S_T0}, 1, 0
	s_cmp_le_u32 {S_NCH}, {S_SKIPC}
	s_cselect_b32 {S_T1}, 1, 0
	s_and_b32 {S_T0}, {S_T0}, {S_T1}
	s_cmp_eq_u32 {S_T0}, 1
	s_cbranch_scc1 .Lfh_tiles_outer
	s_memrealtime s[56:57]
	s_memtime s[62:63]""")
        for k, r in enumerate((VX[0], VX[1], VY[0], VY[1], VZ[0], VZ[1])):
            a(f"\tglobal_load_dword {r}, {V_L4}, {S_SLOT} offset:{SL_XYZ + 256 * k}")
        a(f"""
	s_waitcnt lgkmcnt(0)
	s_mov_b32 s44, {S_OFF}
	s_mov_b32 s45, 0
	s_lshl_b64 {S_TAPE}, {S_TAPE}, 3
	s_add_u32 s44, s44, s10
	s_addc_u32 s45, s45, s11
	s_mov_b32 {S_REM}, {S_LEN}
	s_mov_b32 {S_CI}, 0
	s_mov_b64 {S_DECIDED}, 0
	v_mov_b32 {V_CW}, 0
	v_mov_b32 {V_RESL}, {V_QNAN}
	v_mov_b32 {V_RESH}, {V_QNAN}
	s_waitcnt vmcnt(0)
	s_getpc_b64 {S_RET}
.Lfh_tiles_pc1:
	s_add_u32 s74, s74, .Lfh_tiles_ret1 - .Lfh_tiles_pc1
	s_addc_u32 s75, s75, 0
	s_branch .Lfh_tiles_run
.Lfh_tiles_ret1:
	s_memrealtime s[58:59]
	s_memtime s[90:91]
	s_waitcnt lgkmcnt(0)
	s_sub_u32 s62, s90, s62
	s_subb_u32 s63, s91, s63
	; ---- flush the last partial choice word ----------------------------------------------------
	s_and_b32 {S_T0}, {S_CI}, 15
	s_cmp_eq_u32 {S_T0}, 0
	s_cbranch_scc1 .Lfh_tiles_noflush
	s_lshr_b32 {S_T0}, {S_CI}, 4
	s_lshl_b32 {S_T0}, {S_T0}, 8
	s_bitcmp1_b32 {S_FLAGS}, 1
	s_cbranch_scc1 .Lfh_tiles_flush_export
	s_add_u32 {S_T0}, {S_T0}, {S_CHBASE}
	v_add_u32 {V_TADDR}, {S_T0}, {V_L4}
	ds_write_b32 {V_TADDR}, {V_CW}
	s_branch .Lfh_tiles_noflush
.Lfh_tiles_flush_export:
	s_add_u32 s82, s80, {S_T0}
	s_addc_u32 s83, s81, 0
	global_store_dword {V_L4}, {V_CW}, {S_CHWTMP}
.Lfh_tiles_noflush:
	global_store_dword {V_L4}, {V_RESL}, {S_SLOT} offset:{SL_RES}
	global_store_dword {V_L4}, {V_RESH}, {S_SLOT} offset:{SL_RES + 256}
	s_bitcmp1_b32 {S_FLAGS}, 2                       ; forward pass only (tape groups: k_ttop3d goes on)
	s_cbranch_scc1 .Lfh_tiles_outer
	; ---- classify: ambiguous = act && !(hi < 0) && !(lo > 0); prune those whose trace decided --
	v_cmp_gt_f32_e64 {S_M[0]}, 0, {V_RESH}
	v_cmp_gt_f32_e64 {S_M[1]}, {V_RESL}, 0
	v_mov_b32 {V_COFF}, {S_OFF}
	v_mov_b32 {V_CLEN}, {S_LEN}
	v_mov_b32 {V_CRC}, {S_RC}
	s_or_b64 {S_M[0]}, {S_M[0]}, {S_M[1]}
	s_andn2_b64 {S_PRUNE}, {S_ACT}, {S_M[0]}
	s_and_b64 {S_PRUNE}, {S_PRUNE}, {S_DECIDED}
	s_cmp_eq_u64 {S_PRUNE}, 0
	s_cbranch_scc1 .Lfh_tiles_store
	; ---- arena: every pruned child reserves a slot as long as the parent tape -------------------
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
	s_cbranch_scc1 .Lfh_tiles_overflow
	s_cmp_le_u32 {S_T0}, {S_ARENACAP}
	s_cbranch_scc0 .Lfh_tiles_overflow
	s_bitcmp1_b32 {S_FLAGS}, 1
	s_cbranch_scc0 .Lfh_tiles_sweep_here
	; export mode: mark the lanes to prune (c_len = ~0) and leave the end of their arena slot in c_off
	v_add_u32 {T[0]}, 1, {V_RANK}
	v_mul_lo_u32 {T[0]}, {T[0]}, {S_LEN}
	v_add_u32 {T[0]}, {S_BASE}, {T[0]}
	v_mov_b32 {T[1]}, -1
	v_cndmask_b32_e64 {V_COFF}, {V_COFF}, {T[0]}, {S_PRUNE}
	v_cndmask_b32_e64 {V_CLEN}, {V_CLEN}, {T[1]}, {S_PRUNE}
	s_branch .Lfh_tiles_store
.Lfh_tiles_sweep_here:
	; map[r][lane] = DEAD for r < n_regs: n_regs * 64 bytes written as dwords
	s_lshl_b32 {S_T0}, {S_NREGS}, 6
	v_add_u32 {V_TADDR}, {S_MAPBASE}, {V_L4}
	v_mov_b32 {T[0]}, -1
	s_mov_b32 {S_T1}, 0
.Lfh_tiles_mapinit:
	ds_write_b32 {V_TADDR}, {T[0]}
	v_add_u32 {V_TADDR}, 0x100, {V_TADDR}
	s_add_u32 {S_T1}, {S_T1}, 0x100
	s_cmp_lt_u32 {S_T1}, {S_T0}
	s_cbranch_scc1 .Lfh_tiles_mapinit
	; dst = arena + 8 * (base + (rank + 1) * len): one past the last op of this lane's slot
	v_add_u32 {T[0]}, 1, {V_RANK}
	v_mul_lo_u32 {T[0]}, {T[0]}, {S_LEN}
	v_add_u32 {T[0]}, {S_BASE}, {T[0]}
	v_mov_b32 {T[1]}, 0
	v_mov_b32 {T[4]}, {T[0]}
	v_lshlrev_b64 v[46:47], 3, v[16:17]
	v_mov_b32 {T[2]}, s11
	v_add_co_u32 v46, vcc, s10, v46
	v_addc_co_u32 v47, vcc, {T[2]}, v47, vcc
	s_getpc_b64 {S_RET}
.Lfh_tiles_pc2:
	s_add_u32 s74, s74, .Lfh_tiles_ret2 - .Lfh_tiles_pc2
	s_addc_u32 s75, s75, 0
	s_cmp_le_u32 {S_NREGS}, 32
	s_cbranch_scc1 .Lfh_tiles_prune1
	s_branch .Lfh_tiles_prune4
.Lfh_tiles_ret2:
	; child = {{ base + (rank+1)*len - count, count, high | kept << 16 }} for the pruned lanes
	v_sub_u32 {T[4]}, {T[4]}, {V_COUNT}
	v_lshl_or_b32 {T[5]}, {V_KEPT}, 16, {V_HIGH}
	v_cndmask_b32_e64 {V_COFF}, {V_COFF}, {T[4]}, {S_PRUNE}
	v_cndmask_b32_e64 {V_CLEN}, {V_CLEN}, {V_COUNT}, {S_PRUNE}
	v_cndmask_b32_e64 {V_CRC}, {V_CRC}, {T[5]}, {S_PRUNE}
	s_branch .Lfh_tiles_store
.Lfh_tiles_overflow:
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
.Lfh_tiles_store:
	; diagnostics (kernarg `probe`; the atomics serialise, so never in production runs): ticks
	; (100 MHz) spent in the forward pass / in classify + prune, per level
	s_bitcmp1_b32 {S_FLAGS}, 0
	s_cbranch_scc0 .Lfh_tiles_noprobe
	s_memrealtime s[60:61]
	s_waitcnt lgkmcnt(0)
	s_sub_u32 s60, s60, s58
	s_subb_u32 s61, s61, s59
	s_sub_u32 s58, s58, s56
	s_subb_u32 s59, s59, s57
	s_lshl_b32 {S_T0}, {S_LEVEL}, 3
	v_cmp_eq_u32 vcc, 0, {V_LANE}
	s_and_saveexec_b64 {S_SAVE}, vcc
	v_mov_b32 {T[2]}, {S_T0}
	v_mov_b32 {T[0]}, s58
	v_mov_b32 {T[1]}, s59
	global_atomic_add_x2 {T[2]}, v[16:17], {S_STATE} offset:{o['stat'] + 8 * 32}
	v_mov_b32 {T[0]}, s60
	v_mov_b32 {T[1]}, s61
	global_atomic_add_x2 {T[2]}, v[16:17], {S_STATE} offset:{o['stat'] + 8 * 40}
	v_mov_b32 {T[0]}, s62
	v_mov_b32 {T[1]}, s63
	global_atomic_add_x2 {T[2]}, v[16:17], {S_STATE} offset:{o['stat'] + 8 * 16}
	s_mov_b64 exec, {S_SAVE}
.Lfh_tiles_noprobe:
	global_store_dword {V_L4}, {V_COFF}, {S_SLOT} offset:{SL_COFF}
	global_store_dword {V_L4}, {V_CLEN}, {S_SLOT} offset:{SL_CLEN}
	global_store_dword {V_L4}, {V_CRC}, {S_SLOT} offset:{SL_CRC}
	s_branch .Lfh_tiles_outer
.Lfh_tiles_outer_drain:
	s_waitcnt vmcnt(0)                              ; the per-lane interval loads of the skipped slot
	s_branch .Lfh_tiles_outer
.Lfh_tiles_exit:
	s_endpgm
.Lfh_tiles_end:
	.size {name}, .Lfh_tiles_end - {name}
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
        self.emit_prune(1)
        self.emit_prune(4)


def gen_tiles(a, off, trans=None):
    """fh_tiles, or with `trans` (the compiled routines' assembly) fh_tiles_t: the same kernel with handlers for the transcendental
    opcodes (generated into a scratch buffer, its labels and name renamed)"""
    if trans:
        b = Asm()
        b.uid = a.uid + 200000
        t = Tiles(b, off, trans=trans)
        t.emit_kernel()
        a(b.text().replace(".Lfh_tiles_", ".Lfh_tiles_t_").replace("fh_tiles", "fh_tiles_t").replace("fh_tiles_t_t_", "fh_tiles_t_"))
        a.uid = max(a.uid, b.uid)
        return "fh_tiles_t", 40, t.n_vgpr, [(8, "global_buffer")] + [(4, "by_value")] * 8
    t = Tiles(a, off)
    t.emit_kernel()
    return t.name, 40, N_VGPR, [(8, "global_buffer")] + [(4, "by_value")] * 8
