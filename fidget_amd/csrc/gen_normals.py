"""fh_normals - the normals of a z-slab's hits by an assembly gradient interpreter (gfx950).

What k_normals3d (kernels.hip) does in HIP C++ - a `switch` per tape op and a register file in LDS: 112 us per 1024^3 frame of
prospero.vm, the longest thing between the last leaf kernel and the finished image - with the machinery of the leaf interpreter
(gen_interp.py): one `s_setpc_b64` per op into a table of handlers, the register file in VGPRs addressed with `s_set_gpr_idx_on`.
A gradient value is four VGPRs {v, dx, dy, dz} (fidget-core/src/types/grad.rs; dev_ops.hpp GR), so the interpreter is the
leaf interpreter's with ZB = 4 "samples" per register and handlers that know what the four mean: add / sub stay element-wise,
everything else follows dev_ops.hpp's gr_* functions operation for operation (no fused multiply-add, the IEEE division and square
root sequences of the leaf handlers) - the normals are the C++ kernel's, which are the oracle's, bit for bit
(vm/mod.rs:1091-1397 via dev_ops.hpp GRAD).

Per wave pass: ONE leaf of the slab's list of leaves that own a hit (k_hits3d, kernels.hip: the distinct leaf numbers in the finished
z-buffer words of the footprints whose leaves need <= 32 registers) - its 8 x 8 pixels' z-buffer words say which of them it hit at which
depth; the input gradients of every lane's voxel (dev_ops.hpp xf_grad: the screen -> model matrix applied to {x,1,0,0}, {y,0,1,0},
{z,0,0,1}, then the division by w - always, as the C++ does) are made, the leaf's tape is run for all 64 lanes, and the lanes it hit store
dx, dy, dz and clear the leaf number in their z-buffer word (voxel.rs:447-482).  (Until round 6 a wave took a footprint and ran the tapes
of all the leaves among its pixels one after the other: a launch lasted as long as its busiest wave - 414 us on bear.vm 512^3.)

Tapes with a modulo keep the C++ kernel (its gradient needs div_euclid); fh_normals_t has the transcendental, rng and atan2 handlers.

kernarg: { FhRenderState* S; u32 n_waves (a multiple of 64); u32 axis slots x | y << 8 | z << 16 (0xFF: none); u32 z_lo; u32 z_hi;
           u32 entries per list of k_hits3d; u32 pad }
         - this launch's hits are those with z_lo < depth <= z_hi.
"""
from gen_interp import (Interp, OPS, FILE, S_KERNARG, S_STATE, S_MAT, S_SIGN, S_ABSM, S_ARENA, S_TAPE, S_LEN, S_W1, S_T0, S_OUT, S_A, S_T1, S_PC, S_SAVE,
                        S_M, S_RET, V_LANE, V_QNAN, V_SQRTC, VT, VU, VW, VD, SRC0, SRC1, DST, kernel_header, kernel_footer, common_consts,
                        handler_base, call_interp)

NR = 40      # registers of a leaf tape the kernel takes (4 VGPRs each: a file of 160 behind 64 fixed registers - two waves per SIMD as with 32; capi_render.hpp FH_NORMAL_REGS)
T_BASE = FILE + NR * 4      # register window of the transcendental routines (fh_normals_t)
S_SLOTX, S_SLOTY, S_SLOTZ = "s0", "s1", "s3"
S_WI, S_NWG, S_NFP = "s6", "s7", "s40"
S_WIDTH, S_HEIGHT = "s24", "s25"
S_ZLO, S_ZHI = "s26", "s27"
S_NORMALS, S_LEAVES, S_ZBUF, S_FPLIST = "s[32:33]", "s[34:35]", "s[36:37]", "s[38:39]"
S_TODO = "s[80:81]"
S_CUR = "s78"
V_PIX, V_ID, V_DEPTH, V_NOFF = "v1", "v2", "v3", "v4"       # byte offset of the pixel's z-buffer word, its two halves, byte offset of its normal
V_PX, V_PY, V_PZ = "v5", "v6", "v7"
VRES = ["v10", "v11", "v12", "v13"]
GX, GY, GZ = ["v22", "v23", "v24", "v25"], ["v30", "v31", "v32", "v33"], ["v38", "v39", "v40", "v41"]   # the lanes' input gradients (free scratch of a 4-wide interpreter)
XR = [["v14", "v15", "v16", "v17"], ["v52", "v53", "v54", "v55"], ["v56", "v57", "v58", "v59"], ["v60", "v61", "v62", "v63"]]   # rows of the transform (set-up only)


class GradInterp(Interp):
    INPLACE = set()

    def __init__(self, a, name, off, trans=False):
        super().__init__(a, name, NR, 4, "grad", off, trans=trans)

    def call(self, fn, arg, res, arg2=None):
        """res = <fn>(arg [, arg2]) by the compiled routine fh_tn_<fn> (gen_trans.py, embedded with its vector registers in
        v[T_BASE .. T_BASE + 25] behind the register file); clobbers those, s86..s97 and vcc"""
        here, ret = self.a.label("call"), self.a.label("ret")
        self.a(f"\tv_mov_b32 v{T_BASE}, {arg}" + (f"\n\tv_mov_b32 v{T_BASE + 1}, {arg2}" if arg2 else ""))
        self.a(f"""
	s_getpc_b64 s[96:97]
{here}:
	s_add_u32 s96, s96, {ret} - {here}
	s_addc_u32 s97, s97, 0
	s_branch fh_tn_{fn}
{ret}:
	v_mov_b32 {res}, v{T_BASE}""")

    # ---- scalar helpers on plain VGPRs -----------------------------------------------------------
    def div1(self, n, d, r):
        """r = n / d, IEEE (the div_scale / rcp / fma / div_fmas / div_fixup sequence of Interp.f_div); r may be n or d"""
        t = VD
        self.a(f"""
	v_div_scale_f32 {t[0]}, {S_SAVE}, {d}, {d}, {n}
	v_rcp_f32 {t[1]}, {t[0]}
	v_div_scale_f32 {t[2]}, vcc, {n}, {d}, {n}
	v_fma_f32 {t[3]}, -{t[0]}, {t[1]}, 1.0
	v_fmac_f32 {t[1]}, {t[3]}, {t[1]}
	v_mul_f32 {t[3]}, {t[2]}, {t[1]}
	v_fma_f32 {t[4]}, -{t[0]}, {t[3]}, {t[2]}
	v_fmac_f32 {t[3]}, {t[4]}, {t[1]}
	v_fma_f32 {t[0]}, -{t[0]}, {t[3]}, {t[2]}
	v_div_fmas_f32 {t[0]}, {t[0]}, {t[1]}, {t[3]}
	v_div_fixup_f32 {r}, {t[0]}, {d}, {n}""")

    def sqrt1(self, x, r):
        """r = sqrtf(x), correctly rounded (the full sequence of Interp.f_sqrt)"""
        d = VD
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
	v_cndmask_b32 {r}, {d[0]}, {d[1]}, vcc""")

    def gr_mul(self, A, B, R):
        """dev_ops.hpp gr_mul: v = a.v b.v, d = a.v b.d + b.v a.d"""
        a = self.a
        t0, t1 = VD[5], VD[6]
        for j in (1, 2, 3):
            a(f"\tv_mul_f32 {t0}, {A[0]}, {B[j]}\n\tv_mul_f32 {t1}, {B[0]}, {A[j]}\n\tv_add_f32 {R[j]}, {t0}, {t1}")
        a(f"\tv_mul_f32 {R[0]}, {A[0]}, {B[0]}")

    def gr_div(self, A, B, R):
        """dev_ops.hpp gr_div: d = b.v b.v; v = a.v / b.v; dk = (b.v a.dk - a.v b.dk) / d.  R must not be A or B."""
        a = self.a
        sq, t0, t1 = VD[5], VD[6], VD[7]
        a(f"\tv_mul_f32 {sq}, {B[0]}, {B[0]}")
        for j in (1, 2, 3):
            a(f"\tv_mul_f32 {t0}, {B[0]}, {A[j]}\n\tv_mul_f32 {t1}, {A[0]}, {B[j]}\n\tv_sub_f32 {t0}, {t0}, {t1}")
            self.div1(t0, sq, R[j])
        self.div1(A[0], B[0], R[0])

    def imm_gr(self, dst):
        """dst = gr1(immediate of the op)"""
        self.a(f"\tv_mov_b32 {dst[0]}, {S_W1}\n\tv_mov_b32 {dst[1]}, 0\n\tv_mov_b32 {dst[2]}, 0\n\tv_mov_b32 {dst[3]}, 0")

    def gr1(self, R, v):
        """R = gr1(v): v may be R[0]"""
        self.a(f"\tv_mov_b32 {R[0]}, {v}\n\tv_mov_b32 {R[1]}, 0\n\tv_mov_b32 {R[2]}, 0\n\tv_mov_b32 {R[3]}, 0")

    def operands(self, form):
        """A, B in VT / VU in the op's order (register operand a, second register or immediate); index mode off afterwards"""
        self.read_a(VT)
        if form == "RR":
            self.read_b(VU)
        self.idx_off()
        if form != "RR":
            self.imm_gr(VU)
        return (VT, VU) if form != "IR" else (VU, VT)

    # ---- handlers ---------------------------------------------------------------------------------------
    def handler(self, op, inplace=False):
        a, F = self.a, self.F
        if op == "OUTPUT":             # a shape tape has one output; its gradient stays in VRES for the caller
            self.read_a(VRES)
            return self.ret()
        if op == "INPUT":
            return self.out_of_line("input", self.h_input_grad)
        if op == "COPY_REG":
            self.read_a(VT)
            return self.write_out(VT)
        if op == "COPY_IMM":
            self.idx_on(S_OUT, DST)
            a(f"\tv_mov_b32 {F(0)}, {S_W1}\n\tv_mov_b32 {F(1)}, 0\n\tv_mov_b32 {F(2)}, 0\n\tv_mov_b32 {F(3)}, 0")
            return self.ret()
        if op == "NEG":
            self.idx_on(S_A, SRC1)
            for j in range(4):
                a(f"\tv_xor_b32 {VT[j]}, {S_SIGN}, {F(j)}")
            return self.write_out(VT)
        if op in ("ABS", "SQUARE", "RECIP", "SQRT", "FLOOR", "CEIL", "ROUND", "NOT"):
            def body(op=op):
                self.read_a(VT)
                self.idx_off()
                if op == "ABS":                 # a.v < 0 ? -a : a
                    a(f"\tv_cmp_gt_f32 vcc, 0, {VT[0]}")
                    for j in range(4):
                        a(f"\tv_xor_b32 {VW[j]}, {S_SIGN}, {VT[j]}")
                    for j in range(4):
                        a(f"\tv_cndmask_b32 {VW[j]}, {VT[j]}, {VW[j]}, vcc")
                elif op == "SQUARE":            # gr_mul(a, a): both products of a derivative are the same number
                    for j in (1, 2, 3):
                        a(f"\tv_mul_f32 {VD[5]}, {VT[0]}, {VT[j]}\n\tv_add_f32 {VW[j]}, {VD[5]}, {VD[5]}")
                    a(f"\tv_mul_f32 {VW[0]}, {VT[0]}, {VT[0]}")
                elif op == "RECIP":             # gr_div(gr1(1), a)
                    self.gr1(VU, "1.0")
                    self.gr_div(VU, VT, VW)
                elif op == "SQRT":              # v = sqrtf(a.v); d / (2 v)
                    self.sqrt1(VT[0], VW[0])
                    a(f"\tv_add_f32 {VU[0]}, {VW[0]}, {VW[0]}")
                    for j in (1, 2, 3):
                        self.div1(VT[j], VU[0], VW[j])
                elif op == "NOT":
                    a(f"\tv_cmp_eq_f32 vcc, 0, {VT[0]}\n\ts_nop 1\n\tv_cndmask_b32 {VW[0]}, 0, 1.0, vcc")
                    self.gr1(VW, VW[0])
                else:
                    if op == "ROUND":           # roundf: half away from zero (Interp.f_round)
                        d = VD
                        a(f"""
	v_trunc_f32 {d[0]}, {VT[0]}
	v_sub_f32 {d[1]}, {VT[0]}, {d[0]}
	v_cmp_ge_f32_e64 {S_M[0]}, |{d[1]}|, 0.5
	s_nop 1
	v_cndmask_b32_e64 {d[1]}, 0, 1.0, {S_M[0]}
	v_bfi_b32 {d[1]}, {S_ABSM}, {d[1]}, {VT[0]}
	v_add_f32 {VW[0]}, {d[0]}, {d[1]}""")
                    else:
                        a(f"\t{'v_floor_f32' if op == 'FLOOR' else 'v_ceil_f32'} {VW[0]}, {VT[0]}")
                    self.gr1(VW, VW[0])
                self.write_out(VW)
            return self.out_of_line(op.lower(), body)
        if "_" not in op:
            if not self.trans:
                return self.ret()          # transcendental / rng: never reached (the host keeps those tapes on the C++ kernel)
            def body(op=op):               # dev_ops.hpp GRAD::unary, FULL
                self.read_a(VT)
                self.idx_off()
                v, c = VT[0], VU[0]
                if op == "SIN":
                    self.call("cos", v, c)
                    self.call("sin", v, VW[0])
                    for j in (1, 2, 3):
                        a(f"\tv_mul_f32 {VW[j]}, {VT[j]}, {c}")
                elif op == "COS":
                    self.call("sin", v, c)
                    a(f"\tv_xor_b32 {c}, {S_SIGN}, {c}")
                    self.call("cos", v, VW[0])
                    for j in (1, 2, 3):
                        a(f"\tv_mul_f32 {VW[j]}, {VT[j]}, {c}")
                elif op == "TAN":
                    self.call("cos", v, c)
                    a(f"\tv_mul_f32 {c}, {c}, {c}")
                    self.call("tan", v, VW[0])
                    for j in (1, 2, 3):
                        self.div1(VT[j], c, VW[j])
                elif op in ("ASIN", "ACOS"):
                    a(f"\tv_mul_f32 {VU[1]}, {v}, {v}\n\tv_sub_f32 {VU[1]}, 1.0, {VU[1]}")
                    self.sqrt1(VU[1], c)
                    self.call(op.lower(), v, VW[0])
                    for j in (1, 2, 3):
                        if op == "ACOS":
                            a(f"\tv_xor_b32 {VU[1]}, {S_SIGN}, {VT[j]}")
                            self.div1(VU[1], c, VW[j])
                        else:
                            self.div1(VT[j], c, VW[j])
                elif op == "ATAN":
                    a(f"\tv_mul_f32 {c}, {v}, {v}\n\tv_add_f32 {c}, 1.0, {c}")
                    self.call("atan", v, VW[0])
                    for j in (1, 2, 3):
                        self.div1(VT[j], c, VW[j])
                elif op == "EXP":
                    self.call("exp", v, VW[0])
                    for j in (1, 2, 3):
                        a(f"\tv_mul_f32 {VW[j]}, {VW[0]}, {VT[j]}")
                elif op == "LN":
                    self.call("ln", v, VW[0])
                    for j in (1, 2, 3):
                        self.div1(VT[j], v, VW[j])
                else:                       # RAND: gr1(f_rand(a.v)) - bits (hash >> 9) | 1.0, minus 1 (rng/mod.rs:19-23)
                    self.pcg_consts()
                    self.pcg(v, VW[0])
                    a(f"\tv_lshrrev_b32 {VW[0]}, 9, {VW[0]}\n\tv_or_b32 {VW[0]}, 0x3f800000, {VW[0]}\n\tv_add_f32 {VW[0]}, -1.0, {VW[0]}")
                    self.gr1(VW, VW[0])
                self.write_out(VW)
            return self.out_of_line(op.lower(), body)
        base, form = op.rsplit("_", 1)
        if base == "MOD" or (base in ("ATAN2", "MIX") and not self.trans):
            return self.ret()              # (tapes with a modulo keep the C++ kernel: its gradient needs div_euclid)
        if base in ("ATAN2", "MIX"):
            def body(base=base, form=form):
                A, B = self.operands(form)
                if base == "MIX":           # gr1(rng::mix): hash(a + hash(b)) on the bit patterns (rng/mod.rs:30-33)
                    self.pcg_consts()
                    self.pcg(B[0], VW[0])
                    a(f"\tv_add_u32 {VW[0]}, {A[0]}, {VW[0]}")
                    self.pcg(VW[0], VW[0])
                    self.gr1(VW, VW[0])
                else:                       # atan2(y = a, x = b): d = b.v b.v + a.v a.v; (b.v a.dk - a.v b.dk) / d
                    sq, t0, t1 = VD[5], VD[6], VD[7]
                    a(f"\tv_mul_f32 {sq}, {B[0]}, {B[0]}\n\tv_mul_f32 {t0}, {A[0]}, {A[0]}\n\tv_add_f32 {sq}, {sq}, {t0}")
                    for j in (1, 2, 3):
                        a(f"\tv_mul_f32 {t0}, {B[0]}, {A[j]}\n\tv_mul_f32 {t1}, {A[0]}, {B[j]}\n\tv_sub_f32 {t0}, {t0}, {t1}")
                        self.div1(t0, sq, VW[j])
                    self.call("atan2", A[0], VW[0], B[0])
                self.write_out(VW)
            return self.out_of_line(op.lower(), body)
        def body(base=base, form=form):
            A, B = self.operands(form)
            if base == "ADD":
                for j in range(4):
                    a(f"\tv_add_f32 {VW[j]}, {A[j]}, {B[j]}")
            elif base == "SUB":
                for j in range(4):
                    a(f"\tv_sub_f32 {VW[j]}, {A[j]}, {B[j]}")
            elif base == "MUL" and form == "RR":
                self.gr_mul(A, B, VW)
            elif base == "MUL":              # gr_mul_f (vm/mod.rs:1219-1223): every component times the immediate
                for j in range(4):
                    a(f"\tv_mul_f32 {VW[j]}, {A[j]}, {S_W1}")
            elif base == "DIV":
                self.gr_div(A, B, VW)
            elif base in ("MIN", "MAX"):     # either value NaN -> gr1(NaN); else a.v < b.v ? a : b  (a.v > b.v for max)
                cmp = "v_cmp_lt_f32_e64" if base == "MIN" else "v_cmp_gt_f32_e64"
                a(f"\t{cmp} {S_M[0]}, {A[0]}, {B[0]}\n\tv_cmp_u_f32_e64 {S_M[1]}, {A[0]}, {B[0]}\n\ts_nop 0")
                for j in range(4):
                    a(f"\tv_cndmask_b32_e64 {VW[j]}, {B[j]}, {A[j]}, {S_M[0]}")
                a(f"\tv_cndmask_b32_e64 {VW[0]}, {VW[0]}, {V_QNAN}, {S_M[1]}")
                for j in (1, 2, 3):
                    a(f"\tv_cndmask_b32_e64 {VW[j]}, {VW[j]}, 0, {S_M[1]}")
            elif base in ("AND", "OR"):      # and: a.v == 0 ? a : b; or: a.v != 0 ? a : b
                cmp = "v_cmp_eq_f32_e64" if base == "AND" else "v_cmp_neq_f32_e64"
                a(f"\t{cmp} {S_M[0]}, 0, {A[0]}\n\ts_nop 1")
                for j in range(4):
                    a(f"\tv_cndmask_b32_e64 {VW[j]}, {B[j]}, {A[j]}, {S_M[0]}")
            else:                            # COMPARE: gr1(a < b ? -1 : a == b ? 0 : a > b ? 1 : NaN)
                a(f"""
	v_cmp_gt_f32_e64 {S_M[0]}, {A[0]}, {B[0]}
	v_cmp_eq_f32_e64 {S_M[1]}, {A[0]}, {B[0]}
	v_cmp_lt_f32_e64 {S_M[2]}, {A[0]}, {B[0]}
	v_cndmask_b32_e64 {VW[0]}, {V_QNAN}, 1.0, {S_M[0]}
	v_cndmask_b32_e64 {VW[0]}, {VW[0]}, 0, {S_M[1]}
	v_cndmask_b32_e64 {VW[0]}, {VW[0]}, -1.0, {S_M[2]}""")
                self.gr1(VW, VW[0])
            self.write_out(VW)
        return self.out_of_line(op.lower(), body)

    def h_input_grad(self):
        a = self.a
        lab = {k: a.label("gin_" + k) for k in ("x", "y", "z", "done")}
        self.idx_off()
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
        self.idx_on(S_OUT, DST)
        a(f"\tv_mov_b32 {self.F(0)}, {S_T1}\n\tv_mov_b32 {self.F(1)}, 0\n\tv_mov_b32 {self.F(2)}, 0\n\tv_mov_b32 {self.F(3)}, 0")
        a(f"\ts_branch {lab['done']}")
        for axis, G in (("x", GX), ("y", GY), ("z", GZ)):
            a(f"{lab[axis]}:")
            self.idx_on(S_OUT, DST)
            for k in range(2):
                self.pk_mov(self.FP(k), self.P(G, k))
            if axis != "z":
                a(f"\ts_branch {lab['done']}")
        a(f"{lab['done']}:")
        self.ret()


def gen_normals(a, off, trans=None):
    """fh_normals, or with `trans` (the compiled routines' assembly) fh_normals_t: handlers for the transcendental, rng and atan2 opcodes too"""
    name = "fh_normals_t" if trans else "fh_normals"
    o, m = off, S_MAT
    it = GradInterp(a, name + "_g", off, trans=bool(trans))
    nvg = T_BASE + 26 if trans else FILE + NR * 4
    kernel_header(a, name, 32, nvg)
    a(f"""
	s_load_dwordx2 {S_STATE}, {S_KERNARG}, 0x0
	s_load_dwordx4 s[48:51], {S_KERNARG}, 0x8
	s_load_dword s52, {S_KERNARG}, 0x18
	s_mov_b32 {S_WI}, s2""")
    common_consts(a)
    handler_base(a, it)
    a(f"""
	s_waitcnt lgkmcnt(0)
	s_load_dwordx16 s[{m}:{m + 15}], {S_STATE}, {o['P.mat']}
	s_load_dwordx2 s[24:25], {S_STATE}, {o['P.width']}
	s_load_dwordx2 {S_ARENA}, {S_STATE}, {o['arena']}
	s_load_dwordx2 {S_ZBUF}, {S_STATE}, {o['zbuf']}
	s_load_dwordx2 {S_NORMALS}, {S_STATE}, {o['normals']}
	s_load_dwordx2 {S_LEAVES}, {S_STATE}, {o['leaves']}
	s_load_dwordx2 {S_FPLIST}, {S_STATE}, {o['hit_list']}
	s_lshr_b32 {S_NWG}, s48, 6                      ; wave w walks list w % 64 (k_hits3d's buckets) with a stride of n_waves / 64
	s_bfe_i32 {S_SLOTX}, s49, 0x80000               ; (s0 / s1 held the kernarg pointer until here)
	s_bfe_i32 {S_SLOTY}, s49, 0x80008
	s_bfe_i32 {S_SLOTZ}, s49, 0x80010
	s_mov_b32 {S_ZLO}, s50
	s_mov_b32 {S_ZHI}, s51
	s_waitcnt lgkmcnt(0)
	s_and_b32 {S_T0}, {S_WI}, 63
	s_lshr_b32 {S_WI}, {S_WI}, 6
	s_lshl_b32 {S_T1}, {S_T0}, 8                    ; the list's counter: FH_HIT_STRIDE words apart
	s_add_u32 s86, s38, {S_T1}
	s_addc_u32 s87, s39, 0
	s_load_dword {S_NFP}, {S_PC}, 0x0
	s_mul_i32 {S_T0}, {S_T0}, s52                   ; its entries: behind the 64 counters, `bucket_cap` (kernarg) per list
	s_lshl_b32 {S_T0}, {S_T0}, 2
	s_add_u32 {S_T0}, {S_T0}, {64 * 64 * 4}
	s_add_u32 s38, s38, {S_T0}
	s_addc_u32 s39, s39, 0
	s_waitcnt lgkmcnt(0)
	s_min_u32 {S_NFP}, {S_NFP}, s52
.L{name}_next:
	; ---- next leaf of the list (k_hits3d: the slab's leaves that own a hit): static round robin over the waves --------------------
	s_cmp_ge_u32 {S_WI}, {S_NFP}
	s_cbranch_scc1 .L{name}_exit
	s_lshl_b32 {S_T0}, {S_WI}, 2
	s_add_u32 {S_WI}, {S_WI}, {S_NWG}
	s_add_u32 s86, s38, {S_T0}
	s_addc_u32 s87, s39, 0
	s_load_dword {S_CUR}, {S_PC}, 0x0
	v_and_b32 {V_PX}, 7, {V_LANE}
	v_lshrrev_b32 {V_PY}, 3, {V_LANE}
	s_waitcnt lgkmcnt(0)
	s_sub_u32 {S_T0}, {S_CUR}, 1
	s_mul_hi_u32 {S_T1}, {S_T0}, {o['sizeof_leaf']}
	s_mul_i32 {S_T0}, {S_T0}, {o['sizeof_leaf']}
	s_add_u32 s86, s34, {S_T0}
	s_addc_u32 s87, s35, {S_T1}
	s_load_dwordx4 s[48:51], {S_PC}, 0x0                 ; FhLeaf: tape offset, length, registers | choices, corner x  (the interpreter's
	s_load_dword s52, {S_PC}, 0x10                       ; corner y                                       op batches: free between two tapes)
	s_waitcnt lgkmcnt(0)
	v_add_u32 {V_PX}, s51, {V_PX}
	v_add_u32 {V_PY}, s52, {V_PY}
	v_cmp_gt_u32_e64 {S_M[0]}, {S_WIDTH}, {V_PX}
	v_cmp_gt_u32_e64 {S_M[1]}, {S_HEIGHT}, {V_PY}
	v_mul_u32_u24 {V_NOFF}, {V_PY}, {S_WIDTH}
	v_add_u32 {V_NOFF}, {V_NOFF}, {V_PX}                  ; pixel index
	v_lshlrev_b32 {V_PIX}, 3, {V_NOFF}
	v_mul_u32_u24 {V_NOFF}, 12, {V_NOFF}
	v_mov_b32 {V_ID}, 0
	v_mov_b32 {V_DEPTH}, 0
	s_and_b64 {S_M[0]}, {S_M[0]}, {S_M[1]}
	s_mov_b64 {S_SAVE}, exec
	s_mov_b64 exec, {S_M[0]}
	global_load_dwordx2 v[2:3], {V_PIX}, {S_ZBUF}
	s_mov_b64 exec, {S_SAVE}
	s_waitcnt vmcnt(0)
	; the leaf's hits of this launch: its number still in the pixel's word, z_lo < depth <= z_hi
	v_cmp_lt_u32_e64 {S_M[0]}, {S_ZLO}, {V_DEPTH}
	v_cmp_ge_u32_e64 {S_M[1]}, {S_ZHI}, {V_DEPTH}
	v_cmp_eq_u32_e64 {S_M[2]}, {S_CUR}, {V_ID}
	s_nop 1
	s_and_b64 {S_M[0]}, {S_M[0]}, {S_M[1]}
	s_and_b64 {S_TODO}, {S_M[0]}, {S_M[2]}
	s_cmp_eq_u64 {S_TODO}, 0
	s_cbranch_scc1 .L{name}_next
	; ---- the lanes' input gradients: xf_grad of {{x,1,0,0}}, {{y,0,1,0}}, {{z,0,0,1}} at the voxel above the hit (z = depth - 1) --------
	v_cvt_f32_u32 {V_PX}, {V_PX}
	v_cvt_f32_u32 {V_PY}, {V_PY}
	v_add_u32 {V_PZ}, -1, {V_DEPTH}
	v_cvt_f32_u32 {V_PZ}, {V_PZ}""")
    # r[i] = gr_add(gr_add(gr_add(gr_mul_f(x, m[4i]), gr_mul_f(y, m[4i+1])), gr_mul_f(z, m[4i+2])), gr1(m[4i+3]))
    # component k of gr_mul_f(x, c) is x_k * c with x = {px, 1, 0, 0}: every product and sum is made, as the C++ makes them
    t0, t1 = VD[5], VD[6]
    for i in range(4):
        R = XR[i]
        c = [f"s{m + 4 * i + k}" for k in range(4)]
        for k, (xs, ys, zs) in enumerate(((V_PX, V_PY, V_PZ), ("1.0", "0", "0"), ("0", "1.0", "0"), ("0", "0", "1.0"))):
            a(f"\tv_mov_b32 {t1}, {c[0]}\n\tv_mul_f32 {t0}, {xs}, {t1}" if k else f"\tv_mul_f32 {t0}, {c[0]}, {xs}")
            a(f"\tv_mov_b32 {t1}, {c[1]}\n\tv_mul_f32 {t1}, {ys}, {t1}" if k else f"\tv_mul_f32 {t1}, {c[1]}, {ys}")
            a(f"\tv_add_f32 {t0}, {t0}, {t1}")
            a(f"\tv_mov_b32 {t1}, {c[2]}\n\tv_mul_f32 {t1}, {zs}, {t1}" if k else f"\tv_mul_f32 {t1}, {c[2]}, {zs}")
            a(f"\tv_add_f32 {t0}, {t0}, {t1}")
            a(f"\tv_add_f32 {R[k]}, {c[3] if k == 0 else '0'}, {t0}")
    for i, G in enumerate((GX, GY, GZ)):
        it.gr_div(XR[i], XR[3], G)
    a(f"""
	; ---- the leaf's tape for all 64 lanes --------------------------------------------------------------------------------------
	s_mov_b32 {S_LEN}, s49
	s_mov_b32 s49, 0
	s_lshl_b64 s[48:49], s[48:49], 3
	s_add_u32 s44, s48, s30
	s_addc_u32 s45, s49, s31""")
    call_interp(a, it)
    a(f"""
	; the lanes this leaf hit: normal = (dx, dy, dz), z-buffer word = depth << 32 (normal done)
	s_mov_b64 {S_SAVE}, exec
	s_mov_b64 exec, {S_TODO}
	v_mov_b32 {V_ID}, 0
	global_store_dword {V_NOFF}, v11, {S_NORMALS}              ; (VGPR tuples start at even registers here: dx, then dy dz)
	global_store_dwordx2 {V_NOFF}, v[12:13], {S_NORMALS} offset:4
	global_store_dword {V_PIX}, {V_ID}, {S_ZBUF}
	s_mov_b64 exec, {S_SAVE}
	s_branch .L{name}_next
.L{name}_exit:
	s_waitcnt vmcnt(0)""")
    kernel_footer(a, name, 32, nvg, 102, True)
    if trans:
        import gen_trans
        gen_trans.embed(a, trans, v_base=T_BASE, prefix="fh_tn_")
    it.emit()
    return name, 32, nvg, [(8, "global_buffer")] + [(4, "by_value")] * 6
