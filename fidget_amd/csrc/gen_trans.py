"""Embeds the compiled transcendental routines (trans_funcs.hip -> LLVM IR -> llc, fidget_amd.build) in the interpreters' code object.

Each function body is taken from the compiler's assembly, its registers moved into the window the interpreters keep free
while a handler runs (v0..v25 -> v128..v153, s0..s9 -> s86..s95, return address s[30:31] -> s[96:97]; vcc and exec are used
as they are: exec is restored by the functions themselves), its local labels made unique.  Calling convention for the
handlers: argument(s) in v128 (, v129), result in v128, `s_getpc / s_add / s_branch` with the return address in s[96:97].
The routines' tables (trans_libm.hpp MemTables: 2^(i/32), logf's 1/c and log c, the bits of 4/pi) are loaded pc-relative from
.rodata: `tables()` emits them once per code object under the names the renamed code refers to."""
import re

V_BASE, S_BASE, S_RET = 128, 86, 96
FUNCS = ["sin", "cos", "tan", "asin", "acos", "atan", "exp", "ln", "atan2", "mod"]
FUNCS4 = ["sin4", "cos4", "exp4", "ln4"]     # four samples per call (v0..v3 in and out): embedded with a window of WIDE_V registers
MAX_V, MAX_S, WIDE_V = 26, 10, 64


_TAB_SYM = re.compile(r"_ZZN4fhlm9MemTables\d+(\w+?)EjE1T")   # function-local `static const T[]` of fhlm::MemTables::<name>


def tables(a, path):
    """the routines' constant tables as `fh_tab_<name>` in .rodata (once per code object, before any embed())"""
    txt = open(path).read()
    a("\t.section\t.rodata,\"a\",@progbits")
    found = 0
    for m in re.finditer(r"^(_ZZN4fhlm9MemTables\d+\w+?EjE1T):\n((?:\t\.(?:quad|long)\t[^\n]*\n)+)", txt, re.M):
        a(f"\t.p2align 4\n{_TAB_SYM.sub(lambda t: 'fh_tab_' + t.group(1), m.group(1))}:")
        a("\n".join(l.split(";")[0].rstrip() for l in m.group(2).rstrip("\n").split("\n")))
        found += 1
    assert found == 4, found
    a("\t.text")


def _rename(body, name, V_BASE=V_BASE, prefix="fh_t_", s_map=None, MAX_V=MAX_V):
    def compact(n):
        """the compiler leaves the callee-saved blocks v40..v47, v56..v63, v72..v79, ... alone; the window has no such gaps (blocks
        of eight move as a whole: 64-bit operands stay even-aligned)"""
        b = n // 8
        assert b < 5 or b % 2 == 0, (name, n)      # a callee-saved register would have been spilled to scratch
        return n - 8 * len([x for x in range(5, b) if x % 2])

    def v1(m):
        n = compact(int(m.group(1)))
        assert n < MAX_V, (name, m.group(0))
        return f"v{V_BASE + n}"

    def v2(m):
        a, b = compact(int(m.group(1))), compact(int(m.group(2)))
        assert b < MAX_V and b - a == int(m.group(2)) - int(m.group(1)), (name, m.group(0))
        return f"v[{V_BASE + a}:{V_BASE + b}]"

    def smap(n):
        if n in (30, 31):
            return S_RET + (n - 30)
        assert n < MAX_S, (name, n)
        return S_BASE + n if s_map is None else s_map[n]

    def s1(m):
        return f"s{smap(int(m.group(1)))}"

    def s2(m):
        a, b = int(m.group(1)), int(m.group(2))
        assert smap(b) - smap(a) == b - a
        return f"s[{smap(a)}:{smap(b)}]"

    out = []
    for line in body.split("\n"):
        code = line.split(";")[0].rstrip()
        if not code.strip() or code.strip().startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", code.strip()):
                out.append(re.sub(r"\.LBB(\d+)_(\d+)", rf".L{prefix}{name}_bb\2", code.strip()))
            continue
        code = re.sub(r"\.LBB(\d+)_(\d+)", rf".L{prefix}{name}_bb\2", code)
        sym = _TAB_SYM.search(code)
        if sym:   # `s_add_u32 s4, s4, <table>@rel32@lo+4`: keep the symbol out of the register renaming
            code = code.replace(sym.group(0), "@TAB@")
        code = re.sub(r"\bv\[(\d+):(\d+)\]", v2, code)
        code = re.sub(r"\bv(\d+)\b", v1, code)
        code = re.sub(r"\bs\[(\d+):(\d+)\]", s2, code)
        code = re.sub(r"\bs(\d+)\b", s1, code)
        for bad in ("scratch", "buffer_", "s_swappc", "ds_", "global_store", "global_atomic", "flat_", "m0", "v_writelane", "v_readlane"):
            assert bad not in code, (name, code)
        if sym:
            code = code.replace("@TAB@", "fh_tab_" + sym.group(1))
        out.append(code)
    return "\n".join(out)


# ---- expf for two samples at a time, written by hand and part of the EXP handlers themselves (the leaf kernel of tapes with transcendental opcodes: a tenth of bear.vm's frame
# was the compiled exp4 - 167 instructions per call, 42 per sample - and its window of 64 registers kept the kernel at two waves
# per SIMD) ----------------------------------------------------------------------------------------------------------------------
# The same operations as trans_libm.hpp expf_main_ (glibc e_expf.c: two fused operations for k and r, the table value times a degree-3
# polynomial, all in binary64), 13 instructions per sample in 22 registers of the window:
#  - the table 2^(i/32) lives in TWO VGPRs of the wave (lane l: entry l % 32; EXP_TAB below, loaded once per wave by exp_table_init): a
#    sample's entry comes by two ds_bpermute_b32 instead of a 64-bit load from memory whose latency every call waited for;
#  - `t + (ki << 47)` only touches the table value's high word: (ki << 47) has no low word, so it is one v_lshl_add_u32 of the low word of
#    kd's bit pattern (bits 0..16 of ki reach bits 47..63);
#  - an instruction takes one operand from scalar registers (gfx9: one constant-bus operand): of the pairs (InvLn2N, SHIFT) and (C0, C1)
#    that meet in one fused operation each, SHIFT and C1 are vector pairs set up per call.
# |x| >= 88 (glibc's special cases: overflow, underflow, infinities) in any lane of any sample of the op: the compiled one-sample
# routine for each.  NaN arguments take the main path and come out NaN (the class is what is modelled, as everywhere).
EXP_TAB = 22          # window registers v<base + 22>, v<base + 23>: the table's low / high words


def _f64(x):
    import struct
    b = struct.unpack("<Q", struct.pack("<d", float.fromhex(x) if isinstance(x, str) else x))[0]
    return b & 0xFFFFFFFF, b >> 32


def exp2_table():
    """the 32 values of MemTables::exp2_tab (trans_libm.hpp, from glibc's e_exp2f_data.c)"""
    import os
    txt = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "trans_libm.hpp")).read()
    m = re.search(r"exp2_tab\(uint32_t i\) \{\s*static const uint64_t T\[32\] = \{(.*?)\};", txt, re.S)
    vals = [int(v, 16) for v in re.findall(r"0x[0-9a-fA-F]+", m.group(1))]
    assert len(vals) == 32 and vals[0] == 0x3ff0000000000000
    return vals


def exp_table_init(a, v_base, prefix="fh_t_", lane="v0"):
    """loads the table registers of the hand-written expf and logf (once per wave; clobbers s86..s89; `lane` = the lane's number)"""
    here = a.label("exptab")
    lo, hi = v_base + EXP_TAB, v_base + EXP_TAB + 1
    a(f"""
	s_getpc_b64 s[86:87]
{here}:
	s_mov_b32 s88, {prefix}exp2_tab - {here}
	s_ashr_i32 s89, s88, 31
	s_add_u32 s86, s86, s88
	s_addc_u32 s87, s87, s89
	v_and_b32 v{lo}, 31, {lane}
	v_lshlrev_b32 v{lo}, 3, v{lo}
	global_load_dwordx2 v[{lo}:{hi}], v{lo}, s[86:87]
	v_and_b32 v{v_base + LN_TAB}, 15, {lane}
	v_lshlrev_b32 v{v_base + LN_TAB}, 4, v{v_base + LN_TAB}
	global_load_dwordx4 v[{v_base + LN_TAB}:{v_base + LN_TAB + 3}], v{v_base + LN_TAB}, s[86:87] offset:256
	s_waitcnt vmcnt(0)""")


def exp_consts(a, vb):
    """the constants of exp_pair: s[86:87] InvLn2N, s[88:89] C0, s[90:91] C2, s92 88.0f; v[vb:vb+1] SHIFT, v[vb+2:vb+3] C1"""
    inv, c0, c2 = _f64("0x1.71547652b82fep+5"), _f64("0x1.c6af84b912394p-20"), _f64("0x1.62e42ff0c52d6p-6")
    sh, c1 = _f64("0x1.8p+52"), _f64("0x1.ebfce50fac4f3p-13")
    a(f"""
	s_mov_b32 s92, 0x42b00000                          ; 88.0
	s_mov_b32 s86, {inv[0]:#x}
	s_mov_b32 s87, {inv[1]:#x}
	s_mov_b32 s88, {c0[0]:#x}
	s_mov_b32 s89, {c0[1]:#x}
	s_mov_b32 s90, {c2[0]:#x}
	s_mov_b32 s91, {c2[1]:#x}
	v_mov_b32 v{vb}, {sh[0]:#x}
	v_mov_b32 v{vb + 1}, {sh[1]:#x}
	v_mov_b32 v{vb + 2}, {c1[0]:#x}
	v_mov_b32 v{vb + 3}, {c1[1]:#x}""")


def exp_special(a, vb, xs, slow):
    """branches to `slow` when any of the samples xs is one of expf's special cases (|x| >= 88; after exp_consts; scratch v<vb+4>, vcc)"""
    t = f"v{vb + 4}"
    ab = [f"|{x}|" for x in xs]
    if len(ab) == 2:
        a(f"	v_max_f32_e64 {t}, {ab[0]}, {ab[1]}")
    else:
        a(f"	v_max3_f32 {t}, {ab[0]}, {ab[1]}, {ab[2]}")
        k = 3
        while len(ab) - k >= 2:
            a(f"	v_max3_f32 {t}, {t}, {ab[k]}, {ab[k + 1]}")
            k += 2
        if k < len(ab):
            a(f"	v_max_f32_e64 {t}, {ab[k]}, {t}")
    a(f"	v_cmp_ngt_f32 vcc, s92, {t}\n	s_cbranch_vccnz {slow}")


def exp_pair(a, vb, xin, xout):
    """xout[j] = expf(xin[j]), j = 0, 1, for arguments that are not special (exp_special); window registers v<vb+4>..v<vb+21>, the
    constants of exp_consts, the table registers of exp_table_init"""
    pair = lambda r: f"v[{r}:{r + 1}]"
    SH, C1 = vb, vb + 2
    A, B, E, D = ([vb + 4 + 8 * j + 2 * k for j in range(2)] for k in range(4))     # per sample: xd / z, kd / r / y, kd - SHIFT / r2, the table value
    K = [vb + 20, vb + 21]                                                          # ... and the low word of kd's bits
    TLO, THI = f"v{vb + EXP_TAB}", f"v{vb + EXP_TAB + 1}"
    R2 = range(2)
    for j in R2:
        a(f"	v_cvt_f64_f32 {pair(A[j])}, {xin[j]}")
    for j in R2:
        a(f"	v_fma_f64 {pair(B[j])}, s[86:87], {pair(A[j])}, {pair(SH)}")                 # kd = InvLn2N x + SHIFT
    for j in R2:
        a(f"	v_lshlrev_b32 v{D[j] + 1}, 2, v{B[j]}")
        a(f"	ds_bpermute_b32 v{D[j]}, v{D[j] + 1}, {TLO}")
        a(f"	ds_bpermute_b32 v{D[j] + 1}, v{D[j] + 1}, {THI}")
    for j in R2:
        a(f"	v_add_f64 {pair(E[j])}, {pair(B[j])}, -{pair(SH)}")                           # kd - SHIFT
        a(f"	v_mov_b32 v{K[j]}, v{B[j]}")
    for j in R2:
        a(f"	v_fma_f64 {pair(B[j])}, s[86:87], {pair(A[j])}, -{pair(E[j])}")               # r = InvLn2N x - kd
    for j in R2:
        a(f"	v_fma_f64 {pair(A[j])}, s[88:89], {pair(B[j])}, {pair(C1)}")                  # z = C0 r + C1
        a(f"	v_mul_f64 {pair(E[j])}, {pair(B[j])}, {pair(B[j])}")                          # r2
    for j in R2:
        a(f"	v_fma_f64 {pair(B[j])}, s[90:91], {pair(B[j])}, 1.0")                         # y = C2 r + 1
    for j in R2:
        a(f"	v_fma_f64 {pair(B[j])}, {pair(A[j])}, {pair(E[j])}, {pair(B[j])}")            # y = z r2 + y
    a("	s_waitcnt lgkmcnt(0)")
    for j in R2:
        a(f"	v_lshl_add_u32 v{D[j] + 1}, v{K[j]}, 15, v{D[j] + 1}")                        # s = T[ki % 32] + (ki << 47)
    for j in R2:
        a(f"	v_mul_f64 {pair(B[j])}, {pair(B[j])}, {pair(D[j])}")
    for j in R2:
        a(f"	v_cvt_f32_f64 {xout[j]}, {pair(B[j])}")


def exp_table(a, prefix):
    a(f"	.p2align 3\n{prefix}exp2_tab:")
    for v in exp2_table():
        a(f"	.quad {v:#x}")
    for invc, logc in ln_table():       # (256 bytes behind: logf's {1/c, log c})
        a(f"	.quad {invc:#x}, {logc:#x}")


# ---- logf for two samples at a time, by hand, inside the LN handlers (the compiled one-sample routine: 59 instructions and a call per
# sample; bear.vm's 8 ln ops were a sixth of its leaf kernel's instructions) ------------------------------------------------------
# trans_libm.hpp logf_main_ (glibc e_logf.c) operation for operation, 19 instructions per sample: the 16-entry table {1/c, log c} in
# FOUR VGPRs of the wave (lane l: entry l % 16; LN_TAB, loaded once per wave by exp_table_init), an entry by four ds_bpermute_b32 whose
# address is tmp >> 17 - bits 19..22 of tmp land on the lane-select bits; k, iz, z as the library computes them.  Special arguments
# (x < 2^-126 - zero, negative, subnormal -, +inf, NaN: ix - 0x00800000 >= 0x7f000000) in any lane of any sample of the op: the compiled
# routine, sample by sample.  x = 1 needs no case of its own here: the main path returns +0 for it in round-to-nearest.
LN_TAB = 24           # window registers v<base + 24 .. 27>: 1/c low / high, log c low / high
EXP2_WINDOW = 28


def ln_table():
    """MemTables::log_invc / log_logc of trans_libm.hpp (glibc's e_logf_data.c) as 16 x (invc, logc) bit patterns"""
    import os, struct
    txt = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "trans_libm.hpp")).read()
    out = []
    for name in ("log_invc", "log_logc"):
        m = re.search(name + r"\(uint32_t i\) \{\s*static const double T\[16\] = \{(.*?)\};", txt, re.S)
        vals = [struct.unpack("<Q", struct.pack("<d", float.fromhex(v)))[0] for v in re.findall(r"-?0x[0-9a-fA-F.]+p[-+]?\d+", m.group(1))]
        assert len(vals) == 16, (name, len(vals))
        out.append(vals)
    assert out[0][9] == 0x3ff0000000000000 and out[1][9] == 0
    return list(zip(*out))


def ln_consts(a, vb):
    """the constants of ln_pair: s[86:87] Ln2, s[88:89] A1, s[90:91] A0, s92 / s93 the bounds of an ordinary argument; v[vb:vb+1] A2"""
    ln2, a1, a0, a2 = _f64("0x1.62e42fefa39efp-1"), _f64("0x1.5575b0be00b6ap-2"), _f64("-0x1.00ea348b88334p-2"), _f64("-0x1.ffffef20a4123p-2")
    a(f"""
	s_mov_b32 s92, 0x00800000
	s_mov_b32 s93, 0x7f800000
	s_mov_b32 s86, {ln2[0]:#x}
	s_mov_b32 s87, {ln2[1]:#x}
	s_mov_b32 s88, {a1[0]:#x}
	s_mov_b32 s89, {a1[1]:#x}
	s_mov_b32 s90, {a0[0]:#x}
	s_mov_b32 s91, {a0[1]:#x}
	v_mov_b32 v{vb}, {a2[0]:#x}
	v_mov_b32 v{vb + 1}, {a2[1]:#x}""")


def ln_special(a, vb, xs, slow):
    """branches to `slow` unless 2^-126 <= x < inf for every sample xs in every lane (their bit patterns as unsigned numbers; after
    ln_consts; scratch v<vb+2>, v<vb+3>, vcc)"""
    lo, hi = f"v{vb + 2}", f"v{vb + 3}"
    if len(xs) == 2:
        a(f"	v_min_u32 {lo}, {xs[0]}, {xs[1]}\n	v_max_u32 {hi}, {xs[0]}, {xs[1]}")
    else:
        a(f"	v_min3_u32 {lo}, {xs[0]}, {xs[1]}, {xs[2]}\n	v_max3_u32 {hi}, {xs[0]}, {xs[1]}, {xs[2]}")
        k = 3
        while len(xs) - k >= 2:
            a(f"	v_min3_u32 {lo}, {lo}, {xs[k]}, {xs[k + 1]}\n	v_max3_u32 {hi}, {hi}, {xs[k]}, {xs[k + 1]}")
            k += 2
        if k < len(xs):
            a(f"	v_min_u32 {lo}, {xs[k]}, {lo}\n	v_max_u32 {hi}, {xs[k]}, {hi}")
    a(f"""
	v_cmp_gt_u32 vcc, s92, {lo}
	s_cbranch_vccnz {slow}
	v_cmp_le_u32 vcc, s93, {hi}
	s_cbranch_vccnz {slow}""")


def ln_pair(a, vb, xin, xout):
    """xout[j] = logf(xin[j]), j = 0, 1, for ordinary arguments (ln_special); window registers v<vb+2>..v<vb+21>, the constants of
    ln_consts, the table registers of exp_table_init"""
    pair = lambda r: f"v[{r}:{r + 1}]"
    A2 = vb
    base = [vb + 2 + 10 * j for j in range(2)]
    P1, P2, P3, P4 = ([b + 2 * k for b in base] for k in range(4))     # 1/c, r2 | log c, y | k, y0, y0 + r | z, r
    TMP, S = [b + 8 for b in base], [b + 9 for b in base]
    T = [f"v{vb + LN_TAB + k}" for k in range(4)]
    R2 = range(2)
    for j in R2:
        a(f"	v_add_u32 v{TMP[j]}, 0xc0cd0000, {xin[j]}")                                   # tmp = ix - 0x3f330000
        a(f"	v_lshrrev_b32 v{S[j]}, 17, v{TMP[j]}")                                        # lane = (tmp >> 19) % 64 -> entry (tmp >> 19) % 16
        a(f"	ds_bpermute_b32 v{P1[j]}, v{S[j]}, {T[0]}\n	ds_bpermute_b32 v{P1[j] + 1}, v{S[j]}, {T[1]}")
        a(f"	ds_bpermute_b32 v{P2[j]}, v{S[j]}, {T[2]}\n	ds_bpermute_b32 v{P2[j] + 1}, v{S[j]}, {T[3]}")
    for j in R2:
        a(f"	v_ashrrev_i32 v{S[j]}, 23, v{TMP[j]}")                                        # k
        a(f"	v_and_b32 v{TMP[j]}, 0xff800000, v{TMP[j]}")
    for j in R2:
        a(f"	v_cvt_f64_i32 {pair(P3[j])}, v{S[j]}")
        a(f"	v_sub_u32 v{S[j]}, {xin[j]}, v{TMP[j]}")                                      # iz = ix - (tmp & 0xff800000)
    for j in R2:
        a(f"	v_cvt_f64_f32 {pair(P4[j])}, v{S[j]}")                                        # z
    a("	s_waitcnt lgkmcnt(0)")
    for j in R2:
        a(f"	v_fma_f64 {pair(P4[j])}, {pair(P4[j])}, {pair(P1[j])}, -1.0")                 # r = z / c - 1
        a(f"	v_fma_f64 {pair(P3[j])}, {pair(P3[j])}, s[86:87], {pair(P2[j])}")             # y0 = k Ln2 + log c
    for j in R2:
        a(f"	v_mul_f64 {pair(P1[j])}, {pair(P4[j])}, {pair(P4[j])}")                       # r2
        a(f"	v_fma_f64 {pair(P2[j])}, s[88:89], {pair(P4[j])}, {pair(A2)}")                # y = A1 r + A2
    for j in R2:
        a(f"	v_fma_f64 {pair(P2[j])}, s[90:91], {pair(P1[j])}, {pair(P2[j])}")             # y = A0 r2 + y
        a(f"	v_add_f64 {pair(P3[j])}, {pair(P3[j])}, {pair(P4[j])}")                       # y0 + r
    for j in R2:
        a(f"	v_fma_f64 {pair(P2[j])}, {pair(P2[j])}, {pair(P1[j])}, {pair(P3[j])}")        # y r2 + (y0 + r)
    for j in R2:
        a(f"	v_cvt_f32_f64 {xout[j]}, {pair(P2[j])}")


# ---- sinf / cosf for two samples at a time, by hand, inside the SIN / COS handlers (the compiled four-sample routines: 424 instructions
# per call, 106 per sample - bear.vm's six such ops were a fifth of its leaf kernel's instructions) ------------------------------------
# trans_libm.hpp sincosf_ (glibc s_sinf.c / s_cosf.c, the fast reduction |y| < 120) operation for operation, 32 instructions per sample:
# n and x = y - n pi/2 in binary64, the sine polynomial's value, the cosine polynomial's value, the one the quadrant asks for, the
# small-argument answer (y, 1) by a select.  |y| >= 120 or infinite in any lane of any sample of the op: the compiled routine, sample
# by sample.  A NaN argument comes out NaN of the main path (n = 0, x = NaN).
def sincos_consts(a, vb):
    """s[86:87] 2^24 2/pi (sincos_pair puts the cosine's C4 there afterwards), s[88:89] pi/2, s[90:91] S1, s[92:93] S3, s[94:95] C1,
    s[96:97] C2; v[vb:vb+1] S2, v[vb+2:vb+3] C3"""
    c = {"s88": "0x1.921FB54442D18p0", "s90": "-0x1.555545995a603p-3", "s92": "-0x1.994eb3774cf24p-13", "s94": "-0x1.ffffffd0c621cp-2",
         "s96": "0x1.55553e1068f19p-5"}
    for r, v in c.items():
        lo, hi = _f64(v)
        n = int(r[1:])
        a(f"	s_mov_b32 s{n}, {lo:#x}\n	s_mov_b32 s{n + 1}, {hi:#x}")
    s2, c3 = _f64("0x1.1107605230bc4p-7"), _f64("-0x1.6c087e89a359dp-10")
    a(f"""
	v_mov_b32 v{vb}, {s2[0]:#x}
	v_mov_b32 v{vb + 1}, {s2[1]:#x}
	v_mov_b32 v{vb + 2}, {c3[0]:#x}
	v_mov_b32 v{vb + 3}, {c3[1]:#x}""")


def sincos_special(a, vb, xs, slow):
    """branches to `slow` unless |y| < 120 for every sample in every lane (NaN: the main path's; scratch v<vb+4>, vcc)"""
    t = f"v{vb + 4}"
    ab = [f"|{x}|" for x in xs]
    if len(ab) == 2:
        a(f"	v_max_f32_e64 {t}, {ab[0]}, {ab[1]}")
    else:
        a(f"	v_max3_f32 {t}, {ab[0]}, {ab[1]}, {ab[2]}")
        k = 3
        while len(ab) - k >= 2:
            a(f"	v_max3_f32 {t}, {t}, {ab[k]}, {ab[k + 1]}")
            k += 2
        if k < len(ab):
            a(f"	v_max_f32_e64 {t}, {ab[k]}, {t}")
    a(f"	v_cmp_ngt_f32 vcc, 0x42f00000, {t}\n	s_cbranch_vccnz {slow}")


def sincos_pair(a, vb, xin, xout, is_cos):
    """xout[j] = sinf / cosf(xin[j]), j = 0, 1, |xin| < 120 (sincos_special); window registers v<vb+4>..v<vb+21>, the constants of
    sincos_consts; xout must not be xin (it holds the quadrant meanwhile)"""
    pair = lambda r: f"v[{r}:{r + 1}]"
    S2, C3 = vb, vb + 2
    base = [vb + 4 + 9 * j for j in range(2)]
    # (pairs at even registers: sample 0 takes v+4..v+12 with its single last, sample 1 its single first)
    Pa, Pb, Pc, Pd = ([(b if j == 0 else b + 1) + 2 * k for j, b in enumerate(base)] for k in range(4))
    Rs = [base[0] + 8, base[1]]
    N = xout
    R2 = range(2)
    inv = _f64("0x1.45F306DC9C883p+23")
    c4 = _f64("0x1.99343027bf8c3p-16")
    a(f"	s_mov_b32 s86, {inv[0]:#x}\n	s_mov_b32 s87, {inv[1]:#x}")
    for j in R2:
        a(f"	v_cvt_f64_f32 {pair(Pa[j])}, {xin[j]}")
    for j in R2:
        a(f"	v_mul_f64 {pair(Pb[j])}, {pair(Pa[j])}, s[86:87]")
    for j in R2:
        a(f"	v_cvt_i32_f64 {N[j]}, {pair(Pb[j])}")
    for j in R2:
        a(f"	v_add_u32 {N[j]}, 0x800000, {N[j]}")
    for j in R2:
        a(f"	v_ashrrev_i32 {N[j]}, 24, {N[j]}")                                            # n = ((int32) r + 0x800000) >> 24
    a(f"	s_mov_b32 s86, {c4[0]:#x}\n	s_mov_b32 s87, {c4[1]:#x}")
    for j in R2:
        a(f"	v_cvt_f64_i32 {pair(Pb[j])}, {N[j]}")
    for j in R2:
        a(f"	v_fma_f64 {pair(Pa[j])}, -{pair(Pb[j])}, s[88:89], {pair(Pa[j])}")            # x = y - n pi/2
        a(f"	v_lshrrev_b32 v{Rs[j]}, 1, {N[j]}")
    for j in R2:
        a(f"	v_mul_f64 {pair(Pb[j])}, {pair(Pa[j])}, {pair(Pa[j])}")                       # x2 (before the sign: the same number)
        a(f"	v_xor_b32 v{Rs[j]}, v{Rs[j]}, {N[j]}")
    for j in R2:
        a(f"	v_lshlrev_b32 v{Rs[j]}, 31, v{Rs[j]}")
    for j in R2:
        a(f"	v_xor_b32 v{Pa[j] + 1}, v{Pa[j] + 1}, v{Rs[j]}")                              # x * sign[n & 3], sign = 1 -1 -1 1
    for j in R2:
        a(f"	v_mul_f64 {pair(Pd[j])}, {pair(Pa[j])}, {pair(Pb[j])}")                       # x3
        a(f"	v_fma_f64 {pair(Pc[j])}, {pair(Pb[j])}, s[92:93], {pair(S2)}")                # s1 = S2 + x2 S3
    for j in R2:
        a(f"	v_fma_f64 {pair(Pa[j])}, {pair(Pd[j])}, s[90:91], {pair(Pa[j])}")             # s = x + x3 S1
        a(f"	v_mul_f64 {pair(Pd[j])}, {pair(Pd[j])}, {pair(Pb[j])}")                       # x7
    for j in R2:
        a(f"	v_fma_f64 {pair(Pa[j])}, {pair(Pd[j])}, {pair(Pc[j])}, {pair(Pa[j])}")        # the sine polynomial
    for j in R2:
        a(f"	v_cvt_f32_f64 v{Rs[j]}, {pair(Pa[j])}")
        a(f"	v_mul_f64 {pair(Pa[j])}, {pair(Pb[j])}, {pair(Pb[j])}")                       # x4
    for j in R2:
        a(f"	v_fma_f64 {pair(Pc[j])}, {pair(Pb[j])}, s[86:87], {pair(C3)}")                # c2 = C3 + x2 C4
        a(f"	v_fma_f64 {pair(Pd[j])}, {pair(Pb[j])}, s[94:95], 1.0")                       # c1 = 1 + x2 C1
    for j in R2:
        a(f"	v_mul_f64 {pair(Pb[j])}, {pair(Pa[j])}, {pair(Pb[j])}")                       # x6
        a(f"	v_fma_f64 {pair(Pd[j])}, {pair(Pa[j])}, s[96:97], {pair(Pd[j])}")             # c = c1 + x4 C2
    for j in R2:
        a(f"	v_fma_f64 {pair(Pd[j])}, {pair(Pb[j])}, {pair(Pc[j])}, {pair(Pd[j])}")        # the cosine polynomial
        a(f"	v_lshlrev_b32 v{Pa[j] + 1}, 30, {N[j]}")
    for j in R2:
        a(f"	v_cvt_f32_f64 v{Pa[j]}, {pair(Pd[j])}")
        a(f"	v_and_b32 v{Pa[j] + 1}, 0x80000000, v{Pa[j] + 1}")                            # its sign: quadrants 2, 3
    for j in R2:
        a(f"	v_xor_b32 v{Pa[j]}, v{Pa[j]}, v{Pa[j] + 1}")
        a(f"	v_lshlrev_b32 v{Pc[j]}, 31, {N[j]}")                                          # the quadrant's low bit as a sign
    # sine: odd quadrant -> the cosine polynomial; cosine: even quadrant.  (Lane masks: vcc and s[86:87] - C4 is through; a mask written
    # by a VALU instruction is read two instructions later at the earliest)
    cmpop = "v_cmp_le_i32" if is_cos else "v_cmp_gt_i32"
    a(f"	{cmpop} vcc, 0, v{Pc[0]}")
    a(f"	{cmpop}_e64 s[86:87], 0, v{Pc[1]}")
    for j in R2:
        a(f"	v_and_b32 v{Pc[j]}, 0x7fffffff, {xin[j]}")
    a(f"	v_cndmask_b32 {xout[0]}, v{Rs[0]}, v{Pa[0]}, vcc")
    a(f"	v_cndmask_b32_e64 {xout[1]}, v{Rs[1]}, v{Pa[1]}, s[86:87]")
    for j in R2:
        a(f"	v_cmp_gt_u32 vcc, 0x39800000, v{Pc[j]}")                                      # |y| < 2^-12: y, 1
        a(f"	s_nop 1")
        a(f"	v_cndmask_b32 {xout[j]}, {xout[j]}, {'1.0' if is_cos else xin[j]}, vcc")


def hand(fn):
    """(constants, test for the special arguments, two samples) of the routine written by hand for the opcode fn"""
    if fn == "exp":
        return exp_consts, exp_special, exp_pair
    if fn == "ln":
        return ln_consts, ln_special, ln_pair
    return sincos_consts, sincos_special, (lambda a, vb, xin, xout, c=(fn == "cos"): sincos_pair(a, vb, xin, xout, c))


COPIES = []      # (prefix, v_base, routine names) of every embed(): the probe kernel (gen_interp.py gen_trans_probe) reaches each copy


def embed(a, path, v_base=V_BASE, prefix="fh_t_", s_map=None, wide=False, exp2=None, sincos2=False):
    """the routines as `<prefix><name>` with their vector registers in v[v_base .. v_base + 25] (a second kernel with another register
    window embeds its own copies: `s_branch` reaches 128 KB); s_map: the ten scalar registers s0..s9 go to (default s86..s95; pairs
    must stay even-aligned pairs), the return address always to s[96:97]; wide: also the four-sample routines FUNCS4, and a window of
    WIDE_V registers"""
    txt = open(path).read()
    # wide = True: all of FUNCS4 in a window of WIDE_V registers; "sincos": sin4 / cos4 only, which fit the ordinary window of MAX_V
    extra = FUNCS4 if wide is True else (["sin4", "cos4"] if wide == "sincos" else [])
    COPIES.append((prefix, v_base, FUNCS + extra + (["sin2", "cos2", "exp2", "ln2"] if exp2 else (["sin2", "cos2"] if sincos2 else []))))
    if exp2:       # the kernel's handlers hold the two-sample expf written by hand (exp_pair): its table
        exp_table(a, prefix)
    for f in FUNCS + extra:
        m = re.search(rf"^fh_t_{f}:.*?\n(.*?)^\.Lfunc_end\d+:", txt, re.S | re.M)
        assert m, f
        a(f"\t.p2align 6\n{prefix}{f}:")
        a(_rename(m.group(1), f, v_base, prefix, s_map, WIDE_V if wide is True else (EXP_TAB if exp2 else MAX_V)))
