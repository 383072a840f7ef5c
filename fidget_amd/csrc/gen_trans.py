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


COPIES = []      # (prefix, v_base, routine names) of every embed(): the probe kernel (gen_interp.py gen_trans_probe) reaches each copy


def embed(a, path, v_base=V_BASE, prefix="fh_t_", s_map=None, wide=False):
    """the routines as `<prefix><name>` with their vector registers in v[v_base .. v_base + 25] (a second kernel with another register
    window embeds its own copies: `s_branch` reaches 128 KB); s_map: the ten scalar registers s0..s9 go to (default s86..s95; pairs
    must stay even-aligned pairs), the return address always to s[96:97]; wide: also the four-sample routines FUNCS4, and a window of
    WIDE_V registers"""
    txt = open(path).read()
    # wide = True: all of FUNCS4 in a window of WIDE_V registers; "sincos": sin4 / cos4 only, which fit the ordinary window of MAX_V
    extra = FUNCS4 if wide is True else (["sin4", "cos4"] if wide == "sincos" else [])
    COPIES.append((prefix, v_base, FUNCS + extra))
    for f in FUNCS + extra:
        m = re.search(rf"^fh_t_{f}:.*?\n(.*?)^\.Lfunc_end\d+:", txt, re.S | re.M)
        assert m, f
        a(f"\t.p2align 6\n{prefix}{f}:")
        a(_rename(m.group(1), f, v_base, prefix, s_map, WIDE_V if wide is True else MAX_V))
