#!/usr/bin/env python3
"""A small functional emulator of the gfx950 (CDNA4) ISA subset the assembly interpreters use.

TEST / DEVELOPMENT INFRASTRUCTURE ONLY - nothing in the product imports it.  There is no GPU
in the build container and GPU time is scarce, and a hung assembly kernel can take the GPU
box down; this emulator runs the *assembled* code object (the `llvm-objdump -d` listing of
interp_gfx950.co, i.e. exactly the words the hardware fetches, with real addresses so that
`s_setpc_b64` into the handler tables works) one wavefront at a time on numpy vectors:

  * 64 lanes, EXEC / VCC / SCC / M0, 104 SGPRs, 512 VGPRs, LDS, a flat global memory made of
    registered numpy buffers; `s_set_gpr_idx_*` (M0-relative VGPR operands), VOP3 |x| / -x,
    VOP3P op_sel / neg;
  * IEEE f32 arithmetic with denormals, exact fma (double-rounding corrected); the hardware's
    approximate v_rcp / v_sqrt are emulated as correctly rounded and v_div_fixup returns the
    correctly rounded quotient (the div / sqrt sequences end in the same value either way);
  * a checker for the software-visible hazards of gfx940+ (VALU-written SGPR read by VALU /
    v_readlane / VMEM, VCC -> v_div_fmas, EXEC -> v_readlane, trans -> VALU) which the hardware
    does not interlock and an emulator would otherwise hide;
  * instruction counters per class (SALU, VALU, LDS, VMEM, SMEM) for static cost estimates.

It is validated by running the kernels that are known good on the hardware (fh_tiles,
fh_columns, fh_prune1 of round 1) against the numpy restatements in tests/.
"""
import re
import struct
import subprocess

import numpy as np

U32 = np.uint32
F32 = np.float32
LANES = 64
FULL = (1 << 64) - 1


class EmuError(Exception):
    pass


def _f2u(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


# ---- program -------------------------------------------------------------------------------
class Inst:
    __slots__ = ("addr", "size", "mn", "ops", "mods", "text", "word0")

    def __init__(self, addr, size, mn, ops, mods, text, word0=0):
        self.addr, self.size, self.mn, self.ops, self.mods, self.text, self.word0 = addr, size, mn, ops, mods, text, word0


_MOD_RE = re.compile(r"\b(op_sel|op_sel_hi|neg_lo|neg_hi):\[([0-9,]+)\]|\boffset:(-?\d+)|\b(sc0|sc1|nt|glc|slc|clamp)\b")


def _split_ops(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "[(":
            depth += 1
        elif ch in "])":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


class Program:
    def __init__(self, co_path, objdump="/opt/rocm/lib/llvm/bin/llvm-objdump"):
        txt = subprocess.check_output([objdump, "-d", co_path], text=True)
        self.insts = {}
        self.symbols = {}
        # the .text section's bytes at their addresses: kernels read tables that lie between them (PC-relative global loads)
        self.text_base, self.text = 0, np.zeros(0, dtype=np.uint8)
        try:
            import tempfile
            for line in subprocess.check_output([objdump, "-h", co_path], text=True).split("\n"):
                f = line.split()
                if len(f) >= 4 and f[1] == ".text":
                    self.text_base = int(f[3], 16)
            with tempfile.NamedTemporaryFile(suffix=".bin") as tf:
                subprocess.check_call([objdump.replace("llvm-objdump", "llvm-objcopy"), "-O", "binary", "-j", ".text", co_path, tf.name])
                self.text = np.fromfile(tf.name, dtype=np.uint8)
        except Exception:       # noqa: BLE001  (a kernel that reads its code segment then fails with "outside every buffer")
            pass
        for line in txt.split("\n"):
            m = re.match(r"^([0-9a-f]{16}) <([^>]+)>:", line)
            if m:
                self.symbols[m.group(2)] = int(m.group(1), 16)
                continue
            m = re.match(r"^\t(\S+)\s*(.*?)\s*// ([0-9A-F]{12}): ((?:[0-9A-F]{8} ?)+)", line)
            if not m:
                continue
            mn, rest, addr, words = m.group(1), m.group(2), int(m.group(3), 16), m.group(4).split()
            mods = {}
            for mm in _MOD_RE.finditer(rest):
                if mm.group(1):
                    mods[mm.group(1)] = [int(x) for x in mm.group(2).split(",")]
                elif mm.group(3) is not None:
                    mods["offset"] = int(mm.group(3))
                else:
                    mods[mm.group(4)] = True
            rest = _MOD_RE.sub("", rest).strip()
            if mn == "s_waitcnt" or mn == "s_nop" or mn == "s_endpgm":
                ops = [rest]
            else:
                ops = _split_ops(rest)
            for suf in ("_e32", "_e64"):
                if mn.endswith(suf):
                    mn = mn[: -len(suf)]
            self.insts[addr] = Inst(addr, 4 * len(words), mn, ops, mods, line.split("//")[0].strip(), int(words[0], 16))


# ---- memory --------------------------------------------------------------------------------
class Memory:
    """Flat global memory: registered numpy uint8 buffers at fake device addresses."""

    def __init__(self):
        self.bufs = []   # (base, end, array)
        self.next = 0x7F0000000000

    def alloc(self, nbytes, name=""):
        a = np.zeros(nbytes, dtype=np.uint8)
        return self.map(a, name)

    def map(self, arr, name=""):
        arr = arr.view(np.uint8).reshape(-1)
        base = self.next
        self.next += (len(arr) + 0xFFF + 0x1000) & ~0xFFF
        self.bufs.append((base, base + len(arr), arr, name))
        return base

    def find(self, addr, n):
        for base, end, arr, name in self.bufs:
            if base <= addr and addr + n <= end:
                return arr, addr - base
        raise EmuError(f"global access of {n} bytes at {addr:#x} is outside every buffer")

    def array(self, base):
        for b, e, arr, name in self.bufs:
            if b == base:
                return arr
        raise KeyError(base)

    def read_u32(self, addr, n=1):
        arr, o = self.find(addr, 4 * n)
        return arr[o:o + 4 * n].view(U32).copy() if o % 4 == 0 else np.frombuffer(arr[o:o + 4 * n].tobytes(), U32).copy()

    def write_u32(self, addr, vals):
        vals = np.ascontiguousarray(np.asarray(vals, dtype=U32).reshape(-1))
        arr, o = self.find(addr, 4 * len(vals))
        arr[o:o + 4 * len(vals)] = vals.view(np.uint8)


# ---- the wave ------------------------------------------------------------------------------
TRANS = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_f64", "v_sqrt_f64",
         "v_rsq_f64"}


class Wave:
    def __init__(self, prog, mem, lds_bytes=160 * 1024, check_hazards=True):
        self.p, self.mem = prog, mem
        self.s = np.zeros(128, dtype=U32)      # s0..s103 (+ scratch up to 127)
        self.vcc = 0
        self.exec = FULL
        self.scc = 0
        self.m0 = 0
        self.v = np.zeros((512, LANES), dtype=U32)
        self.lds = np.zeros(lds_bytes, dtype=np.uint8)
        self.pc = 0
        self.idx_on = False
        self.counts = {}
        self.n_inst = 0
        self.check = check_hazards
        self.hz = {}          # hazard bookkeeping: name -> (issue index, ...)
        self.issue = 0        # wait-state clock
        self.trace = None
        self.done = False
        self.lanes = np.arange(LANES, dtype=U32)
        self.mn_counts = {}
        self.n_vgpr = 512     # launch() sets the kernel's allocation
        self.hooks = {}       # address -> python function(wave): native stand-ins for embedded compiled routines

    # -- exec / masks -------------------------------------------------------------------------
    @staticmethod
    def _bits(m):
        return np.array([(m >> i) & 1 for i in range(64)], dtype=bool)

    @staticmethod
    def _pack(b):
        m = 0
        for i in np.nonzero(b)[0]:
            m |= 1 << int(i)
        return m

    # -- scalar operand access ---------------------------------------------------------------------
    def _sreg_index(self, name):
        m = re.match(r"^s(\d+)$", name)
        if m:
            return int(m.group(1)), 1
        m = re.match(r"^s\[(\d+):(\d+)\]$", name)
        if m:
            return int(m.group(1)), int(m.group(2)) - int(m.group(1)) + 1
        return None

    def rs32(self, name):
        """32-bit scalar source"""
        r = self._sreg_index(name)
        if r:
            self._hz_salu_read(r[0], 1)
            return int(self.s[r[0]])
        if name == "vcc_lo":
            return self.vcc & 0xFFFFFFFF
        if name == "vcc_hi":
            return self.vcc >> 32
        if name == "exec_lo":
            return self.exec & 0xFFFFFFFF
        if name == "exec_hi":
            return self.exec >> 32
        if name == "m0":
            return self.m0
        if name == "scc":
            return self.scc
        return self._lit32(name)

    @staticmethod
    def _lit32(name):
        if re.match(r"^-?\d+$", name):
            return int(name) & 0xFFFFFFFF
        if re.match(r"^0x[0-9a-fA-F]+$", name):
            return int(name, 16) & 0xFFFFFFFF
        if re.match(r"^-?\d+\.\d+(e[-+]?\d+)?$", name):
            return _f2u(float(name))
        raise EmuError(f"unknown scalar operand {name!r}")

    def rs64(self, name):
        r = self._sreg_index(name)
        if r:
            if r[1] != 2:
                raise EmuError(f"64-bit operand expects an SGPR pair: {name}")
            self._hz_salu_read(r[0], 2)
            return int(self.s[r[0]]) | (int(self.s[r[0] + 1]) << 32)
        if name == "vcc":
            return self.vcc
        if name == "exec":
            return self.exec
        if re.match(r"^-?\d+$", name):
            return int(name) & FULL          # inline integers are sign extended
        if re.match(r"^0x[0-9a-fA-F]+$", name):
            return int(name, 16)             # 32-bit literal, zero extended
        raise EmuError(f"unknown 64-bit scalar operand {name!r}")

    def ws32(self, name, val):
        val &= 0xFFFFFFFF
        r = self._sreg_index(name)
        if r:
            self.s[r[0]] = val
            return
        if name == "vcc_lo":
            self.vcc = (self.vcc & ~0xFFFFFFFF) | val
        elif name == "vcc_hi":
            self.vcc = (self.vcc & 0xFFFFFFFF) | (val << 32)
        elif name == "exec_lo":
            self.exec = (self.exec & ~0xFFFFFFFF) | val
        elif name == "exec_hi":
            self.exec = (self.exec & 0xFFFFFFFF) | (val << 32)
        elif name == "m0":
            self.m0 = val
        else:
            raise EmuError(f"cannot write scalar {name}")

    def ws64(self, name, val):
        val &= FULL
        r = self._sreg_index(name)
        if r:
            self.s[r[0]] = val & 0xFFFFFFFF
            self.s[r[0] + 1] = val >> 32
        elif name == "vcc":
            self.vcc = val
        elif name == "exec":
            self.exec = val
        else:
            raise EmuError(f"cannot write 64-bit scalar {name}")

    # -- hazards ---------------------------------------------------------------------------------
    def _hz_valu_wrote_sgpr(self, name):
        """a VALU instruction wrote this SGPR / VCC"""
        if not self.check:
            return
        r = self._sreg_index(name)
        regs = range(r[0], r[0] + r[1]) if r else (["vcc"] if name.startswith("vcc") else [])
        for x in regs:
            self.hz[("vs", x)] = self.issue

    def _hz_salu_read(self, idx, n):
        pass   # SALU reads of VALU-written SGPRs are interlocked

    def _hz_need(self, key, states, what):
        t = self.hz.get(key)
        if t is not None and self.issue - t - 1 < states:
            raise EmuError(f"hazard at {self.pc:#x} ({self.cur.text}): {what} needs {states} wait states, has {self.issue - t - 1}")

    def _hz_valu_reads_sgpr(self, name, states=2, what="VALU read of a VALU-written SGPR"):
        if not self.check:
            return
        r = self._sreg_index(name)
        regs = range(r[0], r[0] + r[1]) if r else (["vcc"] if isinstance(name, str) and name.startswith("vcc") else [])
        for x in regs:
            self._hz_need(("vs", x), states, what)

    # -- vector operand access -------------------------------------------------------------------
    def _vreg(self, name):
        m = re.match(r"^v(\d+)$", name)
        if m:
            return int(m.group(1)), 1
        m = re.match(r"^v\[(\d+):(\d+)\]$", name)
        if m:
            return int(m.group(1)), int(m.group(2)) - int(m.group(1)) + 1
        return None

    def _rel(self, which):
        """M0-relative offset of VALU operand `which` (0..2 sources, 3 destination)"""
        if not self.idx_on:
            return 0
        mode = (self.m0 >> 12) & 0xF
        return (self.m0 & 0xFF) if (mode >> which) & 1 else 0

    def vsrc(self, name, which, width=1):
        """VALU source: returns (array [width][64] of uint32, neg, abs)"""
        neg = ab = False
        if name.startswith("-") and len(name) > 1 and name[1] in "vs|":
            neg, name = True, name[1:]
        if name.startswith("|") and name.endswith("|"):
            ab, name = True, name[1:-1]
        r = self._vreg(name)
        if r:
            base = r[0] + self._rel(which)
            if base + width > self.n_vgpr:
                raise EmuError(f"VGPR v{base} (+{width}) is outside the kernel's allocation of {self.n_vgpr} ({self.cur.text})")
            if r[1] != width and not (width == 1 and r[1] == 1):
                if r[1] < width:
                    raise EmuError(f"operand {name} narrower than {width} dwords")
            if self.check:
                for k in range(width):
                    tw = self.hz.get(("tr", base + k))
                    if tw is not None and self.issue - tw - 1 < 1 and self.cur.mn not in TRANS:
                        raise EmuError(f"hazard at {self.pc:#x} ({self.cur.text}): non-trans VALU reads v{base + k} written by a trans op 0 wait states ago")
            vals = self.v[base:base + width].copy()
        else:
            rr = self._sreg_index(name)
            if rr or name in ("vcc", "vcc_lo", "vcc_hi", "exec", "exec_lo", "exec_hi", "m0"):
                self._hz_valu_reads_sgpr(name)
                if width == 1:
                    x = self.rs32(name) if not (rr and rr[1] == 2) else self.rs64(name) & 0xFFFFFFFF
                    vals = np.full((1, LANES), x, dtype=U32)
                else:
                    x = self.rs64(name)
                    vals = np.stack([np.full(LANES, x & 0xFFFFFFFF, dtype=U32), np.full(LANES, x >> 32, dtype=U32)])
            else:
                x = self._lit32(name)
                if width == 1:
                    vals = np.full((1, LANES), x, dtype=U32)
                else:   # 64-bit literal / inline: integers sign-extend, floats are f64 constants (unused)
                    if re.match(r"^-?\d+$", name):
                        y = int(name) & FULL
                        vals = np.stack([np.full(LANES, y & 0xFFFFFFFF, dtype=U32), np.full(LANES, y >> 32, dtype=U32)])
                    else:
                        vals = np.stack([np.full(LANES, x, dtype=U32), np.zeros(LANES, dtype=U32)])
        return vals, neg, ab

    def fsrc(self, name, which):
        vals, neg, ab = self.vsrc(name, which)
        f = vals[0].view(F32).copy()
        if ab:
            f = np.abs(f)
        if neg:
            f = -f
        return f

    def usrc(self, name, which):
        vals, neg, ab = self.vsrc(name, which)
        if neg or ab:
            raise EmuError("float modifier on an integer operand")
        return vals[0]

    def vdst(self, name, vals, width=1):
        """masked write of [width][64] (or [64]) uint32 to a VGPR destination"""
        r = self._vreg(name)
        if not r:
            raise EmuError(f"bad VGPR destination {name}")
        base = r[0] + self._rel(3)
        if base + width > self.n_vgpr:
            raise EmuError(f"VGPR write v{base} (+{width}) is outside the kernel's allocation of {self.n_vgpr} ({self.cur.text})")
        vals = np.asarray(vals)
        if vals.dtype == F32:
            vals = vals.view(U32)
        vals = vals.astype(U32, copy=False).reshape(width, LANES)
        m = self._bits(self.exec)
        for k in range(width):
            self.v[base + k][m] = vals[k][m]
            if self.check:
                if self.cur.mn in TRANS:
                    self.hz[("tr", base + k)] = self.issue
                else:
                    self.hz.pop(("tr", base + k), None)
                self.hz[("vw", base + k)] = self.issue

    def sdst_mask(self, name, bits):
        """VALU compare result: lanes outside EXEC write 0"""
        m = self._pack(bits & self._bits(self.exec))
        if name == "vcc":
            self.vcc = m
        else:
            self.ws64(name, m)
        self._hz_valu_wrote_sgpr(name)

    def mask_src(self, name):
        self._hz_valu_reads_sgpr(name, 2, "VALU read of a VALU-written lane mask")
        return self._bits(self.rs64(name))

    # -- float helpers ------------------------------------------------------------------------------
    @staticmethod
    def fma32(a, b, c):
        """exact fused multiply-add in f32 (a*b exact in f64; the sum's double rounding is corrected)"""
        a64, b64, c64 = a.astype(np.float64), b.astype(np.float64), c.astype(np.float64)
        with np.errstate(all="ignore"):
            p = a64 * b64
            s = p + c64
            # TwoSum error term
            bb = s - p
            err = (p - (s - bb)) + (c64 - bb)
            r = s.astype(F32)
            # s exactly halfway between two f32 values and a non-zero error: redo the rounding
            bits = s.view(np.uint64)
            half = (bits & np.uint64(0x1FFFFFFF)) == np.uint64(0x10000000)
            fix = half & (err != 0) & np.isfinite(s)
            if fix.any():
                nudged = np.where(err > 0, np.nextafter(s, np.inf), np.nextafter(s, -np.inf))
                r = np.where(fix, nudged.astype(F32), r)
        return r

    # -- run ------------------------------------------------------------------------------------------
    def count(self, klass):
        self.counts[klass] = self.counts.get(klass, 0) + 1

    def run(self, entry, max_inst=50_000_000):
        self.pc = entry
        self.done = False
        while not self.done:
            h = self.hooks.get(self.pc)
            if h is not None:
                self.pc = h(self)
                continue
            inst = self.p.insts.get(self.pc)
            if inst is None:
                raise EmuError(f"pc {self.pc:#x} is not an instruction")
            self.cur = inst
            self.n_inst += 1
            if self.n_inst > max_inst:
                raise EmuError("instruction limit exceeded (runaway loop?)")
            self.mn_counts[inst.mn] = self.mn_counts.get(inst.mn, 0) + 1
            if self.trace is not None:
                self.trace(self, inst)
            nxt = self.pc + inst.size
            self.issue += 1
            r = self.step(inst)
            self.pc = nxt if r is None else r

    def step(self, i):
        mn, o = i.mn, i.ops
        f = getattr(self, "i_" + mn, None)
        if f is not None:
            return f(i, *o)
        if mn.startswith("s_cmp_"):
            self.count("salu")
            return self.i_s_cmp(mn, *o)
        if mn.startswith("v_cmp_"):
            self.count("valu")
            return self.i_v_cmp(mn, *o)
        raise EmuError(f"unimplemented instruction: {i.text}")

    # ================= SALU =====================================================================
    def i_s_nop(self, i, n):
        self.count("salu")
        self.issue += int(n, 0)

    def i_s_setprio(self, i, n):
        self.count("salu")      # (issue priority within the SIMD: nothing to model for one wave)

    def i_s_waitcnt(self, i, *a):
        self.count("salu")

    def i_s_endpgm(self, i, *a):
        self.done = True

    def i_s_mov_b32(self, i, d, s):
        self.count("salu")
        self.ws32(d, self.rs32(s))

    def i_s_mov_b64(self, i, d, s):
        self.count("salu")
        self.ws64(d, self.rs64(s))

    def _s2(self, i, d, a, b, fn, bits=32, scc="nz"):
        self.count("salu")
        if bits == 32:
            x, y = self.rs32(a), self.rs32(b)
            r = fn(x, y)
            res = r & 0xFFFFFFFF
            self.ws32(d, res)
        else:
            x, y = self.rs64(a), self.rs64(b)
            r = fn(x, y)
            res = r & FULL
            self.ws64(d, res)
        if scc == "nz":
            self.scc = 1 if res != 0 else 0
        return r

    def i_s_add_u32(self, i, d, a, b):
        r = self._s2(i, d, a, b, lambda x, y: x + y, scc=None)
        self.scc = 1 if r > 0xFFFFFFFF else 0

    def i_s_addc_u32(self, i, d, a, b):
        c = self.scc
        r = self._s2(i, d, a, b, lambda x, y: x + y + c, scc=None)
        self.scc = 1 if r > 0xFFFFFFFF else 0

    def i_s_sub_u32(self, i, d, a, b):
        r = self._s2(i, d, a, b, lambda x, y: x - y, scc=None)
        self.scc = 1 if r < 0 else 0

    def i_s_subb_u32(self, i, d, a, b):
        c = self.scc
        r = self._s2(i, d, a, b, lambda x, y: x - y - c, scc=None)
        self.scc = 1 if r < 0 else 0

    def i_s_add_i32(self, i, d, a, b):
        self.count("salu")
        x, y = self.rs32(a), self.rs32(b)
        sx, sy = x - (1 << 32) * (x >> 31), y - (1 << 32) * (y >> 31)
        r = sx + sy
        self.ws32(d, r)
        self.scc = 1 if r > 0x7FFFFFFF or r < -0x80000000 else 0

    def i_s_sub_i32(self, i, d, a, b):
        self.count("salu")
        x, y = self.rs32(a), self.rs32(b)
        sx, sy = x - (1 << 32) * (x >> 31), y - (1 << 32) * (y >> 31)
        r = sx - sy
        self.ws32(d, r)
        self.scc = 1 if r > 0x7FFFFFFF or r < -0x80000000 else 0

    def i_s_mul_i32(self, i, d, a, b):
        self._s2(i, d, a, b, lambda x, y: x * y, scc=None)

    def i_s_mul_hi_u32(self, i, d, a, b):
        self._s2(i, d, a, b, lambda x, y: (x * y) >> 32, scc=None)

    def i_s_min_u32(self, i, d, a, b):
        x, y = self.rs32(a), self.rs32(b)
        self.count("salu")
        self.ws32(d, min(x, y))
        self.scc = 1 if x <= y else 0   # SCC = 1 if S0 is the minimum

    def i_s_max_u32(self, i, d, a, b):
        x, y = self.rs32(a), self.rs32(b)
        self.count("salu")
        self.ws32(d, max(x, y))
        self.scc = 1 if x >= y else 0

    def i_s_and_b32(self, i, d, a, b):
        self._s2(i, d, a, b, lambda x, y: x & y)

    def i_s_or_b32(self, i, d, a, b):
        self._s2(i, d, a, b, lambda x, y: x | y)

    def i_s_xor_b32(self, i, d, a, b):
        self._s2(i, d, a, b, lambda x, y: x ^ y)

    def i_s_andn2_b32(self, i, d, a, b):
        self._s2(i, d, a, b, lambda x, y: x & ~y)

    def i_s_and_b64(self, i, d, a, b):
        self._s2(i, d, a, b, lambda x, y: x & y, 64)

    def i_s_or_b64(self, i, d, a, b):
        self._s2(i, d, a, b, lambda x, y: x | y, 64)

    def i_s_xor_b64(self, i, d, a, b):
        self._s2(i, d, a, b, lambda x, y: x ^ y, 64)

    def i_s_andn2_b64(self, i, d, a, b):
        self._s2(i, d, a, b, lambda x, y: x & ~y, 64)

    def i_s_orn2_b64(self, i, d, a, b):
        self._s2(i, d, a, b, lambda x, y: x | (~y & FULL), 64)

    def i_s_not_b64(self, i, d, a):
        self.count("salu")
        r = ~self.rs64(a) & FULL
        self.ws64(d, r)
        self.scc = 1 if r else 0

    def i_s_not_b32(self, i, d, a):
        self.count("salu")
        r = ~self.rs32(a) & 0xFFFFFFFF
        self.ws32(d, r)
        self.scc = 1 if r else 0

    def i_s_lshl_b32(self, i, d, a, b):
        self._s2(i, d, a, b, lambda x, y: x << (y & 31))

    def i_s_lshr_b32(self, i, d, a, b):
        self._s2(i, d, a, b, lambda x, y: x >> (y & 31))

    def i_s_ashr_i32(self, i, d, a, b):
        self._s2(i, d, a, b, lambda x, y: (x - (1 << 32) * (x >> 31)) >> (y & 31))

    def i_s_lshl_b64(self, i, d, a, b):
        self.count("salu")
        r = (self.rs64(a) << (self.rs32(b) & 63)) & FULL
        self.ws64(d, r)
        self.scc = 1 if r else 0

    def i_s_lshr_b64(self, i, d, a, b):
        self.count("salu")
        r = self.rs64(a) >> (self.rs32(b) & 63)
        self.ws64(d, r)
        self.scc = 1 if r else 0

    def i_s_bfe_u32(self, i, d, a, b):
        self.count("salu")
        x, y = self.rs32(a), self.rs32(b)
        off, w = y & 31, (y >> 16) & 0x7F
        r = (x >> off) & ((1 << w) - 1) if w else 0
        self.ws32(d, r)
        self.scc = 1 if r else 0

    def i_s_bfe_i32(self, i, d, a, b):
        self.count("salu")
        x, y = self.rs32(a), self.rs32(b)
        off, w = y & 31, (y >> 16) & 0x7F
        r = (x >> off) & ((1 << w) - 1) if w else 0
        if w and w < 32 and (r >> (w - 1)) & 1:
            r |= (0xFFFFFFFF << w) & 0xFFFFFFFF
        self.ws32(d, r)
        self.scc = 1 if r else 0

    def i_s_bfm_b64(self, i, d, a, b):
        self.count("salu")
        w, off = self.rs32(a) & 63, self.rs32(b) & 63
        self.ws64(d, (((1 << w) - 1) << off) & FULL)

    def i_s_bfm_b32(self, i, d, a, b):
        self.count("salu")
        w, off = self.rs32(a) & 31, self.rs32(b) & 31
        self.ws32(d, (((1 << w) - 1) << off))

    def i_s_cselect_b32(self, i, d, a, b):
        self.count("salu")
        x, y = self.rs32(a), self.rs32(b)
        self.ws32(d, x if self.scc else y)

    def i_s_cselect_b64(self, i, d, a, b):
        self.count("salu")
        x, y = self.rs64(a), self.rs64(b)
        self.ws64(d, x if self.scc else y)

    def i_s_bcnt1_i32_b64(self, i, d, a):
        self.count("salu")
        r = bin(self.rs64(a)).count("1")
        self.ws32(d, r)
        self.scc = 1 if r else 0

    def i_s_bcnt1_i32_b32(self, i, d, a):
        self.count("salu")
        r = bin(self.rs32(a)).count("1")
        self.ws32(d, r)
        self.scc = 1 if r else 0

    def i_s_ff1_i32_b64(self, i, d, a):
        self.count("salu")
        x = self.rs64(a)
        self.ws32(d, (x & -x).bit_length() - 1 if x else 0xFFFFFFFF)

    def i_s_ff1_i32_b32(self, i, d, a):
        self.count("salu")
        x = self.rs32(a)
        self.ws32(d, (x & -x).bit_length() - 1 if x else 0xFFFFFFFF)

    def i_s_flbit_i32_b64(self, i, d, a):
        self.count("salu")
        x = self.rs64(a)
        self.ws32(d, 64 - x.bit_length() if x else 0xFFFFFFFF)

    def i_s_flbit_i32_b32(self, i, d, a):
        self.count("salu")
        x = self.rs32(a)
        self.ws32(d, 32 - x.bit_length() if x else 0xFFFFFFFF)

    def i_s_bitset0_b64(self, i, d, a):
        self.count("salu")
        self.ws64(d, self.rs64(d) & ~(1 << (self.rs32(a) & 63)))

    def i_s_bitset1_b64(self, i, d, a):
        self.count("salu")
        self.ws64(d, self.rs64(d) | (1 << (self.rs32(a) & 63)))

    def i_s_bitset0_b32(self, i, d, a):
        self.count("salu")
        self.ws32(d, self.rs32(d) & ~(1 << (self.rs32(a) & 31)))

    def i_s_bitset1_b32(self, i, d, a):
        self.count("salu")
        self.ws32(d, self.rs32(d) | (1 << (self.rs32(a) & 31)))

    def i_s_bitcmp1_b32(self, i, a, b):
        self.count("salu")
        self.scc = (self.rs32(a) >> (self.rs32(b) & 31)) & 1

    def i_s_bitcmp0_b32(self, i, a, b):
        self.count("salu")
        self.scc = 1 - ((self.rs32(a) >> (self.rs32(b) & 31)) & 1)

    def i_s_bitcmp1_b64(self, i, a, b):
        self.count("salu")
        self.scc = (self.rs64(a) >> (self.rs32(b) & 63)) & 1

    def i_s_bitcmp0_b64(self, i, a, b):
        self.count("salu")
        self.scc = 1 - ((self.rs64(a) >> (self.rs32(b) & 63)) & 1)

    def i_s_cmp(self, mn, a, b):
        m = re.match(r"s_cmp_(eq|lg|gt|ge|lt|le)_(u32|i32|u64)", mn)
        if not m:
            raise EmuError(f"unimplemented {mn}")
        op, ty = m.groups()
        if ty == "u64":
            x, y = self.rs64(a), self.rs64(b)
        else:
            x, y = self.rs32(a), self.rs32(b)
            if ty == "i32":
                x, y = x - (1 << 32) * (x >> 31), y - (1 << 32) * (y >> 31)
        self.scc = int({"eq": x == y, "lg": x != y, "gt": x > y, "ge": x >= y, "lt": x < y, "le": x <= y}[op])

    def _branch(self, i, off):
        o = i.word0 & 0xFFFF          # SOPP simm16 (the listing shows a symbol when the target has one)
        if o >= 0x8000:
            o -= 0x10000
        return i.addr + 4 + 4 * o

    def i_s_branch(self, i, off):
        self.count("salu")
        return self._branch(i, off)

    def i_s_cbranch_scc0(self, i, off):
        self.count("salu")
        return self._branch(i, off) if not self.scc else None

    def i_s_cbranch_scc1(self, i, off):
        self.count("salu")
        return self._branch(i, off) if self.scc else None

    def i_s_cbranch_execz(self, i, off):
        self.count("salu")
        return self._branch(i, off) if self.exec == 0 else None

    def i_s_cbranch_execnz(self, i, off):
        self.count("salu")
        return self._branch(i, off) if self.exec != 0 else None

    def i_s_cbranch_vccz(self, i, off):
        self.count("salu")
        return self._branch(i, off) if self.vcc == 0 else None

    def i_s_cbranch_vccnz(self, i, off):
        self.count("salu")
        return self._branch(i, off) if self.vcc != 0 else None

    def i_s_setpc_b64(self, i, a):
        self.count("salu")
        return self.rs64(a)

    def i_s_swappc_b64(self, i, d, a):
        self.count("salu")
        t = self.rs64(a)
        self.ws64(d, i.addr + 4)
        return t

    def i_s_getpc_b64(self, i, d):
        self.count("salu")
        self.ws64(d, i.addr + 4)

    def i_s_and_saveexec_b64(self, i, d, a):
        self.count("salu")
        x = self.rs64(a)
        self.ws64(d, self.exec)
        self.exec &= x
        self.scc = 1 if self.exec else 0

    def i_s_or_saveexec_b64(self, i, d, a):
        self.count("salu")
        x = self.rs64(a)
        self.ws64(d, self.exec)
        self.exec |= x
        self.scc = 1 if self.exec else 0

    def i_s_memtime(self, i, d):
        self.count("smem")
        self.ws64(d, self.n_inst * 4)

    def i_s_memrealtime(self, i, d):
        self.count("smem")
        self.ws64(d, self.n_inst // 6)

    @staticmethod
    def _idx_mode(mode):
        m = re.match(r"gpr_idx\((.*)\)", mode)
        if m:
            return sum({"SRC0": 1, "SRC1": 2, "SRC2": 4, "DST": 8}[t.strip()] for t in m.group(1).split(",") if t.strip())
        return int(mode, 0) & 0xF

    def i_s_set_gpr_idx_on(self, i, a, mode):
        self.count("salu")
        self.m0 = (self.m0 & ~0xF0FF) | (self.rs32(a) & 0xFF) | (self._idx_mode(mode) << 12)
        self.idx_on = True

    def i_s_set_gpr_idx_idx(self, i, a):
        self.count("salu")
        self.m0 = (self.m0 & ~0xFF) | (self.rs32(a) & 0xFF)

    def i_s_set_gpr_idx_off(self, i):
        self.count("salu")
        self.idx_on = False

    def i_s_set_gpr_idx_mode(self, i, mode):
        self.count("salu")
        self.m0 = (self.m0 & ~0xF000) | (self._idx_mode(mode) << 12)

    # -- SMEM ---------------------------------------------------------------------------------------
    def _sload(self, i, d, base, off, n):
        self.count("smem")
        addr = self.rs64(base) + (self.rs32(off) if not re.match(r"^(0x)?[0-9a-fA-F]+$", off) else int(off, 0))
        addr &= ~3
        vals = self.mem.read_u32(addr, n)
        r = self._sreg_index(d)
        if r is None and d == "vcc":
            self.vcc = int(vals[0]) | (int(vals[1]) << 32)
            return
        if r[1] != n:
            raise EmuError(f"s_load destination width mismatch: {i.text}")
        self.s[r[0]:r[0] + n] = vals

    def i_s_load_dword(self, i, d, b, o):
        self._sload(i, d, b, o, 1)

    def i_s_load_dwordx2(self, i, d, b, o):
        self._sload(i, d, b, o, 2)

    def i_s_load_dwordx4(self, i, d, b, o):
        self._sload(i, d, b, o, 4)

    def i_s_load_dwordx8(self, i, d, b, o):
        self._sload(i, d, b, o, 8)

    def i_s_load_dwordx16(self, i, d, b, o):
        self._sload(i, d, b, o, 16)

    # ================= VALU =====================================================================
    def _v1(self, i, d, a, fn, kind="u"):
        self.count("valu")
        x = self.fsrc(a, 0) if kind == "f" else self.usrc(a, 0)
        with np.errstate(all="ignore"):
            self.vdst(d, fn(x))

    def _v2(self, i, d, a, b, fn, kind="u"):
        self.count("valu")
        if kind == "f":
            x, y = self.fsrc(a, 0), self.fsrc(b, 1)
        else:
            x, y = self.usrc(a, 0), self.usrc(b, 1)
        with np.errstate(all="ignore"):
            self.vdst(d, fn(x, y))

    def i_v_mov_b32(self, i, d, a):
        self._v1(i, d, a, lambda x: x)

    def i_v_mov_b64(self, i, d, a):
        self.count("valu")
        vals, _, _ = self.vsrc(a, 0, 2)
        self.vdst(d, vals, 2)

    def i_v_not_b32(self, i, d, a):
        self._v1(i, d, a, lambda x: ~x)

    def i_v_add_u32(self, i, d, a, b):
        self._v2(i, d, a, b, lambda x, y: x + y)

    def i_v_sub_u32(self, i, d, a, b):
        self._v2(i, d, a, b, lambda x, y: x - y)

    def i_v_subrev_u32(self, i, d, a, b):
        self._v2(i, d, a, b, lambda x, y: y - x)

    def i_v_mul_lo_u32(self, i, d, a, b):
        self._v2(i, d, a, b, lambda x, y: (x.astype(np.uint64) * y.astype(np.uint64)).astype(U32))

    def i_v_mul_hi_u32(self, i, d, a, b):
        self._v2(i, d, a, b, lambda x, y: ((x.astype(np.uint64) * y.astype(np.uint64)) >> np.uint64(32)).astype(U32))

    def i_v_mad_u32_u24(self, i, d, a, b, c):
        self.count("valu")
        x, y, z = self.usrc(a, 0), self.usrc(b, 1), self.usrc(c, 2)
        self.vdst(d, (((x & U32(0xFFFFFF)).astype(np.uint64) * (y & U32(0xFFFFFF)).astype(np.uint64) + z.astype(np.uint64)) & np.uint64(0xFFFFFFFF)).astype(U32))

    def i_v_mul_u32_u24(self, i, d, a, b):
        self._v2(i, d, a, b, lambda x, y: ((x & U32(0xFFFFFF)).astype(np.uint64) * (y & U32(0xFFFFFF)).astype(np.uint64)).astype(U32))

    def i_v_and_b32(self, i, d, a, b):
        self._v2(i, d, a, b, lambda x, y: x & y)

    def i_v_or_b32(self, i, d, a, b):
        self._v2(i, d, a, b, lambda x, y: x | y)

    def i_v_xor_b32(self, i, d, a, b):
        self._v2(i, d, a, b, lambda x, y: x ^ y)

    def i_v_lshlrev_b32(self, i, d, a, b):
        self._v2(i, d, a, b, lambda x, y: y << (x & U32(31)))

    def i_v_lshrrev_b32(self, i, d, a, b):
        self._v2(i, d, a, b, lambda x, y: y >> (x & U32(31)))

    def i_v_ashrrev_i32(self, i, d, a, b):
        self._v2(i, d, a, b, lambda x, y: (y.view(np.int32) >> (x & U32(31)).astype(np.int32)).view(U32))

    def i_v_min_u32(self, i, d, a, b):
        self._v2(i, d, a, b, np.minimum)

    def i_v_max_u32(self, i, d, a, b):
        self._v2(i, d, a, b, np.maximum)

    def i_v_min_i32(self, i, d, a, b):
        self._v2(i, d, a, b, lambda x, y: np.minimum(x.view(np.int32), y.view(np.int32)).view(U32))

    def i_v_max_i32(self, i, d, a, b):
        self._v2(i, d, a, b, lambda x, y: np.maximum(x.view(np.int32), y.view(np.int32)).view(U32))

    def _v3u(self, i, d, a, b, c, fn):
        self.count("valu")
        x, y, z = self.usrc(a, 0), self.usrc(b, 1), self.usrc(c, 2)
        self.vdst(d, fn(x, y, z))

    def i_v_min3_u32(self, i, d, a, b, c):
        self._v3u(i, d, a, b, c, lambda x, y, z: np.minimum(np.minimum(x, y), z))

    def i_v_max3_u32(self, i, d, a, b, c):
        self._v3u(i, d, a, b, c, lambda x, y, z: np.maximum(np.maximum(x, y), z))

    def i_v_lshl_or_b32(self, i, d, a, b, c):
        self._v3u(i, d, a, b, c, lambda x, y, z: (x << (y & U32(31))) | z)

    def i_v_lshl_add_u32(self, i, d, a, b, c):
        self._v3u(i, d, a, b, c, lambda x, y, z: (x << (y & U32(31))) + z)

    def i_v_add_lshl_u32(self, i, d, a, b, c):
        self._v3u(i, d, a, b, c, lambda x, y, z: (x + y) << (z & U32(31)))

    def i_v_add3_u32(self, i, d, a, b, c):
        self._v3u(i, d, a, b, c, lambda x, y, z: x + y + z)

    def i_v_and_or_b32(self, i, d, a, b, c):
        self._v3u(i, d, a, b, c, lambda x, y, z: (x & y) | z)

    def i_v_or3_b32(self, i, d, a, b, c):
        self._v3u(i, d, a, b, c, lambda x, y, z: x | y | z)

    def i_v_bfe_u32(self, i, d, a, b, c):
        def f(x, y, z):
            w = z & U32(31)
            return np.where(w == 0, U32(0), (x >> (y & U32(31))) & ((U32(1) << w) - U32(1))).astype(U32)
        self._v3u(i, d, a, b, c, f)

    def i_v_bfi_b32(self, i, d, a, b, c):
        self._v3u(i, d, a, b, c, lambda x, y, z: (x & y) | (~x & z))

    def i_v_perm_b32(self, i, d, a, b, c):
        def f(x, y, z):
            src = (x.astype(np.uint64) << np.uint64(32)) | y.astype(np.uint64)
            out = np.zeros(LANES, dtype=U32)
            for k in range(4):
                sel = (z >> U32(8 * k)) & U32(0xFF)
                byte = np.zeros(LANES, dtype=U32)
                for ln in range(LANES):
                    s = int(sel[ln])
                    if s <= 7:
                        byte[ln] = (int(src[ln]) >> (8 * s)) & 0xFF
                    elif s == 12:
                        byte[ln] = 0
                    elif s >= 13:
                        byte[ln] = 0xFF
                    else:
                        raise EmuError("v_perm_b32 selector 8..11 not emulated")
                out |= byte << U32(8 * k)
            return out
        self._v3u(i, d, a, b, c, f)

    def i_v_ffbl_b32(self, i, d, a):
        def f(x):
            out = np.full(LANES, 0xFFFFFFFF, dtype=U32)
            for ln in range(LANES):
                v = int(x[ln])
                if v:
                    out[ln] = (v & -v).bit_length() - 1
            return out
        self._v1(i, d, a, f)

    def i_v_ffbh_u32(self, i, d, a):
        def f(x):
            out = np.full(LANES, 0xFFFFFFFF, dtype=U32)
            for ln in range(LANES):
                v = int(x[ln])
                if v:
                    out[ln] = 32 - v.bit_length()
            return out
        self._v1(i, d, a, f)

    def i_v_bcnt_u32_b32(self, i, d, a, b):
        self._v2(i, d, a, b, lambda x, y: np.array([bin(int(v)).count("1") for v in x], dtype=U32) + y)

    def i_v_lshlrev_b64(self, i, d, a, b):
        self.count("valu")
        sh = self.usrc(a, 0)
        vals, _, _ = self.vsrc(b, 1, 2)
        x = vals[0].astype(np.uint64) | (vals[1].astype(np.uint64) << np.uint64(32))
        r = x << (sh & U32(63)).astype(np.uint64)
        self.vdst(d, np.stack([(r & np.uint64(0xFFFFFFFF)).astype(U32), (r >> np.uint64(32)).astype(U32)]), 2)

    def i_v_lshrrev_b64(self, i, d, a, b):
        self.count("valu")
        sh = self.usrc(a, 0)
        vals, _, _ = self.vsrc(b, 1, 2)
        x = vals[0].astype(np.uint64) | (vals[1].astype(np.uint64) << np.uint64(32))
        r = x >> (sh & U32(63)).astype(np.uint64)
        self.vdst(d, np.stack([(r & np.uint64(0xFFFFFFFF)).astype(U32), (r >> np.uint64(32)).astype(U32)]), 2)

    def i_v_lshl_add_u64(self, i, d, a, b, c):
        self.count("valu")
        va, _, _ = self.vsrc(a, 0, 2)
        sh = self.usrc(b, 1)
        vc, _, _ = self.vsrc(c, 2, 2)
        x = va[0].astype(np.uint64) | (va[1].astype(np.uint64) << np.uint64(32))
        z = vc[0].astype(np.uint64) | (vc[1].astype(np.uint64) << np.uint64(32))
        with np.errstate(over="ignore"):
            r = (x << (sh & U32(63)).astype(np.uint64)) + z
        self.vdst(d, np.stack([(r & np.uint64(0xFFFFFFFF)).astype(U32), (r >> np.uint64(32)).astype(U32)]), 2)

    def i_v_add_co_u32(self, i, d, sd, a, b):
        self.count("valu")
        x, y = self.usrc(a, 0).astype(np.uint64), self.usrc(b, 1).astype(np.uint64)
        r = x + y
        self.vdst(d, (r & np.uint64(0xFFFFFFFF)).astype(U32))
        self.sdst_mask(sd, r > np.uint64(0xFFFFFFFF))

    def i_v_sub_co_u32(self, i, d, sd, a, b):
        self.count("valu")
        x, y = self.usrc(a, 0), self.usrc(b, 1)
        self.vdst(d, x - y)
        self.sdst_mask(sd, y > x)

    def i_v_addc_co_u32(self, i, d, sd, a, b, c):
        self.count("valu")
        x, y = self.usrc(a, 0).astype(np.uint64), self.usrc(b, 1).astype(np.uint64)
        cin = self._bits(self.rs64(c)).astype(np.uint64)   # (carry chains are interlocked: no wait states needed)
        r = x + y + cin
        self.vdst(d, (r & np.uint64(0xFFFFFFFF)).astype(U32))
        self.sdst_mask(sd, r > np.uint64(0xFFFFFFFF))

    def i_v_cndmask_b32(self, i, d, a, b, m):
        self.count("valu")
        vals_a, na, aa = self.vsrc(a, 0)
        vals_b, nb, ab = self.vsrc(b, 1)
        if na or aa or nb or ab:
            fa, fb = self.fsrc(a, 0), self.fsrc(b, 1)
            vals_a, vals_b = fa.view(U32)[None], fb.view(U32)[None]
        mask = self.mask_src(m)
        self.vdst(d, np.where(mask, vals_b[0], vals_a[0]))

    def i_v_readfirstlane_b32(self, i, d, a):
        self.count("valu")
        r = self._vreg(a)
        e = self.exec
        lane = (e & -e).bit_length() - 1 if e else 0
        # measured on MI355X (tools/probe_isa.py): SRC0-relative indexing applies to v_readfirstlane / v_readlane
        self.ws32(d, int(self.v[r[0] + self._rel(0)][lane]))
        self._hz_valu_wrote_sgpr(d)

    def i_v_readlane_b32(self, i, d, a, l):
        self.count("valu")
        r = self._vreg(a)
        r = (r[0] + self._rel(0), r[1])     # (measured: the scalar destination is NOT displaced by DST-relative mode)
        if self.check:
            self._hz_valu_reads_sgpr(l, 4, "v_readlane lane select written by VALU")
            t = self.hz.get(("vw", r[0]))
            if t is not None and self.issue - t - 1 < 1:
                raise EmuError(f"hazard at {self.pc:#x} ({self.cur.text}): v_readlane of a VGPR written 0 wait states ago")
            self._hz_need("vexec", 4, "v_readlane after a VALU write of EXEC")
        lane = self.rs32(l) & 63
        self.ws32(d, int(self.v[r[0]][lane]))
        self._hz_valu_wrote_sgpr(d)

    def i_v_writelane_b32(self, i, d, a, l):
        self.count("valu")
        if self.idx_on:
            raise EmuError("v_writelane with the GPR index mode on")
        r = self._vreg(d)
        if self.check:
            self._hz_valu_reads_sgpr(l, 4, "v_writelane lane select written by VALU")
            self._hz_valu_reads_sgpr(a, 2)
        lane = self.rs32(l) & 63
        self.v[r[0]][lane] = self.rs32(a)
        if self.check:
            self.hz[("vw", r[0])] = self.issue

    def i_v_mbcnt_lo_u32_b32(self, i, d, a, b):
        self.count("valu")
        m = self.usrc(a, 0)
        base = self.usrc(b, 1)
        out = np.zeros(LANES, dtype=U32)
        for ln in range(LANES):
            lo = int(m[ln]) & ((1 << min(ln, 32)) - 1)
            out[ln] = bin(lo).count("1")
        self.vdst(d, out + base)

    def i_v_mbcnt_hi_u32_b32(self, i, d, a, b):
        self.count("valu")
        m = self.usrc(a, 0)
        base = self.usrc(b, 1)
        out = np.zeros(LANES, dtype=U32)
        for ln in range(LANES):
            hi = int(m[ln]) & ((1 << max(ln - 32, 0)) - 1)
            out[ln] = bin(hi).count("1")
        self.vdst(d, out + base)

    # -- f32 --------------------------------------------------------------------------------------
    def i_v_add_f32(self, i, d, a, b):
        self._v2(i, d, a, b, lambda x, y: x + y, "f")

    def i_v_sub_f32(self, i, d, a, b):
        self._v2(i, d, a, b, lambda x, y: x - y, "f")

    def i_v_subrev_f32(self, i, d, a, b):
        self._v2(i, d, a, b, lambda x, y: y - x, "f")

    def i_v_mul_f32(self, i, d, a, b):
        self._v2(i, d, a, b, lambda x, y: x * y, "f")

    @staticmethod
    def _fmin(x, y):
        # IEEE mode: minNum semantics (a NaN operand loses), sNaN quieting ignored
        r = np.where(np.isnan(x), y, np.where(np.isnan(y), x, np.minimum(x, y)))
        # -0 < +0
        z = (x == 0) & (y == 0)
        r = np.where(z, np.where(np.signbit(x) | np.signbit(y), F32(-0.0), F32(0.0)), r)
        return r.astype(F32)

    @staticmethod
    def _fmax(x, y):
        r = np.where(np.isnan(x), y, np.where(np.isnan(y), x, np.maximum(x, y)))
        z = (x == 0) & (y == 0)
        r = np.where(z, np.where(np.signbit(x) & np.signbit(y), F32(-0.0), F32(0.0)), r)
        return r.astype(F32)

    def i_v_min3_f32(self, i, d, a, b, c):
        self.count("valu")
        x, y, z = self.fsrc(a, 0), self.fsrc(b, 1), self.fsrc(c, 2)
        self.vdst(d, self._fmin(self._fmin(x, y), z))

    def i_v_max3_f32(self, i, d, a, b, c):
        self.count("valu")
        x, y, z = self.fsrc(a, 0), self.fsrc(b, 1), self.fsrc(c, 2)
        self.vdst(d, self._fmax(self._fmax(x, y), z))

    def i_v_minimum3_f32(self, i, d, a, b, c):
        # gfx950: IEEE-754-2019 minimum - a NaN operand wins, -0 < +0 (measured over every pair of special values and 4 M random
        # pairs: tools/probe_minimum3.cpp, profiles/r06a/probe_minimum3.txt)
        self.count("valu")
        x, y, z = self.fsrc(a, 0), self.fsrc(b, 1), self.fsrc(c, 2)
        nan = np.isnan(x) | np.isnan(y) | np.isnan(z)
        with np.errstate(all="ignore"):
            r = self._fmin(self._fmin(x, y), z)
        self.vdst(d, np.where(nan, F32(np.nan), r).astype(F32))

    def i_v_maximum3_f32(self, i, d, a, b, c):
        self.count("valu")
        x, y, z = self.fsrc(a, 0), self.fsrc(b, 1), self.fsrc(c, 2)
        nan = np.isnan(x) | np.isnan(y) | np.isnan(z)
        with np.errstate(all="ignore"):
            r = self._fmax(self._fmax(x, y), z)
        self.vdst(d, np.where(nan, F32(np.nan), r).astype(F32))

    def i_v_min_f32(self, i, d, a, b):
        self._v2(i, d, a, b, self._fmin, "f")

    def i_v_max_f32(self, i, d, a, b):
        self._v2(i, d, a, b, self._fmax, "f")

    def i_v_fma_f32(self, i, d, a, b, c):
        self.count("valu")
        x, y, z = self.fsrc(a, 0), self.fsrc(b, 1), self.fsrc(c, 2)
        self.vdst(d, self.fma32(x, y, z))

    def i_v_fmac_f32(self, i, d, a, b):
        self.count("valu")
        x, y = self.fsrc(a, 0), self.fsrc(b, 1)
        r = self._vreg(d)
        z = self.v[r[0] + self._rel(3)].view(F32).copy()
        self.vdst(d, self.fma32(x, y, z))

    def i_v_mad_f32(self, i, d, a, b, c):
        raise EmuError("v_mad_f32 does not exist on gfx950")

    def i_v_floor_f32(self, i, d, a):
        self._v1(i, d, a, np.floor, "f")

    def i_v_ceil_f32(self, i, d, a):
        self._v1(i, d, a, np.ceil, "f")

    def i_v_trunc_f32(self, i, d, a):
        self._v1(i, d, a, np.trunc, "f")

    def i_v_rndne_f32(self, i, d, a):
        self._v1(i, d, a, np.rint, "f")

    def i_v_fract_f32(self, i, d, a):
        self._v1(i, d, a, lambda x: np.minimum(x - np.floor(x), F32(np.nextafter(F32(1), F32(0)))), "f")

    def i_v_rcp_f32(self, i, d, a):
        self._v1(i, d, a, lambda x: (np.float64(1.0) / x.astype(np.float64)).astype(F32), "f")

    def i_v_sqrt_f32(self, i, d, a):
        self._v1(i, d, a, lambda x: np.sqrt(x.astype(np.float64)).astype(F32), "f")

    def i_v_rsq_f32(self, i, d, a):
        self._v1(i, d, a, lambda x: (1.0 / np.sqrt(x.astype(np.float64))).astype(F32), "f")

    def i_v_cvt_f32_u32(self, i, d, a):
        self._v1(i, d, a, lambda x: x.astype(np.float64).astype(F32))

    def i_v_cvt_f32_i32(self, i, d, a):
        self._v1(i, d, a, lambda x: x.view(np.int32).astype(np.float64).astype(F32))

    def i_v_cvt_u32_f32(self, i, d, a):
        def f(x):
            y = np.where(np.isnan(x), 0.0, np.clip(np.trunc(x.astype(np.float64)), 0, 4294967295.0))
            return y.astype(np.uint64).astype(U32)
        self._v1(i, d, a, f, "f")

    def i_v_cvt_i32_f32(self, i, d, a):
        def f(x):
            y = np.where(np.isnan(x), 0.0, np.clip(np.trunc(x.astype(np.float64)), -2147483648.0, 2147483647.0))
            return y.astype(np.int64).astype(np.int32).view(U32)
        self._v1(i, d, a, f, "f")

    def i_v_ldexp_f32(self, i, d, a, b):
        self.count("valu")
        x = self.fsrc(a, 0)
        e = self.usrc(b, 1).view(np.int32)
        with np.errstate(all="ignore"):
            self.vdst(d, np.ldexp(x.astype(np.float64), e).astype(F32))

    def i_v_frexp_mant_f32(self, i, d, a):
        self._v1(i, d, a, lambda x: np.where(np.isfinite(x), np.frexp(x.astype(np.float64))[0], x).astype(F32), "f")

    def i_v_frexp_exp_i32_f32(self, i, d, a):
        self._v1(i, d, a, lambda x: np.where(np.isfinite(x) & (x != 0), np.frexp(x.astype(np.float64))[1], 0).astype(np.int32).view(U32), "f")

    def i_v_div_scale_f32(self, i, d, sd, a, b, c):
        """Scaling for the division sequence.  The emulated v_div_fixup recomputes the quotient from its
        operands, so the scaled values only have to keep the intermediate steps finite: return S0, VCC = 0."""
        self.count("valu")
        x = self.fsrc(a, 0)
        self.fsrc(b, 1), self.fsrc(c, 2)
        self.vdst(d, x)
        self.sdst_mask(sd, np.zeros(LANES, dtype=bool))

    def i_v_div_fmas_f32(self, i, d, a, b, c):
        self.count("valu")
        if self.check:
            self._hz_need(("vs", "vcc"), 4, "v_div_fmas after a VALU write of VCC")
        x, y, z = self.fsrc(a, 0), self.fsrc(b, 1), self.fsrc(c, 2)
        self.vdst(d, self.fma32(x, y, z))

    def i_v_div_fixup_f32(self, i, d, q, den, num):
        self.count("valu")
        self.fsrc(q, 0)
        dd, nn = self.fsrc(den, 1).astype(np.float64), self.fsrc(num, 2).astype(np.float64)
        with np.errstate(all="ignore"):
            self.vdst(d, (nn / dd).astype(F32))

    def i_v_cmp_class_f32(self, i, sd, a, b):
        self.count("valu")
        x = self.fsrc(a, 0)
        cls = self.usrc(b, 1)
        bits = x.view(U32)
        exp = (bits >> U32(23)) & U32(0xFF)
        man = bits & U32(0x7FFFFF)
        neg = (bits >> U32(31)) == 1
        snan = (exp == 255) & (man != 0) & ((man >> U32(22)) == 0)
        qnan = (exp == 255) & ((man >> U32(22)) == 1)
        inf = (exp == 255) & (man == 0)
        den = (exp == 0) & (man != 0)
        zero = (exp == 0) & (man == 0)
        norm = (exp != 0) & (exp != 255)
        tests = [snan, qnan, inf & neg, norm & neg, den & neg, zero & neg, zero & ~neg, den & ~neg, norm & ~neg, inf & ~neg]
        r = np.zeros(LANES, dtype=bool)
        for k, t in enumerate(tests):
            r |= t & (((cls >> U32(k)) & U32(1)) == 1)
        self.sdst_mask(sd, r)

    def i_v_cmp(self, mn, sd, a, b):
        m = re.match(r"v_cmp_(\w+)_(f32|u32|i32)$", mn)
        if not m:
            raise EmuError(f"unimplemented {mn}")
        op, ty = m.groups()
        if ty == "f32":
            x, y = self.fsrc(a, 0), self.fsrc(b, 1)
            un = np.isnan(x) | np.isnan(y)
            with np.errstate(all="ignore"):
                r = {"lt": x < y, "eq": x == y, "le": x <= y, "gt": x > y, "lg": (x < y) | (x > y), "ge": x >= y, "o": ~un, "u": un,
                     "nge": ~(x >= y), "nlg": ~((x < y) | (x > y)), "ngt": ~(x > y), "nle": ~(x <= y), "neq": ~(x == y), "nlt": ~(x < y),
                     "f": np.zeros(LANES, bool), "tru": np.ones(LANES, bool)}[op]
        else:
            x, y = self.usrc(a, 0), self.usrc(b, 1)
            if ty == "i32":
                x, y = x.view(np.int32), y.view(np.int32)
            r = {"lt": x < y, "eq": x == y, "le": x <= y, "gt": x > y, "ne": x != y, "ge": x >= y}[op]
        self.sdst_mask(sd, r)

    # -- binary64 (the hand-written expf of gen_trans.py) ------------------------------------------------
    _F64_INLINE = {"0": 0.0, "0.5": 0.5, "-0.5": -0.5, "1.0": 1.0, "-1.0": -1.0, "2.0": 2.0, "-2.0": -2.0, "4.0": 4.0, "-4.0": -4.0}

    def dsrc(self, name, which):
        """a 64-bit float source: VGPR / SGPR pair with -, | | modifiers, or a float inline constant (the f64 value)"""
        if name in self._F64_INLINE:
            return np.full(LANES, self._F64_INLINE[name], dtype=np.float64)
        vals, neg, ab = self.vsrc(name, which, 2)
        bits = vals[0].astype(np.uint64) | (vals[1].astype(np.uint64) << np.uint64(32))
        if ab:
            bits = bits & np.uint64(0x7FFFFFFFFFFFFFFF)
        if neg:
            bits = bits ^ np.uint64(0x8000000000000000)
        return bits.view(np.float64)

    def ddst(self, name, x):
        bits = np.ascontiguousarray(x, dtype=np.float64).view(np.uint64)
        self.vdst(name, np.stack([(bits & np.uint64(0xFFFFFFFF)).astype(U32), (bits >> np.uint64(32)).astype(U32)]), 2)

    @staticmethod
    def fma64(a, b, c):
        """exact fused multiply-add in f64, lane by lane in rational arithmetic (int / int division rounds to nearest even)"""
        from fractions import Fraction
        with np.errstate(all="ignore"):
            r = a * b + c                   # infinities, NaNs, signed zeros: the unfused result's
        for k in range(len(a)):
            x, y, z = float(a[k]), float(b[k]), float(c[k])
            if not (np.isfinite(x) and np.isfinite(y) and np.isfinite(z)):
                continue
            e = Fraction(x) * Fraction(y) + Fraction(z)
            if e != 0:
                try:
                    r[k] = float(e)
                except OverflowError:
                    r[k] = np.inf if e > 0 else -np.inf
        return r

    def i_v_fma_f64(self, i, d, a, b, c):
        self.count("valu")
        self.ddst(d, self.fma64(self.dsrc(a, 0), self.dsrc(b, 1), self.dsrc(c, 2)))

    def i_v_add_f64(self, i, d, a, b):
        self.count("valu")
        with np.errstate(all="ignore"):
            self.ddst(d, self.dsrc(a, 0) + self.dsrc(b, 1))

    def i_v_mul_f64(self, i, d, a, b):
        self.count("valu")
        with np.errstate(all="ignore"):
            self.ddst(d, self.dsrc(a, 0) * self.dsrc(b, 1))

    def i_v_cvt_f64_f32(self, i, d, a):
        self.count("valu")
        self.ddst(d, self.fsrc(a, 0).astype(np.float64))

    def i_v_cvt_i32_f64(self, i, d, a):
        self.count("valu")
        x = self.dsrc(a, 0)
        with np.errstate(all="ignore"):
            y = np.where(np.isnan(x), 0.0, np.clip(np.trunc(x), -2147483648.0, 2147483647.0))
        self.vdst(d, y.astype(np.int64).astype(np.int32).view(U32))

    def i_v_cvt_f64_i32(self, i, d, a):
        self.count("valu")
        self.ddst(d, self.usrc(a, 0).view(np.int32).astype(np.float64))

    def i_v_cvt_f32_f64(self, i, d, a):
        self.count("valu")
        with np.errstate(all="ignore"):
            self.vdst(d, self.dsrc(a, 0).astype(F32))

    # -- packed f32 (VOP3P) ----------------------------------------------------------------------------
    def _pk(self, i, d, a, b, fn, c=None):
        self.count("valu")
        srcs = [self.vsrc(a, 0, 2)[0], self.vsrc(b, 1, 2)[0]] + ([self.vsrc(c, 2, 2)[0]] if c else [])
        n = len(srcs)
        osl = i.mods.get("op_sel", [0] * n)
        osh = i.mods.get("op_sel_hi", [1] * n)
        nlo = i.mods.get("neg_lo", [0] * n)
        nhi = i.mods.get("neg_hi", [0] * n)
        sgn = lambda x, neg: (x.view(U32) ^ U32(0x80000000)).view(F32) if neg else x.view(F32)   # (sign flip: exact, NaN-safe)
        lo_in = [sgn(srcs[k][osl[k]], nlo[k]) for k in range(n)]
        hi_in = [sgn(srcs[k][osh[k]], nhi[k]) for k in range(n)]
        # (negation by multiplication keeps NaN payloads irrelevant; -0 handled: -1 * 0 = -0)
        with np.errstate(all="ignore"):
            lo, hi = fn(*lo_in), fn(*hi_in)
        self.vdst(d, np.stack([lo.astype(F32).view(U32), hi.astype(F32).view(U32)]), 2)

    def i_v_pk_add_f32(self, i, d, a, b):
        self._pk(i, d, a, b, lambda x, y: x + y)

    def i_v_pk_mul_f32(self, i, d, a, b):
        self._pk(i, d, a, b, lambda x, y: x * y)

    def i_v_pk_fma_f32(self, i, d, a, b, c):
        self._pk(i, d, a, b, self.fma32, c)

    def i_v_pk_mov_b32(self, i, d, a, b):
        self.count("valu")
        sa, sb = self.vsrc(a, 0, 2)[0], self.vsrc(b, 1, 2)[0]
        osl = i.mods.get("op_sel", [0, 0])
        self.vdst(d, np.stack([sa[osl[0]], sb[osl[1]]]), 2)

    # ================= LDS ======================================================================
    def _ds_addr(self, i, a):
        r = self._vreg(a)
        return self.v[r[0]].astype(np.int64) + i.mods.get("offset", 0)

    def _ds_read(self, i, d, a, nbytes, width):
        self.count("lds")
        addr = self._ds_addr(i, a)
        m = self._bits(self.exec)
        out = np.zeros((width, LANES), dtype=U32)
        for ln in np.nonzero(m)[0]:
            ad = int(addr[ln])
            if ad < 0 or ad + nbytes > len(self.lds):
                self.lds_oob_reads = getattr(self, "lds_oob_reads", 0) + 1   # the hardware returns 0 for out-of-range reads
                continue
            raw = self.lds[ad:ad + nbytes].tobytes()
            if nbytes < 4:
                out[0, ln] = int.from_bytes(raw, "little")
            else:
                out[:, ln] = np.frombuffer(raw, dtype=U32)
        self.vdst(d, out, width)

    def _ds_write(self, i, a, dsrc, nbytes, width):
        self.count("lds")
        addr = self._ds_addr(i, a)
        r = self._vreg(dsrc)
        m = self._bits(self.exec)
        for ln in np.nonzero(m)[0]:
            ad = int(addr[ln])
            if ad < 0 or ad + nbytes > len(self.lds):
                raise EmuError(f"LDS write out of range: lane {ln} addr {ad} ({i.text})")
            if nbytes < 4:
                self.lds[ad:ad + nbytes] = np.frombuffer(int(self.v[r[0]][ln]).to_bytes(4, "little")[:nbytes], dtype=np.uint8)
            else:
                self.lds[ad:ad + nbytes] = self.v[r[0]:r[0] + width, ln].copy().view(np.uint8)

    def i_ds_bpermute_b32(self, i, d, a, b):
        """dst[lane] = src[(addr[lane] >> 2) & 63] for the active lanes (no LDS memory involved; a disabled source lane reads as 0)"""
        self.count("lds")
        addr = self.v[self._vreg(a)[0]].astype(np.int64) + i.mods.get("offset", 0)
        src = self.v[self._vreg(b)[0]]
        m = self._bits(self.exec)
        lane = (addr >> 2) & 63
        vals = np.where(m[lane], src[lane], 0).astype(U32)
        self.lgkm_pending = getattr(self, "lgkm_pending", 0)
        self.vdst(d, vals)

    def i_ds_read_b32(self, i, d, a):
        self._ds_read(i, d, a, 4, 1)

    def i_ds_read_b64(self, i, d, a):
        self._ds_read(i, d, a, 8, 2)

    def i_ds_read_b128(self, i, d, a):
        self._ds_read(i, d, a, 16, 4)

    def i_ds_read_u8(self, i, d, a):
        self._ds_read(i, d, a, 1, 1)

    def i_ds_read_u16(self, i, d, a):
        self._ds_read(i, d, a, 2, 1)

    def i_ds_write_b32(self, i, a, d):
        self._ds_write(i, a, d, 4, 1)

    def i_ds_write_b64(self, i, a, d):
        self._ds_write(i, a, d, 8, 2)

    def i_ds_write_b128(self, i, a, d):
        self._ds_write(i, a, d, 16, 4)

    def i_ds_write_b8(self, i, a, d):
        self._ds_write(i, a, d, 1, 1)

    def i_ds_write_b16(self, i, a, d):
        self._ds_write(i, a, d, 2, 1)

    # ================= global memory ================================================================
    def _gaddr(self, i, vaddr, saddr):
        off = i.mods.get("offset", 0)
        r = self._vreg(vaddr)
        if saddr == "off":
            if r[1] != 2:
                raise EmuError(f"global access with 'off' needs a 64-bit address: {i.text}")
            a = self.v[r[0]].astype(np.uint64) | (self.v[r[0] + 1].astype(np.uint64) << np.uint64(32))
            return [int(x) + off for x in a]
        if self.check:
            self._hz_valu_reads_sgpr(saddr, 5, "VMEM read of a VALU-written SGPR")
        base = self.rs64(saddr)
        return [base + int(x) + off for x in self.v[r[0]]]

    def _gload(self, i, d, vaddr, saddr, width):
        self.count("vmem")
        addrs = self._gaddr(i, vaddr, saddr)
        m = self._bits(self.exec)
        out = np.zeros((width, LANES), dtype=U32)
        for ln in np.nonzero(m)[0]:
            out[:, ln] = self.mem.read_u32(addrs[ln], width)
        self.vdst(d, out, width)

    def i_global_load_dword(self, i, d, va, sa):
        self._gload(i, d, va, sa, 1)

    def i_global_load_dwordx2(self, i, d, va, sa):
        self._gload(i, d, va, sa, 2)

    def i_global_load_dwordx3(self, i, d, va, sa):
        self._gload(i, d, va, sa, 3)

    def i_global_load_dwordx4(self, i, d, va, sa):
        self._gload(i, d, va, sa, 4)

    def i_global_load_ubyte(self, i, d, va, sa):
        self.count("vmem")
        addrs = self._gaddr(i, va, sa)
        m = self._bits(self.exec)
        out = np.zeros((1, LANES), dtype=U32)
        for ln in np.nonzero(m)[0]:
            arr, o = self.mem.find(addrs[ln], 1)
            out[0, ln] = arr[o]
        self.vdst(d, out, 1)

    def _gstore(self, i, va, dsrc, sa, width):
        self.count("vmem")
        addrs = self._gaddr(i, va, sa)
        r = self._vreg(dsrc)
        m = self._bits(self.exec)
        for ln in np.nonzero(m)[0]:
            self.mem.write_u32(addrs[ln], self.v[r[0]:r[0] + width, ln])

    def i_global_store_dword(self, i, va, d, sa):
        self._gstore(i, va, d, sa, 1)

    def i_global_store_dwordx2(self, i, va, d, sa):
        self._gstore(i, va, d, sa, 2)

    def i_global_store_dwordx4(self, i, va, d, sa):
        self._gstore(i, va, d, sa, 4)

    def i_global_store_byte(self, i, va, d, sa):
        self.count("vmem")
        addrs = self._gaddr(i, va, sa)
        r = self._vreg(d)
        m = self._bits(self.exec)
        for ln in np.nonzero(m)[0]:
            arr, o = self.mem.find(addrs[ln], 1)
            arr[o] = int(self.v[r[0]][ln]) & 0xFF

    def _gatomic(self, i, ops, fn, width):
        self.count("vmem")
        ret = bool(i.mods.get("sc0") or i.mods.get("glc"))
        if ret:
            d, va, dsrc, sa = ops
        else:
            va, dsrc, sa = ops
        addrs = self._gaddr(i, va, sa)
        r = self._vreg(dsrc)
        m = self._bits(self.exec)
        out = np.zeros((width, LANES), dtype=U32)
        for ln in np.nonzero(m)[0]:
            old = self.mem.read_u32(addrs[ln], width)
            out[:, ln] = old
            if width == 1:
                o, v = int(old[0]), int(self.v[r[0]][ln])
                self.mem.write_u32(addrs[ln], [fn(o, v) & 0xFFFFFFFF])
            else:
                o = int(old[0]) | (int(old[1]) << 32)
                v = int(self.v[r[0]][ln]) | (int(self.v[r[0] + 1][ln]) << 32)
                n = fn(o, v) & FULL
                self.mem.write_u32(addrs[ln], [n & 0xFFFFFFFF, n >> 32])
        if ret:
            self.vdst(d, out, width)

    def i_global_atomic_add(self, i, *ops):
        self._gatomic(i, ops, lambda o, v: o + v, 1)

    def i_global_atomic_sub(self, i, *ops):
        self._gatomic(i, ops, lambda o, v: o - v, 1)

    def i_global_atomic_umin(self, i, *ops):
        self._gatomic(i, ops, min, 1)

    def i_global_atomic_add_x2(self, i, *ops):
        self._gatomic(i, ops, lambda o, v: o + v, 2)

    def i_global_atomic_umax(self, i, *ops):
        self._gatomic(i, ops, max, 1)

    def i_global_atomic_umax_x2(self, i, *ops):
        self._gatomic(i, ops, max, 2)

    def i_global_atomic_or(self, i, *ops):
        self._gatomic(i, ops, lambda o, v: o | v, 1)


def launch(prog, mem, kernel, kernarg_bytes, n_workgroups=1, grid_y=1, lds_bytes=160 * 1024, check_hazards=True, wg_id_sgpr=2, wg_y_sgpr=3,
           max_inst=50_000_000, trace=None, n_vgpr=512, hooks=None):
    """Run `kernel` for every single-wave workgroup of the grid, one after the other.
    Conventions of the interpreters: s[0:1] = kernarg segment, s2 = workgroup id x, s3 = workgroup id y (when enabled), v0 = lane id.
    Returns the list of waves (for their counters)."""
    ka = mem.map(np.frombuffer(bytes(kernarg_bytes) + b"\0" * 64, dtype=np.uint8).copy(), "kernarg")
    if len(prog.text) and not any(name == ".text" for _, _, _, name in mem.bufs):
        mem.bufs.append((prog.text_base, prog.text_base + len(prog.text), prog.text, ".text"))
    waves = []
    for y in range(grid_y):
        for x in range(n_workgroups):
            w = Wave(prog, mem, lds_bytes, check_hazards)
            w.s[0], w.s[1] = ka & 0xFFFFFFFF, ka >> 32
            w.s[wg_id_sgpr] = x
            if wg_y_sgpr is not None:
                w.s[wg_y_sgpr] = y
            w.v[0] = np.arange(LANES, dtype=U32)
            w.trace = trace
            w.hooks = hooks or {}
            w.n_vgpr = n_vgpr
            w.run(prog.symbols[kernel], max_inst)
            waves.append(w)
    return waves
