"""fh_float_eval_{16x4,32x2} (the assembly bulk interpreter, BulkEvaluator<f32>: fidget-core/src/vm/mod.rs:788-1049) on the CPU
emulator against numpy (tests/emu_util.py ref_f32, which follows dev_ops.hpp) - values bit for bit, NaN for NaN.  The samples are
chosen for the handlers' two-speed paths: sqrt takes a short sequence unless a sample of the op is below 2^-96, zero or negative;
an in-place min / max takes one test and one select per sample unless the sum of a's samples is a NaN (a NaN sample, or inf - inf)."""
import numpy as np
import pytest

import emu_util as U
from emu_util import E, F32, U32

OP = {n: i for i, n in enumerate(U.OPS)}


def run_bulk(kernel, zb, tape, inputs, n, n_out, trans=False):
    mem = E.Memory()
    a_tape = mem.map(np.concatenate([np.asarray(tape, np.uint64), np.zeros(16, np.uint64)]))      # (the scalar fetch runs two batches of 4 ops ahead)
    vars_ = np.zeros((max(inputs) + 1, n), F32)
    for s, v in inputs.items():
        vars_[s] = v
    out = np.zeros((n_out, n), F32)
    a_vars, a_out = mem.map(vars_), mem.map(out)
    ka = np.array([a_tape & 0xFFFFFFFF, a_tape >> 32, a_vars & 0xFFFFFFFF, a_vars >> 32, a_out & 0xFFFFFFFF, a_out >> 32, len(tape), n], U32)
    nr = {4: 16, 2: 32}[zb]
    hooks = U.trans_hooks(U.program(), f"fh_tb{nr}_", 64 + nr * zb) if trans else None      # (the compiled routines: stood in for by the host's libm, as for fh_columns_t)
    E.launch(U.program(), mem, kernel, ka.tobytes(), (n + 64 * zb - 1) // (64 * zb), lds_bytes=16, n_vgpr=64 + nr * zb + (26 if trans else 0), wg_y_sgpr=None, hooks=hooks)
    return out


def same(a, b):
    a, b = np.asarray(a, F32), np.asarray(b, F32)
    return ((a.view(U32) == b.view(U32)) | (np.isnan(a) & np.isnan(b))).all()


SPECIAL = np.array([0.0, -0.0, 1.0, 2.0, 4.0, 1e-30, 2.0 ** -96, 2.0 ** -97, 1e-40, 1e-45, np.inf, -np.inf, np.nan, -1.0, -1e-40, 3.0,
                    2.0 ** -126, 1.5, 0.3, 7.0e37, 3.4e38, 2.0 ** 120, 0.1, 10.0], F32)


@pytest.mark.parametrize("kernel,zb", [("fh_float_eval_16x4", 4), ("fh_float_eval_32x2", 2)])
@pytest.mark.parametrize("fill", ["ordinary", "special", "one_special_per_op"])
def test_sqrt_min_max_two_speed_paths(kernel, zb, fill):
    rng = np.random.default_rng(5)
    n = 64 * zb * 2
    if fill == "ordinary":            # every op takes the short sequences
        x = rng.uniform(0.01, 100.0, n).astype(F32); y = rng.uniform(-50.0, 50.0, n).astype(F32)
    elif fill == "special":           # ... the long ones
        x = rng.choice(SPECIAL, n).astype(F32); y = rng.choice(SPECIAL, n).astype(F32)
    else:                              # ... and both, op by op: one special sample in some of the lanes' sample groups
        x = rng.uniform(0.01, 100.0, n).astype(F32); y = rng.uniform(-50.0, 50.0, n).astype(F32)
        k = rng.choice(n, 12, replace=False)
        x[k] = rng.choice(SPECIAL, 12); y[k[:6]] = rng.choice(SPECIAL, 6)
    P = U.pack
    tape = [P(OP["INPUT"], 0, 0, 0), P(OP["INPUT"], 1, 0, 1),
            P(OP["SQRT"], 2, 0, 0), P(OP["OUTPUT"], 0, 2, 0),            # sqrt(x), not in place
            P(OP["COPY_REG"], 3, 0, 0), P(OP["MIN_RR"], 3, 3, 1), P(OP["OUTPUT"], 0, 3, 1),      # min(x, y) in place
            P(OP["COPY_REG"], 4, 1, 0), P(OP["MAX_RR"], 4, 4, 0), P(OP["OUTPUT"], 0, 4, 2),      # max(y, x) in place
            P(OP["SUB_RR"], 5, 0, 0), P(OP["MAX_RR"], 5, 5, 1), P(OP["OUTPUT"], 0, 5, 3),        # max(x - x, y): NaN where x is inf or NaN
            P(OP["SQRT"], 1, 1, 0), P(OP["MIN_RR"], 1, 1, 2), P(OP["OUTPUT"], 0, 1, 4),          # min(sqrt(y), sqrt(x)) in place
            P(OP["MIN_RR"], 6, 0, 2), P(OP["MAX_RI"], 6, 6, int(U.f2u(0.5))), P(OP["OUTPUT"], 0, 6, 5)]   # the general forms
    got = run_bulk(kernel, zb, tape, {0: x, 1: y}, n, 6)
    want = U.ref_f32(np.asarray(tape, np.uint64), {0: x, 1: y}, n)
    for slot in range(6):
        assert same(got[slot], want[slot]), f"output {slot}: {np.nonzero(got[slot].view(U32) != want[slot].view(U32))[0][:8]}"


@pytest.mark.parametrize("kernel,zb", [("fh_float_eval_16x4_t", 4), ("fh_float_eval_32x2_t", 2)])
def test_bulk_kernels_with_transcendental_handlers(kernel, zb):
    """fh_float_eval_*_t: the bulk interpreter with handlers for sin cos tan asin acos atan exp ln, atan2, the modulo and the rng opcodes
    (they call the compiled routines embedded behind the kernel) next to the plain ones, in place and not; a last, partial wave"""
    rng = np.random.default_rng(11)
    n = 64 * zb + 37
    x = rng.uniform(-3.0, 3.0, n).astype(F32); y = rng.uniform(-0.99, 0.99, n).astype(F32)
    x[:6] = [0.0, -0.0, np.inf, -np.inf, np.nan, 1e-30]
    P = U.pack
    tape = [P(OP["INPUT"], 0, 0, 0), P(OP["INPUT"], 1, 0, 1)]
    outs = 0
    for name in ("SIN", "COS", "TAN", "ATAN", "EXP"):
        tape += [P(OP[name], 2, 0, 0), P(OP["OUTPUT"], 0, 2, outs)]; outs += 1
    for name in ("ASIN", "ACOS"):
        tape += [P(OP[name], 3, 1, 0), P(OP["OUTPUT"], 0, 3, outs)]; outs += 1
    tape += [P(OP["ABS"], 4, 0, 0), P(OP["LN"], 4, 4, 0), P(OP["OUTPUT"], 0, 4, outs)]; outs += 1          # ln |x|, in place
    tape += [P(OP["ATAN2_RR"], 5, 1, 0), P(OP["OUTPUT"], 0, 5, outs)]; outs += 1
    tape += [P(OP["MOD_RI"], 6, 0, int(U.f2u(0.7))), P(OP["OUTPUT"], 0, 6, outs)]; outs += 1
    tape += [P(OP["MUL_RR"], 7, 2, 3), P(OP["ADD_RR"], 7, 7, 4), P(OP["SQRT"], 7, 7, 0), P(OP["OUTPUT"], 0, 7, outs)]; outs += 1     # plain handlers among them
    got = run_bulk(kernel, zb, tape, {0: x, 1: y}, n, outs, trans=True)
    want = U.ref_f32(np.asarray(tape, np.uint64), {0: x, 1: y}, n)
    for slot in range(outs):
        assert same(got[slot], want[slot]), (slot, got[slot][:8], np.asarray(want[slot])[:8])
