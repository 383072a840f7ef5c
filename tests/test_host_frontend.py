"""CPU-only checks of the product's host side (no GPU, no compute calls):

* libfidget_hip.so loads and exports every symbol declared in include/fidget_hip.h;
* the product's own graph -> tape compiler (fidget_amd/csrc/host_graph.hpp, written
  independently of the oracle) yields, op for op, the same program as the oracle's
  restatement of the reference (Context rules, SsaTape order, variable numbering);
* the reference wire format (fidget_bytecode words, produced here by the oracle's
  Bytecode::new restatement, incl. a 2-register tape with Mem load/store) imports to the
  same program;
* host simplify (the algorithm the device prune sweep shares) agrees with the oracle's
  VmData::simplify on which ops survive.
"""
import ctypes
import os
import re

import numpy as np
import pytest

import fidget_amd as F
import oracle as O
from conftest import ROOT, model_path
from kat_util import build_stress_fn

MODELS = ["prospero.vm", "hi.vm", "bear.vm", "colonnade.vm", "quarter.vm", "tanglecube.vm"]


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "fidget_hip.h")).read() + open(os.path.join(ROOT, "include", "fidget_hip_debug.h")).read()
    declared = sorted(set(re.findall(r"\b(fhip_\w+)\s*\(", hdr)) - {"fhip_status"})
    assert len(declared) >= 35
    lib = ctypes.CDLL(F.LIB_PATH)
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, f"missing exports: {missing}"
    assert sorted(F.EXPORTS) == declared


def test_integration_binding_names_the_declared_functions():
    """INTEGRATION.md's `extern "C"` block (the reference-side binding) only names functions the header declares, with the
    header's argument count"""
    hdr = open(os.path.join(ROOT, "include", "fidget_hip.h")).read()
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = md[md.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    ffi = dict(re.findall(r"pub fn (fhip_\w+)\s*\(([^;]*?)\)\s*(?:->[^;]*)?;", block, re.S))
    assert len(ffi) >= 30
    flat = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    for name, args in ffi.items():
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", flat, re.S)
        assert m, f"{name} is not declared in include/fidget_hip.h"
        n_c = len([a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"])
        n_rs = len([a for a in args.split(",") if a.strip()])
        assert n_c == n_rs, f"{name}: {n_rs} arguments in INTEGRATION.md, {n_c} in the header"


def canon_product(shape):
    """(name, form, imm-or-slot) per op, evaluation order, registers abstracted to value numbers."""
    cur, nxt, out = {}, 0, []
    for name, ro, ra, rb, imm in shape.ops():
        base, form = name, ""
        for suf in ("RR", "RI", "IR"):
            if name.endswith(suf) and name not in ("CopyReg",):
                base, form = name[:-2], suf
        if name == "Output":
            out.append(("Output", cur[ra], imm))
            continue
        args = []
        if name not in ("Input", "CopyImm"):
            args.append(cur[ra])
        if form == "RR":
            args.append(cur[rb])
        cur[ro] = nxt
        out.append((base, form, tuple(args), imm if form in ("RI", "IR") or name in ("Input", "CopyImm") else 0, nxt))
        nxt += 1
    return out


def canon_oracle(shape):
    cur, nxt, out = {}, 0, []
    fm = {"RegReg": "RR", "RegImm": "RI", "ImmReg": "IR", "Reg": "", "": ""}
    for name, form, o, a, b, idx, imm in shape.asm_ops():
        if name == "Output":
            out.append(("Output", cur[a], idx))
            continue
        if name == "Load":
            cur[o] = cur[("m", idx)]
            continue
        if name == "Store":
            cur[("m", idx)] = cur[a]
            continue
        args = []
        if name not in ("Input", "CopyImm"):
            args.append(cur[a])
        if form == "RegReg":
            args.append(cur[b])
        cur[o] = nxt
        payload = idx if name == "Input" else (imm if (form in ("RegImm", "ImmReg") or name == "CopyImm") else 0)
        out.append((name, fm[form], tuple(args), payload, nxt))
        nxt += 1
    return out


@pytest.mark.parametrize("name", MODELS)
def test_compiler_matches_oracle_program(name):
    p = F.Shape.from_vm(model_path(name))
    o = O.Shape.from_vm(model_path(name))
    assert p.size() == o.ssa_len() and p.choice_count() == o.choice_count() and p.var_count() == o.var_count()
    assert [p.axis_index(a) for a in range(3)] == [o.axis_index(a) for a in range(3)]
    assert canon_product(p) == canon_oracle(o)
    assert p.slot_count() <= 256


@pytest.mark.parametrize("n", [4, 32, 256])
def test_compiler_matches_oracle_on_stress_graphs(n):
    pc, pn = build_stress_fn(F, n)
    oc, on = build_stress_fn(O, n)
    assert len(pc) == len(oc)
    assert canon_product(F.Shape(pc, pn)) == canon_oracle(O.Shape(oc, on))


@pytest.mark.parametrize("name,n_regs", [("hi.vm", 255), ("hi.vm", 3), ("colonnade.vm", 255), ("colonnade.vm", 6),
                                         ("prospero.vm", 255), ("prospero.vm", 24)])
def test_reference_bytecode_import(name, n_regs):
    # the words a Rust `fidget-hip` shim would pass: fidget_bytecode::Bytecode::new(&VmData<N>)
    o = O.Shape.from_vm(model_path(name), n_regs=n_regs)
    words, regs, mem = o.bytecode()
    if n_regs < 255:
        assert mem > 0  # the small-register tapes really contain Mem load/store ops
    p = F.Shape.from_bytecode(words, axis_slots=[o.axis_index(a) for a in range(3)])
    ref = F.Shape.from_vm(model_path(name))
    assert canon_product(p) == canon_product(ref)
    assert p.choice_count() == o.choice_count()


def test_bad_bytecode_is_rejected():
    with pytest.raises(F.FidgetHipError):
        F.Shape.from_bytecode([1, 2, 3, 4])
    with pytest.raises(F.FidgetHipError):
        F.Shape.from_bytecode([0xFFFFFFFF, 0, 0x00FFFF21 | (99 << 0), 0, 0xFFFFFFFF, 0xFFFFFFFF])


def surviving(ops_canon):
    return [(o[0], o[1], o[3]) for o in ops_canon if o[0] != "Output"]


@pytest.mark.parametrize("name", ["hi.vm", "colonnade.vm", "prospero.vm"])
def test_host_simplify_matches_oracle(name):
    """Prune with real traces (from oracle interval evaluation on a few tiles): the product's
    child must hold exactly the oracle child's non-copy ops, in order, with fewer-or-equal copies."""
    o = O.Shape.from_vm(model_path(name))
    p = F.Shape.from_vm(model_path(name))
    rng = np.random.default_rng(7)
    done = 0
    for _ in range(40):
        c = rng.uniform(-1, 1, 3)
        h = float(rng.choice([0.5, 0.25, 0.06, 0.015]))
        _, trace = o.eval_interval((c[0] - h, c[0] + h), (c[1] - h, c[1] + h), (c[2] - h, c[2] + h))
        if trace is None:
            continue
        oc, pc = o.simplify(trace), p.simplify(trace)
        a = [x for x in surviving(canon_oracle(oc)) if x[0] != "CopyReg"]
        b = [x for x in surviving(canon_product(pc)) if x[0] != "CopyReg"]
        assert a == b
        assert pc.size() <= oc.ssa_len() and pc.choice_count() == oc.choice_count()
        # a second-generation prune of the child with an all-Both trace is the identity
        assert pc.simplify([3] * pc.choice_count()).size() == pc.size()
        done += 1
    assert done >= 5


def test_simplify_errors():
    p = F.Shape.from_vm(model_path("hi.vm"))
    with pytest.raises(ValueError):
        p.simplify([3] * (p.choice_count() - 1))  # BadChoiceSlice
    with pytest.raises(ValueError):
        p.simplify([0] * p.choice_count())        # Choice::Unknown is invalid after evaluation
