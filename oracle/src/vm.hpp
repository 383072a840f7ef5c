// ORACLE — TEST INFRASTRUCTURE ONLY (see types.hpp header).
//
// Restates the VM backend (the reference's own in-suite oracle, `VmShape`):
//   fidget-core/src/vm/data.rs  (VmData 64-117, simplify 123-318, VmWorkspace 355-408)
//   fidget-core/src/vm/mod.rs   (VmIntervalEval 325-538, VmPointEval 543-760,
//                                VmFloatSliceEval 794-1086, VmGradSliceEval 1091-1397)
//   fidget-bytecode/src/lib.rs  (Bytecode::new 203-332)
#pragma once
#include <memory>

#include "compiler.hpp"

namespace orc {

struct VmData {
    SsaTape ssa;
    RegTape asm_;
    std::shared_ptr<VarMap> vars;
    uint32_t N = 255;

    size_t len() const { return asm_.len(); }
    size_t choice_count() const { return ssa.choice_count; }
    size_t output_count() const { return ssa.output_count; }
    size_t slot_count() const { return asm_.slot_count; }
};
typedef std::shared_ptr<VmData> VmDataP;

// data.rs:78-86
static inline VmDataP vmdata_new(const Context& ctx, const std::vector<Node>& roots, uint32_t N) {
    auto d = std::make_shared<VmData>();
    auto vars = std::make_shared<VarMap>();
    if (!ssa_tape_new(ctx, roots, d->ssa, *vars)) return nullptr;
    d->asm_ = reg_tape_new(d->ssa, N);
    d->vars = vars;
    d->N = N;
    return d;
}

// data.rs:355-408
struct VmWorkspace {
    RegisterAllocator alloc;
    std::vector<uint32_t> bind;
    uint32_t count = 0;
    explicit VmWorkspace(uint32_t N) : alloc(N) {}
    bool active(uint32_t i, uint32_t* out) const {
        if (bind[i] != UNASSIGNED) { *out = bind[i]; return true; }
        return false;
    }
    uint32_t get_or_insert_active(uint32_t i) {
        if (bind[i] == UNASSIGNED) bind[i] = count++;
        return bind[i];
    }
    void set_active(uint32_t i, uint32_t b) { bind[i] = b; }
    void reset(size_t tape_len) {
        alloc.reset(tape_len);
        bind.assign(tape_len, UNASSIGNED);
        count = 0;
    }
};

// data.rs:123-318.  Returns nullptr on a bad choice-slice length (BadChoiceSlice).
static inline VmDataP vmdata_simplify(const VmData& self, const uint8_t* choices, size_t n_choices, uint32_t M) {
    if (n_choices != self.choice_count()) return nullptr;
    VmWorkspace ws(M);
    ws.reset(self.ssa.tape.size());
    size_t choice_count = 0, output_count = 0;
    size_t ci = n_choices;  // choices.iter().rev()
    auto next_choice = [&]() -> uint8_t { assert(ci > 0); return choices[--ci]; };
    std::vector<TOp> ops_out;
    ops_out.reserve(self.ssa.tape.size());

    for (TOp op : self.ssa.tape) {
        if (op.op == O_OUTPUT) {
            op.a = ws.get_or_insert_active(op.a);
            ws.alloc.op(op);
            ops_out.push_back(op);
            output_count++;
            continue;
        }
        uint32_t index = op.out;
        uint32_t new_index;
        if (!ws.active(index, &new_index)) {
            if (has_choice(op)) next_choice();
            continue;
        }
        switch (op.op) {
            case O_INPUT:
            case O_COPY_IMM: op.out = new_index; break;
            case O_COPY_REG: {
                uint32_t new_src;
                if (ws.active(op.a, &new_src)) {
                    op.out = new_index;
                    op.a = new_src;
                } else {
                    ws.set_active(op.a, new_index);
                    continue;
                }
                break;
            }
            case O_MIN:
            case O_MAX:
            case O_AND:
            case O_OR: {
                uint8_t c = next_choice();
                if (op.form == F_REG_IMM) {
                    if (c == LEFT) {
                        uint32_t new_arg;
                        if (ws.active(op.a, &new_arg)) {
                            op = TOp{O_COPY_REG, F_REG, new_index, new_arg, 0, 0, 0};
                        } else {
                            ws.set_active(op.a, new_index);
                            continue;
                        }
                    } else if (c == RIGHT) {
                        op = TOp{O_COPY_IMM, F_NONE, new_index, 0, 0, 0, op.imm};
                    } else if (c == BOTH) {
                        choice_count++;
                        op.out = new_index;
                        op.a = ws.get_or_insert_active(op.a);
                    } else {
                        fprintf(stderr, "oracle: Choice::Unknown in simplify\n");
                        abort();
                    }
                } else {  // RegReg
                    if (c == LEFT) {
                        uint32_t nl;
                        if (ws.active(op.a, &nl)) {
                            op = TOp{O_COPY_REG, F_REG, new_index, nl, 0, 0, 0};
                        } else {
                            ws.set_active(op.a, new_index);
                            continue;
                        }
                    } else if (c == RIGHT) {
                        uint32_t nr;
                        if (ws.active(op.b, &nr)) {
                            op = TOp{O_COPY_REG, F_REG, new_index, nr, 0, 0, 0};
                        } else {
                            ws.set_active(op.b, new_index);
                            continue;
                        }
                    } else if (c == BOTH) {
                        choice_count++;
                        op.out = new_index;
                        op.a = ws.get_or_insert_active(op.a);
                        op.b = ws.get_or_insert_active(op.b);
                    } else {
                        fprintf(stderr, "oracle: Choice::Unknown in simplify\n");
                        abort();
                    }
                }
                break;
            }
            default:
                op.out = new_index;
                op.a = ws.get_or_insert_active(op.a);
                if (op.form == F_REG_REG) op.b = ws.get_or_insert_active(op.b);
        }
        ws.alloc.op(op);
        ops_out.push_back(op);
    }
    assert((size_t)ws.count + 1 == ops_out.size());
    auto d = std::make_shared<VmData>();
    d->ssa.tape = std::move(ops_out);
    d->ssa.choice_count = choice_count;
    d->ssa.output_count = output_count;
    d->asm_ = ws.alloc.finalize();
    d->vars = self.vars;
    d->N = M;
    return d;
}

// ---------------------------------------------------------------------------
// Per-type op tables.  `b` is the right operand for RegReg, Sem<T>::from(imm)
// for RegImm; for ImmReg the caller swaps (a = from(imm), b = reg).
template <class T> struct Sem;

template <> struct Sem<float> {
    typedef float V;
    static V from(float f) { return f; }
    static V unary(Opc o, V a) {
        switch (o) {
            case O_NEG: return -a;
            case O_ABS: return fabsf(a);
            case O_RECIP: return 1.0f / a;
            case O_SQRT: return sqrtf(a);
            case O_SQUARE: return a * a;
            case O_FLOOR: return floorf(a);
            case O_CEIL: return ceilf(a);
            case O_ROUND: return roundf(a);
            case O_SIN: return sinf(a);
            case O_COS: return cosf(a);
            case O_TAN: return tanf(a);
            case O_ASIN: return asinf(a);
            case O_ACOS: return acosf(a);
            case O_ATAN: return atanf(a);
            case O_EXP: return expf(a);
            case O_LN: return logf(a);
            case O_NOT: return f_not(a);
            case O_RAND: return f_rand(a);
            default: return NANF;
        }
    }
    static V mul_imm(V a, float imm) { return a * imm; }
    static V binary(Opc o, V a, V b, Choice* c) {
        switch (o) {
            case O_ADD: return a + b;
            case O_SUB: return a - b;
            case O_MUL: return a * b;
            case O_DIV: return a / b;
            case O_ATAN2: return atan2f(a, b);
            case O_COMPARE: return f_compare(a, b);
            case O_MIX: return f_mix(a, b);
            case O_MOD: return rem_euclid(a, b);
            case O_MIN: { FC r = f_min_choice(a, b); *c = r.c; return r.v; }
            case O_MAX: { FC r = f_max_choice(a, b); *c = r.c; return r.v; }
            case O_AND: { FC r = f_and_choice(a, b); *c = r.c; return r.v; }
            case O_OR: { FC r = f_or_choice(a, b); *c = r.c; return r.v; }
            default: return NANF;
        }
    }
};

template <> struct Sem<Interval> {
    typedef Interval V;
    static V from(float f) { return Interval(f); }
    static V unary(Opc o, V a) {
        switch (o) {
            case O_NEG: return i_neg(a);
            case O_ABS: return i_abs(a);
            case O_RECIP: return i_recip(a);
            case O_SQRT: return i_sqrt(a);
            case O_SQUARE: return i_square(a);
            case O_FLOOR: return i_floor(a);
            case O_CEIL: return i_ceil(a);
            case O_ROUND: return i_round(a);
            case O_SIN: return i_sin(a);
            case O_COS: return i_cos(a);
            case O_TAN: return i_tan(a);
            case O_ASIN: return i_asin(a);
            case O_ACOS: return i_acos(a);
            case O_ATAN: return i_atan(a);
            case O_EXP: return i_exp(a);
            case O_LN: return i_ln(a);
            case O_NOT: return i_not(a);
            case O_RAND: return i_rand(a);
            default: return I_nan();
        }
    }
    static V mul_imm(V a, float imm) { return i_mul_f(a, imm); }  // vm/mod.rs:410-412
    static V binary(Opc o, V a, V b, Choice* c) {
        switch (o) {
            case O_ADD: return i_add(a, b);
            case O_SUB: return i_sub(a, b);
            case O_MUL: return i_mul(a, b);
            case O_DIV: return i_div(a, b);
            case O_ATAN2: return i_atan2(a, b);
            case O_COMPARE: return i_compare(a, b);
            case O_MIX: return i_mix(a, b);
            case O_MOD: return i_rem_euclid(a, b);
            case O_MIN: { IC r = i_min_choice(a, b); *c = r.c; return r.v; }
            case O_MAX: { IC r = i_max_choice(a, b); *c = r.c; return r.v; }
            case O_AND: { IC r = i_and_choice(a, b); *c = r.c; return r.v; }
            case O_OR: { IC r = i_or_choice(a, b); *c = r.c; return r.v; }
            default: return I_nan();
        }
    }
};

template <> struct Sem<Grad> {
    typedef Grad V;
    static V from(float f) { return Grad(f); }
    static V unary(Opc o, V a) {
        switch (o) {
            case O_NEG: return g_neg(a);
            case O_ABS: return g_abs(a);
            case O_RECIP: return g_div(Grad(1.0f), a);  // vm/mod.rs:1127-1132
            case O_SQRT: return g_sqrt(a);
            case O_SQUARE: return g_mul(a, a);  // vm/mod.rs:1138-1143
            case O_FLOOR: return g_floor(a);
            case O_CEIL: return g_ceil(a);
            case O_ROUND: return g_round(a);
            case O_SIN: return g_sin(a);
            case O_COS: return g_cos(a);
            case O_TAN: return g_tan(a);
            case O_ASIN: return g_asin(a);
            case O_ACOS: return g_acos(a);
            case O_ATAN: return g_atan(a);
            case O_EXP: return g_exp(a);
            case O_LN: return g_ln(a);
            case O_NOT: return g_not(a);
            case O_RAND: return g_rand(a);
            default: return Grad(NANF);
        }
    }
    static V mul_imm(V a, float imm) { return g_mul_f(a, imm); }  // vm/mod.rs:1219-1223
    static V binary(Opc o, V a, V b, Choice*) {
        switch (o) {
            case O_ADD: return g_add(a, b);
            case O_SUB: return g_sub(a, b);
            case O_MUL: return g_mul(a, b);
            case O_DIV: return g_div(a, b);
            case O_ATAN2: return g_atan2(a, b);
            case O_COMPARE: return g_compare(a, b);
            case O_MIX: return g_mix(a, b);
            case O_MOD: return g_rem_euclid(a, b);
            case O_MIN: return g_min(a, b);
            case O_MAX: return g_max(a, b);
            case O_AND: return g_and(a, b);
            case O_OR: return g_or(a, b);
            default: return Grad(NANF);
        }
    }
};

// Tracing evaluator (vm/mod.rs:297-538 for Interval, 541-760 for f32).
// `choices` must hold choice_count entries; it is reset to Unknown here
// (resize_slots, 314-319).  Returns whether a trace should be reported
// (`simplify`, 529-536).  Fails (returns -1) on too few vars (var/mod.rs:151-165).
template <class T>
struct TracingEval {
    std::vector<T> slots;
    std::vector<T> out;
    std::vector<uint8_t> choices;
    int eval(const VmData& tape, const T* vars, size_t nvars) {
        if ((int)nvars < tape.vars->len()) return -1;
        if (slots.size() < tape.slot_count()) slots.resize(tape.slot_count(), Sem<T>::from(NANF));
        out.resize(tape.output_count(), Sem<T>::from(NANF));
        choices.assign(tape.choice_count(), UNKNOWN);
        bool simplify = false;
        size_t ci = 0;
        T* v = slots.data();
        const std::vector<TOp>& ops = tape.asm_.tape;
        for (size_t k = ops.size(); k-- > 0;) {  // iter_asm = reversed (data.rs:321-323)
            const TOp& op = ops[k];
            switch (op.op) {
                case O_OUTPUT: out[op.idx] = v[op.a]; break;
                case O_INPUT: v[op.out] = vars[op.idx]; break;
                case O_COPY_REG: v[op.out] = v[op.a]; break;
                case O_COPY_IMM: v[op.out] = Sem<T>::from(op.imm); break;
                case O_LOAD: v[op.out] = v[op.idx]; break;
                case O_STORE: v[op.idx] = v[op.a]; break;
                default:
                    if (is_unary(op.op)) {
                        v[op.out] = Sem<T>::unary(op.op, v[op.a]);
                    } else {
                        Choice c = BOTH;
                        T r;
                        if (op.form == F_REG_REG) r = Sem<T>::binary(op.op, v[op.a], v[op.b], &c);
                        else if (op.form == F_REG_IMM) {
                            if (op.op == O_MUL) r = Sem<T>::mul_imm(v[op.a], op.imm);
                            else r = Sem<T>::binary(op.op, v[op.a], Sem<T>::from(op.imm), &c);
                        } else r = Sem<T>::binary(op.op, Sem<T>::from(op.imm), v[op.a], &c);
                        v[op.out] = r;
                        if (has_choice(op)) {
                            choices[ci++] |= c;
                            simplify |= (c != BOTH);
                        }
                    }
            }
        }
        return simplify ? 1 : 0;
    }
};

// Bulk evaluator (vm/mod.rs:766-1397): slots[reg][lane], one inner loop per op.
template <class T>
struct BulkEval {
    std::vector<std::vector<T>> slots;
    std::vector<std::vector<T>> out;
    // vars[i] points at `size` values.  Returns -1 if too few vars.
    int eval(const VmData& tape, const T* const* vars, size_t nvars, size_t size) {
        if ((int)nvars < tape.vars->len()) return -1;
        if (slots.size() < tape.slot_count()) slots.resize(tape.slot_count());
        for (auto& s : slots) if (s.size() < size) s.resize(size, Sem<T>::from(NANF));
        out.resize(tape.output_count());
        for (auto& o : out) o.assign(size, Sem<T>::from(NANF));
        const std::vector<TOp>& ops = tape.asm_.tape;
        Choice dummy;
        for (size_t k = ops.size(); k-- > 0;) {
            const TOp& op = ops[k];
            switch (op.op) {
                case O_OUTPUT: std::copy(slots[op.a].begin(), slots[op.a].begin() + size, out[op.idx].begin()); break;
                case O_INPUT: std::copy(vars[op.idx], vars[op.idx] + size, slots[op.out].begin()); break;
                case O_COPY_REG: { T* o = slots[op.out].data(); const T* a = slots[op.a].data(); for (size_t i = 0; i < size; i++) o[i] = a[i]; break; }
                case O_COPY_IMM: { T* o = slots[op.out].data(); T im = Sem<T>::from(op.imm); for (size_t i = 0; i < size; i++) o[i] = im; break; }
                case O_LOAD: { T* o = slots[op.out].data(); const T* a = slots[op.idx].data(); for (size_t i = 0; i < size; i++) o[i] = a[i]; break; }
                case O_STORE: { T* o = slots[op.idx].data(); const T* a = slots[op.a].data(); for (size_t i = 0; i < size; i++) o[i] = a[i]; break; }
                default: {
                    T* o = slots[op.out].data();
                    const T* a = slots[op.a].data();
                    if (is_unary(op.op)) {
                        bulk_unary(op.op, o, a, size);
                    } else if (op.form == F_REG_REG) {
                        const T* b = slots[op.b].data();
                        for (size_t i = 0; i < size; i++) o[i] = Sem<T>::binary(op.op, a[i], b[i], &dummy);
                    } else if (op.form == F_REG_IMM) {
                        if (op.op == O_MUL) {
                            for (size_t i = 0; i < size; i++) o[i] = Sem<T>::mul_imm(a[i], op.imm);
                        } else {
                            T im = Sem<T>::from(op.imm);
                            bulk_reg_imm(op.op, o, a, im, size);
                        }
                    } else {
                        T im = Sem<T>::from(op.imm);
                        for (size_t i = 0; i < size; i++) o[i] = Sem<T>::binary(op.op, im, a[i], &dummy);
                    }
                }
            }
        }
        return 0;
    }
    // Hoist the opcode switch out of the lane loop for the common ops so the
    // CPU baseline is a fair restatement of the reference's per-op loops.
    static void bulk_unary(Opc opc, T* o, const T* a, size_t size) {
#define ORC_U(OPC) case OPC: for (size_t i = 0; i < size; i++) o[i] = Sem<T>::unary(OPC, a[i]); break;
        switch (opc) {
            ORC_U(O_NEG) ORC_U(O_ABS) ORC_U(O_SQRT) ORC_U(O_SQUARE) ORC_U(O_RECIP)
            default: for (size_t i = 0; i < size; i++) o[i] = Sem<T>::unary(opc, a[i]);
        }
#undef ORC_U
    }
    static void bulk_reg_imm(Opc opc, T* o, const T* a, T im, size_t size) {
        Choice dummy;
#define ORC_B(OPC) case OPC: for (size_t i = 0; i < size; i++) o[i] = Sem<T>::binary(OPC, a[i], im, &dummy); break;
        switch (opc) {
            ORC_B(O_ADD) ORC_B(O_SUB) ORC_B(O_MIN) ORC_B(O_MAX) ORC_B(O_DIV)
            default: for (size_t i = 0; i < size; i++) o[i] = Sem<T>::binary(opc, a[i], im, &dummy);
        }
#undef ORC_B
    }
};

// ---------------------------------------------------------------------------
// fidget-bytecode/src/lib.rs:69-104 (BytecodeOp numbering) and 203-332
enum BytecodeOp : uint8_t {
    BC_OUTPUT, BC_INPUT, BC_COPY, BC_NEG, BC_ABS, BC_RECIP, BC_SQRT, BC_SQUARE, BC_FLOOR, BC_CEIL,
    BC_ROUND, BC_NOT, BC_RAND, BC_SIN, BC_COS, BC_TAN, BC_ASIN, BC_ACOS, BC_ATAN, BC_EXP, BC_LN,
    BC_ADD, BC_SUB, BC_MUL, BC_DIV, BC_ATAN2, BC_COMPARE, BC_MIX, BC_MOD, BC_MIN, BC_MAX, BC_AND,
    BC_OR, BC_MEM
};
static inline BytecodeOp bytecode_op(Opc o) {
    switch (o) {
        case O_OUTPUT: return BC_OUTPUT;
        case O_INPUT: return BC_INPUT;
        case O_COPY_REG:
        case O_COPY_IMM: return BC_COPY;
        case O_NEG: return BC_NEG;
        case O_ABS: return BC_ABS;
        case O_RECIP: return BC_RECIP;
        case O_SQRT: return BC_SQRT;
        case O_SQUARE: return BC_SQUARE;
        case O_FLOOR: return BC_FLOOR;
        case O_CEIL: return BC_CEIL;
        case O_ROUND: return BC_ROUND;
        case O_SIN: return BC_SIN;
        case O_COS: return BC_COS;
        case O_TAN: return BC_TAN;
        case O_ASIN: return BC_ASIN;
        case O_ACOS: return BC_ACOS;
        case O_ATAN: return BC_ATAN;
        case O_EXP: return BC_EXP;
        case O_LN: return BC_LN;
        case O_NOT: return BC_NOT;
        case O_RAND: return BC_RAND;
        case O_ADD: return BC_ADD;
        case O_SUB: return BC_SUB;
        case O_MUL: return BC_MUL;
        case O_DIV: return BC_DIV;
        case O_ATAN2: return BC_ATAN2;
        case O_COMPARE: return BC_COMPARE;
        case O_MIX: return BC_MIX;
        case O_MOD: return BC_MOD;
        case O_MIN: return BC_MIN;
        case O_MAX: return BC_MAX;
        case O_AND: return BC_AND;
        case O_OR: return BC_OR;
        case O_LOAD:
        case O_STORE: return BC_MEM;
    }
    return BC_MEM;
}
struct Bytecode {
    uint8_t reg_count = 0;
    uint32_t mem_count = 0;
    std::vector<uint32_t> data;
    bool reserved_register = false;
};
static inline Bytecode bytecode_new(const VmData& t) {
    Bytecode bc;
    auto map = repack_map(t.asm_);
    bc.data = {0xFFFFFFFFu, 0u};
    const uint32_t mem_offset = t.N;
    const std::vector<TOp>& ops = t.asm_.tape;
    for (size_t k = ops.size(); k-- > 0;) {
        const TOp& op = ops[k];
        uint8_t word[4] = {0xFF, 0xFF, 0xFF, 0xFF};
        bool has_imm = false;
        uint32_t imm = 0;
        auto store_reg = [&](int i, uint32_t r) {
            uint8_t m = map[(uint8_t)r];
            if (m == 0xFF) { bc.reserved_register = true; return; }
            bc.reg_count = std::max<uint8_t>(bc.reg_count, m + 1);
            word[i] = m;
        };
        switch (op.op) {
            case O_INPUT: store_reg(1, op.out); imm = op.idx; has_imm = true; break;
            case O_OUTPUT: store_reg(1, op.a); imm = op.idx; has_imm = true; break;
            case O_LOAD:
                store_reg(1, op.out);
                word[2] = 0xFF;
                bc.mem_count = std::max(bc.mem_count, op.idx + 1 - mem_offset);
                imm = op.idx - mem_offset; has_imm = true;
                break;
            case O_STORE:
                store_reg(2, op.a);
                word[1] = 0xFF;
                bc.mem_count = std::max(bc.mem_count, op.idx + 1 - mem_offset);
                imm = op.idx - mem_offset; has_imm = true;
                break;
            case O_COPY_IMM: store_reg(1, op.out); word[2] = 0xFF; imm = f2u(op.imm); has_imm = true; break;
            default:
                if (op.form == F_REG) { store_reg(1, op.out); store_reg(2, op.a); }
                else if (op.form == F_REG_IMM) { store_reg(1, op.out); store_reg(2, op.a); word[3] = 0xFF; imm = f2u(op.imm); has_imm = true; }
                else if (op.form == F_IMM_REG) { store_reg(1, op.out); store_reg(3, op.a); word[2] = 0xFF; imm = f2u(op.imm); has_imm = true; }
                else { store_reg(1, op.out); store_reg(2, op.a); store_reg(3, op.b); }
        }
        word[0] = (uint8_t)bytecode_op(op.op);
        uint32_t w = (uint32_t)word[0] | ((uint32_t)word[1] << 8) | ((uint32_t)word[2] << 16) | ((uint32_t)word[3] << 24);
        bc.data.push_back(w);
        bc.data.push_back(has_imm ? imm : 0xFF000000u);
    }
    bc.data.push_back(0xFFFFFFFFu);
    bc.data.push_back(0xFFFFFFFFu);
    return bc;
}

}  // namespace orc
