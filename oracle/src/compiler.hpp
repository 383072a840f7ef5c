// ORACLE — TEST INFRASTRUCTURE ONLY (see types.hpp header).
//
// Restates graph -> SSA tape -> register-allocated tape:
//   fidget-core/src/compiler/op.rs        (SsaOp / RegOp variants, 4-306)
//   fidget-core/src/compiler/ssa_tape.rs  (SsaTape::new, 39-261)
//   fidget-core/src/compiler/lru.rs       (Lru, 19-76)
//   fidget-core/src/compiler/alloc.rs     (RegisterAllocator, 13-708)
//   fidget-core/src/compiler/reg_tape.rs  (RegTape::new 26-32, repack_map 46-61)
//
// The reference's RegisterAllocator<N> / Lru<N> are const-generic; here N is a
// run-time field so one binary serves VmFunction (N=255), GenericVmFunction<3>
// and the N=2 load/store known-answer tests.
#pragma once
#include <algorithm>
#include <cassert>

#include "context.hpp"

namespace orc {

// Base operation (RegReg / RegImm / ImmReg is carried in `form`)
enum Opc : uint8_t {
    O_OUTPUT, O_INPUT, O_COPY_REG, O_COPY_IMM,
    O_NEG, O_ABS, O_RECIP, O_SQRT, O_SQUARE, O_FLOOR, O_CEIL, O_ROUND, O_SIN, O_COS, O_TAN,
    O_ASIN, O_ACOS, O_ATAN, O_EXP, O_LN, O_NOT, O_RAND,
    O_ADD, O_SUB, O_MUL, O_DIV, O_ATAN2, O_COMPARE, O_MIX, O_MOD, O_MIN, O_MAX, O_AND, O_OR,
    O_LOAD, O_STORE
};
enum Form : uint8_t { F_NONE, F_REG, F_REG_REG, F_REG_IMM, F_IMM_REG };

// One struct for both SsaOp (u32 registers) and RegOp (u8 registers + memory
// slots >= N), compiler/op.rs:162 / 298-306.
//   Output(reg, i):  a = reg, idx = i
//   Input(out, i):   out, idx = i
//   CopyImm(out,imm) / unary(out, a) / RegImm & ImmReg(out, a, imm) / RegReg(out, a, b)
//   Load(reg, mem):  out = reg, idx = mem ; Store(reg, mem): a = reg, idx = mem
struct TOp {
    Opc op;
    Form form;
    uint32_t out, a, b, idx;
    float imm;
};
static inline bool is_unary(Opc o) { return o >= O_NEG && o <= O_RAND; }
static inline bool is_binary(Opc o) { return o >= O_ADD && o <= O_OR; }
static inline bool has_choice(const TOp& t) { return t.op >= O_MIN && t.op <= O_OR; }

static inline Opc unary_opc(UnaryOpcode u) { return (Opc)(O_NEG + (int)u); }
static inline Opc binary_opc(BinaryOpcode b) {
    switch (b) {
        case B_ADD: return O_ADD;
        case B_SUB: return O_SUB;
        case B_MUL: return O_MUL;
        case B_DIV: return O_DIV;
        case B_ATAN: return O_ATAN2;
        case B_MIN: return O_MIN;
        case B_MAX: return O_MAX;
        case B_COMPARE: return O_COMPARE;
        case B_MOD: return O_MOD;
        case B_AND: return O_AND;
        case B_OR: return O_OR;
        case B_MIX: return O_MIX;
    }
    return O_ADD;
}
static inline bool commutes_to_reg_imm(Opc o) {
    // ssa_tape.rs:131-165: Add/Mul/Min/Max use the RegImm form for (imm, reg)
    return o == O_ADD || o == O_MUL || o == O_MIN || o == O_MAX;
}

// ssa_tape.rs:22-32
struct SsaTape {
    std::vector<TOp> tape;  // root first (reverse evaluation order)
    size_t choice_count = 0;
    size_t output_count = 0;
};

// ssa_tape.rs:39-261
static inline bool ssa_tape_new(const Context& ctx, const std::vector<Node>& roots, SsaTape& out, VarMap& vars) {
    const size_t n = ctx.len();
    struct Slot { uint8_t kind; uint32_t reg; float imm; };  // 0 = unset, 1 = Reg, 2 = Immediate
    std::vector<Slot> mapping(n, Slot{0, 0, 0});
    std::vector<uint32_t> parent_count(n, 0);
    uint32_t slot_count = 0;

    std::vector<uint8_t> seen(n, 0);
    std::vector<Node> todo(roots.begin(), roots.end());
    while (!todo.empty()) {
        Node node = todo.back();
        todo.pop_back();
        if (node >= n) return false;
        if (seen[node]) continue;
        seen[node] = 1;
        const NodeOp& op = ctx.ops[node];
        if (op.kind == N_CONST) {
            mapping[node] = Slot{2, 0, op.c};
        } else {
            if (op.kind == N_INPUT) vars.insert(op.var);
            mapping[node] = Slot{1, slot_count++, 0};
        }
        if (op.kind == N_BINARY) {
            parent_count[op.a]++; todo.push_back(op.a);
            parent_count[op.b]++; todo.push_back(op.b);
        } else if (op.kind == N_UNARY) {
            parent_count[op.a]++; todo.push_back(op.a);
        }
    }

    std::fill(seen.begin(), seen.end(), 0);
    todo.assign(roots.begin(), roots.end());
    size_t choice_count = 0;
    std::vector<TOp>& tape = out.tape;
    tape.clear();
    for (size_t i = 0; i < roots.size(); i++) {
        const Slot& s = mapping[roots[i]];
        if (s.kind == 1) {
            tape.push_back(TOp{O_OUTPUT, F_NONE, 0, s.reg, 0, (uint32_t)i, 0});
        } else {
            uint32_t o = slot_count++;
            tape.push_back(TOp{O_OUTPUT, F_NONE, 0, o, 0, (uint32_t)i, 0});
            tape.push_back(TOp{O_COPY_IMM, F_NONE, o, 0, 0, 0, s.imm});
        }
    }
    while (!todo.empty()) {
        Node node = todo.back();
        todo.pop_back();
        if (parent_count[node] > 0) continue;
        if (seen[node]) continue;
        seen[node] = 1;
        const NodeOp& op = ctx.ops[node];
        if (op.kind == N_BINARY) {
            todo.push_back(op.a); parent_count[op.a]--;
            todo.push_back(op.b); parent_count[op.b]--;
        } else if (op.kind == N_UNARY) {
            todo.push_back(op.a); parent_count[op.a]--;
        }
        const Slot& me = mapping[node];
        if (me.kind != 1) continue;  // constants become immediates
        uint32_t i = me.reg;
        TOp t{};
        switch (op.kind) {
            case N_INPUT:
                t = TOp{O_INPUT, F_NONE, i, 0, 0, (uint32_t)vars.get(op.var), 0};
                break;
            case N_CONST: assert(false); break;
            case N_BINARY: {
                Opc o = binary_opc((BinaryOpcode)op.opcode);
                if (o >= O_MIN && o <= O_OR) choice_count++;
                const Slot& l = mapping[op.a];
                const Slot& r = mapping[op.b];
                if (l.kind == 1 && r.kind == 1) {
                    t = TOp{o, F_REG_REG, i, l.reg, r.reg, 0, 0};
                } else if (l.kind == 1 && r.kind == 2) {
                    t = TOp{o, F_REG_IMM, i, l.reg, 0, 0, r.imm};
                } else if (l.kind == 2 && r.kind == 1) {
                    if (o == O_AND || o == O_OR) {
                        fprintf(stderr, "oracle: And/Or ImmReg must be collapsed\n");
                        abort();
                    }
                    t = TOp{o, commutes_to_reg_imm(o) ? F_REG_IMM : F_IMM_REG, i, r.reg, 0, 0, l.imm};
                } else {
                    fprintf(stderr, "oracle: cannot handle f(imm, imm)\n");
                    abort();
                }
                break;
            }
            case N_UNARY: {
                const Slot& l = mapping[op.a];
                if (l.kind != 1) { fprintf(stderr, "oracle: cannot handle f(imm)\n"); abort(); }
                t = TOp{unary_opc((UnaryOpcode)op.opcode), F_REG, i, l.reg, 0, 0, 0};
                break;
            }
        }
        tape.push_back(t);
    }
    out.choice_count = choice_count;
    out.output_count = roots.size();
    return true;
}

// lru.rs:19-76
struct Lru {
    struct LNode { uint8_t prev, next; };
    std::vector<LNode> data;
    uint8_t head = 0;
    Lru() {}
    explicit Lru(int n) : data(n), head(0) {
        for (int i = 0; i < n; i++) {
            data[i].next = (uint8_t)((i + 1) % n);
            data[i].prev = (uint8_t)(i == 0 ? n - 1 : i - 1);
        }
    }
    void remove(uint8_t i) {
        LNode node = data[i];
        data[node.prev].next = data[i].next;
        data[node.next].prev = data[i].prev;
    }
    void insert_before(uint8_t i, uint8_t next) {
        uint8_t prev = data[next].prev;
        data[prev].next = i;
        data[next].prev = i;
        data[i] = LNode{prev, next};
    }
    void poke(uint8_t i) {
        uint8_t prev_newest = head;
        if (prev_newest == i) return;
        if (data[prev_newest].prev != i) {
            remove(i);
            insert_before(i, head);
        }
        head = i;
    }
    uint8_t pop() {
        uint8_t out = data[head].prev;
        head = out;
        return out;
    }
};

// reg_tape.rs:9-17
struct RegTape {
    std::vector<TOp> tape;  // root first, like the SSA tape
    uint32_t slot_count = 0;
    size_t len() const { return tape.size(); }
};

static const uint32_t UNASSIGNED = 0xFFFFFFFFu;

// alloc.rs:13-708
struct RegisterAllocator {
    uint32_t N;
    std::vector<uint32_t> allocations;
    std::vector<uint32_t> registers;
    Lru register_lru;
    std::vector<uint8_t> spare_registers;
    std::vector<uint32_t> spare_memory;
    RegTape out;

    struct Alloc { int kind; uint32_t v; };  // 0 = Register, 1 = Memory, 2 = Unassigned

    explicit RegisterAllocator(uint32_t n) : N(n) {}

    // alloc.rs:87-98 (new() at 50-63 is the same state)
    void reset(size_t size) {
        assert(N <= 255);
        allocations.assign(size, UNASSIGNED);
        registers.assign(N, UNASSIGNED);
        register_lru = Lru((int)N);
        spare_registers.clear();
        for (int i = (int)N - 1; i >= 0; i--) spare_registers.push_back((uint8_t)i);
        spare_memory.clear();
        out.tape.clear();
        out.slot_count = 0;
    }
    RegTape finalize() {
        RegTape t;
        std::swap(t, out);
        return t;
    }
    uint32_t get_memory() {  // 116-125
        if (!spare_memory.empty()) {
            uint32_t p = spare_memory.back();
            spare_memory.pop_back();
            return p;
        }
        uint32_t o = out.slot_count;
        out.slot_count += 1;
        assert(o >= N);
        return o;
    }
    uint8_t oldest_reg() { return register_lru.pop(); }
    Alloc get_allocation(uint32_t n) {  // 142-151
        uint32_t i = allocations[n];
        if (i < N) {
            register_lru.poke((uint8_t)i);
            return Alloc{0, i};
        }
        if (i == UNASSIGNED) return Alloc{2, 0};
        return Alloc{1, i};
    }
    bool get_spare_register(uint8_t* r) {  // 155-159
        if (spare_registers.empty()) return false;
        *r = spare_registers.back();
        spare_registers.pop_back();
        out.slot_count = std::max(out.slot_count, (uint32_t)*r + 1);
        return true;
    }
    uint8_t get_register() {  // 162-184
        uint8_t reg;
        if (get_spare_register(&reg)) {
            assert(registers[reg] == UNASSIGNED);
            register_lru.poke(reg);
            return reg;
        }
        reg = oldest_reg();
        uint32_t mem = get_memory();
        uint32_t prev_node = registers[reg];
        allocations[prev_node] = mem;
        registers[reg] = UNASSIGNED;
        out.tape.push_back(TOp{O_LOAD, F_NONE, reg, 0, 0, mem, 0});
        return reg;
    }
    void rebind_register(uint32_t n, uint8_t reg) {  // 187-198
        assert(allocations[n] >= N);
        assert(registers[reg] != UNASSIGNED);
        uint32_t prev_node = registers[reg];
        allocations[prev_node] = UNASSIGNED;
        registers[reg] = n;
        allocations[n] = reg;
    }
    void bind_register(uint32_t n, uint8_t reg) {  // 201-209
        assert(allocations[n] >= N);
        assert(registers[reg] == UNASSIGNED);
        registers[reg] = n;
        allocations[n] = reg;
    }
    void release_reg(uint8_t reg) {  // 213-225
        assert(reg < N);
        uint32_t node = registers[reg];
        assert(node != UNASSIGNED);
        registers[reg] = UNASSIGNED;
        spare_registers.push_back(reg);
        allocations[node] = UNASSIGNED;
    }
    void release_mem(uint32_t mem) {  // 228-233
        assert(mem >= N);
        spare_memory.push_back(mem);
    }
    void push_store(uint8_t reg, uint32_t mem) {  // 329-332
        out.tape.push_back(TOp{O_STORE, F_NONE, 0, reg, 0, mem, 0});
        release_mem(mem);
    }
    uint8_t get_out_reg(uint32_t o) {  // 340-354
        Alloc a = get_allocation(o);
        if (a.kind == 0) return (uint8_t)a.v;
        if (a.kind == 1) {
            uint8_t r_a = get_register();
            push_store(r_a, a.v);
            bind_register(o, r_a);
            return r_a;
        }
        fprintf(stderr, "oracle: cannot have unassigned output\n");
        abort();
    }
    // One-register form (unary, RegImm, ImmReg), alloc.rs:357-406
    void op_reg_fn(const TOp& src) {
        uint8_t r_x = get_out_reg(src.out);
        Alloc a = get_allocation(src.a);
        TOp t = src;
        t.out = r_x;
        if (a.kind == 0) {
            assert(r_x != a.v);
            t.a = a.v;
            out.tape.push_back(t);
            release_reg(r_x);
        } else if (a.kind == 1) {
            uint8_t r_a = get_register();
            push_store(r_a, a.v);
            t.a = r_a;
            out.tape.push_back(t);
            release_reg(r_x);
            bind_register(src.a, r_a);
        } else {
            t.a = r_x;
            out.tape.push_back(t);
            rebind_register(src.a, r_x);
        }
    }
    // Two-register form, alloc.rs:419-604
    void op_reg_reg(const TOp& src) {
        uint8_t r_x = get_out_reg(src.out);
        uint32_t lhs = src.a, rhs = src.b;
        Alloc L = get_allocation(lhs);
        Alloc R = get_allocation(rhs);
        TOp t = src;
        t.out = r_x;
        auto emit = [&](uint32_t a, uint32_t b) { t.a = a; t.b = b; out.tape.push_back(t); };
        if (L.kind == 0 && R.kind == 0) {
            emit(L.v, R.v);
            release_reg(r_x);
        } else if (L.kind == 1 && R.kind == 0) {
            uint8_t r_a = get_register();
            push_store(r_a, L.v);
            emit(r_a, R.v);
            release_reg(r_x);
            bind_register(lhs, r_a);
        } else if (L.kind == 0 && R.kind == 1) {
            uint8_t r_a = get_register();
            push_store(r_a, R.v);
            emit(L.v, r_a);
            release_reg(r_x);
            bind_register(rhs, r_a);
        } else if (L.kind == 1 && R.kind == 1 && lhs == rhs) {
            uint8_t r_a = get_register();
            push_store(r_a, L.v);
            emit(r_a, r_a);
            release_reg(r_x);
            bind_register(lhs, r_a);
        } else if (L.kind == 1 && R.kind == 1) {
            uint8_t r_a = get_register();
            uint8_t r_b = get_register();
            push_store(r_a, L.v);
            push_store(r_b, R.v);
            emit(r_a, r_b);
            release_reg(r_x);
            bind_register(lhs, r_a);
            bind_register(rhs, r_b);
        } else if (L.kind == 2 && R.kind == 0) {
            emit(r_x, R.v);
            rebind_register(lhs, r_x);
        } else if (L.kind == 0 && R.kind == 2) {
            emit(L.v, r_x);
            rebind_register(rhs, r_x);
        } else if (L.kind == 2 && R.kind == 2 && lhs == rhs) {
            emit(r_x, r_x);
            rebind_register(lhs, r_x);
        } else if (L.kind == 2 && R.kind == 2) {
            uint8_t r_a = get_register();
            emit(r_x, r_a);
            rebind_register(lhs, r_x);
            bind_register(rhs, r_a);
        } else if (L.kind == 2 && R.kind == 1) {
            uint8_t r_a = get_register();
            assert(r_a != r_x);
            assert(lhs != rhs);
            push_store(r_a, R.v);
            emit(r_x, r_a);
            rebind_register(lhs, r_x);
            bind_register(rhs, r_a);
        } else {  // (Memory, Unassigned)
            uint8_t r_a = get_register();
            assert(r_a != r_x);
            assert(lhs != rhs);
            push_store(r_a, L.v);
            emit(r_a, r_x);
            bind_register(lhs, r_a);
            rebind_register(rhs, r_x);
        }
    }
    void op_out_only(const TOp& src) {  // 670-674
        uint8_t r_x = get_out_reg(src.out);
        TOp t = src;
        t.out = r_x;
        out.tape.push_back(t);
        release_reg(r_x);
    }
    void op_output(const TOp& src) {  // 692-707
        Alloc a = get_allocation(src.a);
        TOp t = src;
        if (a.kind == 0) {
            t.a = a.v;
            out.tape.push_back(t);
        } else if (a.kind == 1) {
            uint8_t r_a = get_register();
            push_store(r_a, a.v);
            t.a = r_a;
            out.tape.push_back(t);
            bind_register(src.a, r_a);
        } else {
            uint8_t r_a = get_register();
            t.a = r_a;
            out.tape.push_back(t);
            bind_register(src.a, r_a);
        }
    }
    // alloc.rs:269-327
    void op(const TOp& s) {
        switch (s.op) {
            case O_OUTPUT: op_output(s); break;
            case O_INPUT:
            case O_COPY_IMM: op_out_only(s); break;
            case O_LOAD:
            case O_STORE: assert(false); break;
            default:
                if (s.form == F_REG_REG) op_reg_reg(s);
                else op_reg_fn(s);  // CopyReg, unary, RegImm, ImmReg
        }
    }
};

// reg_tape.rs:26-32
static inline RegTape reg_tape_new(const SsaTape& ssa, uint32_t N) {
    RegisterAllocator alloc(N);
    alloc.reset(ssa.tape.size());
    for (const TOp& op : ssa.tape) alloc.op(op);
    return alloc.finalize();
}

// Visit every register of a RegOp (compiler/op.rs:310-470); memory slots and
// input/output indices are not registers.
template <class F>
static inline void visit_regs(const TOp& t, F f) {
    switch (t.op) {
        case O_OUTPUT: f(t.a); break;
        case O_INPUT:
        case O_COPY_IMM:
        case O_LOAD: f(t.out); break;
        case O_STORE: f(t.a); break;
        default:
            f(t.out);
            f(t.a);
            if (t.form == F_REG_REG) f(t.b);
    }
}

// reg_tape.rs:46-61: registers renumbered by descending use count (ties by
// ascending register number, from the `(Reverse(count), reg)` sort key)
static inline std::map<uint8_t, uint8_t> repack_map(const RegTape& rt) {
    std::map<uint8_t, size_t> counts;
    for (const TOp& t : rt.tape) visit_regs(t, [&](uint32_t r) { counts[(uint8_t)r]++; });
    std::vector<std::pair<size_t, uint8_t>> sorted;
    for (auto& kv : counts) sorted.push_back({kv.second, kv.first});
    std::sort(sorted.begin(), sorted.end(), [](const auto& a, const auto& b) {
        if (a.first != b.first) return a.first > b.first;
        return a.second < b.second;
    });
    std::map<uint8_t, uint8_t> m;
    for (size_t i = 0; i < sorted.size(); i++) m[sorted[i].second] = (uint8_t)i;
    return m;
}

}  // namespace orc
