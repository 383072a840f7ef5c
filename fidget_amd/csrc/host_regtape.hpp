// The reference's register allocator and wire format for a tape of this library (host side, no device involved).
//
// The device tapes of fidget-hip are allocated densely over up to 4096 registers and never spill (host_graph.hpp allocate; register
// files that do not fit a CU live in HBM).  The reference lowers the same SSA program differently: `RegisterAllocator<N>`
// (fidget-core/src/compiler/alloc.rs:13-708) walks it root first, keeps at most N values in registers - evicting the least
// recently used one (`Lru<N>`, compiler/lru.rs:19-76) to a memory slot >= N with a `Load` / `Store` pair - and `Bytecode::new`
// (fidget-bytecode/src/lib.rs:203-332) serialises the result as [op, out, lhs, rhs][imm] words with the registers renumbered by
// frequency (`RegTape::repack_map`, compiler/reg_tape.rs:46-61).  A caller of the reference's API that asks this backend for
// `VmData<N>`-shaped answers - `len()` with its loads and stores (vm/data.rs:415-436), `iter_asm()`, the bytecode of a tape - gets
// them from here: the tape's ops in SSA terms (its registers are single-assignment between writes: the last writer of a register is
// the value) through the same allocation, op for op.  What the device executes is unchanged.
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "host_graph.hpp"

namespace fh {

enum : uint8_t { FH_REG_LOAD = FH_OP_COUNT, FH_REG_STORE = FH_OP_COUNT + 1 };   // RegOp::Load / RegOp::Store (compiler/op.rs:298-306)

// One RegOp: FhOp opcodes keep their operand form; w = immediate bits / input or output slot / memory slot (Load, Store).
struct RegOp {
    uint8_t op, out, a, b;
    uint32_t w;
};

struct RegTapeOut {
    std::vector<RegOp> ops;      // as the reference keeps them: root first (iter_asm() is the reverse)
    uint32_t slot_count = 0;     // registers 0..N and memory slots N.. in use (reg_tape.rs:13-17)
};

namespace regtape_detail {
constexpr uint32_t NONE = 0xFFFFFFFFu;

// compiler/lru.rs:19-76: a ring of the N registers, `head` the most recently used, its predecessor the least
struct Ring {
    std::vector<uint8_t> prev, next;
    uint8_t head = 0;
    explicit Ring(uint32_t n) : prev(n), next(n) {
        for (uint32_t i = 0; i < n; i++) { next[i] = (uint8_t)((i + 1) % n); prev[i] = (uint8_t)(i ? i - 1 : n - 1); }
    }
    void touch(uint8_t i) {                     // Lru::poke
        if (head == i) return;
        if (prev[head] != i) {
            next[prev[i]] = next[i]; prev[next[i]] = prev[i];                       // out of the ring ...
            const uint8_t before = prev[head];
            next[before] = i; prev[head] = i; prev[i] = before; next[i] = head;       // ... and back in, just before the head
        }
        head = i;
    }
    uint8_t oldest() { head = prev[head]; return head; }                            // Lru::pop: the oldest becomes the newest
};

struct Alloc {
    const uint32_t N;
    std::vector<uint32_t> where;        // per SSA value: register (< N), memory slot (>= N) or NONE   (alloc.rs:14-21)
    std::vector<uint32_t> holder;       // per register: the SSA value in it or NONE                     (alloc.rs:23-29)
    Ring ring;
    std::vector<uint8_t> spare_regs;    // most recently released at the back                              (alloc.rs:36-39)
    std::vector<uint32_t> spare_mem;
    RegTapeOut out;
    bool starved = false;

    Alloc(uint32_t n, size_t values) : N(n), where(values, NONE), holder(n, NONE), ring(n) {
        for (uint32_t r = n; r-- > 0;) spare_regs.push_back((uint8_t)r);        // (0..N).rev(): register 0 is handed out first
    }
    void push(uint8_t op, uint8_t o, uint8_t a, uint8_t b, uint32_t w) { out.ops.push_back(RegOp{op, o, a, b, w}); }

    enum Kind { REG, MEM, UNSET };
    Kind look(uint32_t v, uint32_t& at) {                                         // get_allocation (alloc.rs:141-150)
        at = where[v];
        if (at == NONE) return UNSET;
        if (at < N) { ring.touch((uint8_t)at); return REG; }
        return MEM;
    }
    uint32_t fresh_mem() {                                                        // get_memory (alloc.rs:113-122)
        if (!spare_mem.empty()) { const uint32_t m = spare_mem.back(); spare_mem.pop_back(); return m; }
        if (out.slot_count < N) out.slot_count = N;        // (the reference asserts slot_count >= N here: every register is taken by now)
        return out.slot_count++;
    }
    uint8_t take_reg() {                                                          // get_register (alloc.rs:160-183)
        if (!spare_regs.empty()) {
            const uint8_t r = spare_regs.back();
            spare_regs.pop_back();
            out.slot_count = std::max(out.slot_count, (uint32_t)r + 1);
            ring.touch(r);
            return r;
        }
        const uint8_t r = ring.oldest();      // evict: when read forward, its value comes back from memory right after this op
        // (N too small for this op - its output and two operands from memory need three registers -: the reference trips an
        // assertion later, in release_reg; here the tape is refused)
        if (holder[r] == NONE) { starved = true; return r; }
        const uint32_t m = fresh_mem();
        where[holder[r]] = m;
        holder[r] = NONE;
        push(FH_REG_LOAD, r, 0, 0, m);
        return r;
    }
    void bind(uint32_t v, uint8_t r) { holder[r] = v; where[v] = r; }             // bind_register
    void rebind(uint32_t v, uint8_t r) { if (holder[r] == NONE) { starved = true; return; } where[holder[r]] = NONE; holder[r] = v; where[v] = r; }   // rebind_register
    void release(uint8_t r) { if (holder[r] == NONE) { starved = true; return; } where[holder[r]] = NONE; holder[r] = NONE; spare_regs.push_back(r); }   // release_reg
    void store(uint8_t r, uint32_t m) { push(FH_REG_STORE, 0, r, 0, m); spare_mem.push_back(m); }       // push_store + release_mem
    uint8_t out_reg(uint32_t v, bool& ok) {                                       // get_out_reg (alloc.rs:341-356)
        uint32_t at;
        switch (look(v, at)) {
            case REG: return (uint8_t)at;
            case MEM: { const uint8_t r = take_reg(); store(r, at); bind(v, r); return r; }
            default: ok = false; return 0;      // an op whose value nobody wants: not a tape the reference would hand over
        }
    }
    // one register operand (alloc.rs:358-412)
    void unary(uint8_t op, uint32_t o, uint32_t arg, uint32_t w, bool& ok) {
        const uint8_t rx = out_reg(o, ok);
        if (!ok) return;
        uint32_t at;
        switch (look(arg, at)) {
            case REG: push(op, rx, (uint8_t)at, 0, w); release(rx); break;
            case MEM: { const uint8_t ra = take_reg(); store(ra, at); push(op, rx, ra, 0, w); release(rx); bind(arg, ra); break; }
            case UNSET: push(op, rx, rx, 0, w); rebind(arg, rx); break;
        }
    }
    // two register operands: the table of alloc.rs:427-611
    void binary(uint8_t op, uint32_t o, uint32_t l, uint32_t r, bool& ok) {
        const uint8_t rx = out_reg(o, ok);
        if (!ok) return;
        uint32_t al, ar;
        const Kind kl = look(l, al), kr = look(r, ar);
        if (kl == REG && kr == REG) { push(op, rx, (uint8_t)al, (uint8_t)ar, 0); release(rx); }
        else if (kl == MEM && kr == REG) { const uint8_t ra = take_reg(); store(ra, al); push(op, rx, ra, (uint8_t)ar, 0); release(rx); bind(l, ra); }
        else if (kl == REG && kr == MEM) { const uint8_t ra = take_reg(); store(ra, ar); push(op, rx, (uint8_t)al, ra, 0); release(rx); bind(r, ra); }
        else if (kl == MEM && kr == MEM && l == r) { const uint8_t ra = take_reg(); store(ra, al); push(op, rx, ra, ra, 0); release(rx); bind(l, ra); }
        else if (kl == MEM && kr == MEM) {
            const uint8_t ra = take_reg(), rb = take_reg();
            store(ra, al); store(rb, ar);
            push(op, rx, ra, rb, 0);
            release(rx); bind(l, ra); bind(r, rb);
        }
        else if (kl == UNSET && kr == REG) { push(op, rx, rx, (uint8_t)ar, 0); rebind(l, rx); }
        else if (kl == REG && kr == UNSET) { push(op, rx, (uint8_t)al, rx, 0); rebind(r, rx); }
        else if (kl == UNSET && kr == UNSET && l == r) { push(op, rx, rx, rx, 0); rebind(l, rx); }
        else if (kl == UNSET && kr == UNSET) { const uint8_t ra = take_reg(); push(op, rx, rx, ra, 0); rebind(l, rx); bind(r, ra); }
        else if (kl == UNSET && kr == MEM) { const uint8_t ra = take_reg(); store(ra, ar); push(op, rx, rx, ra, 0); rebind(l, rx); bind(r, ra); }
        else { const uint8_t ra = take_reg(); store(ra, al); push(op, rx, ra, rx, 0); bind(l, ra); rebind(r, rx); }      // MEM, UNSET
    }
    void leaf(uint8_t op, uint32_t o, uint32_t w, bool& ok) {                     // op_out_only: CopyImm, Input
        const uint8_t rx = out_reg(o, ok);
        if (!ok) return;
        push(op, rx, 0, 0, w);
        release(rx);
    }
    void output(uint32_t arg, uint32_t slot) {                                    // op_output (alloc.rs:687-705)
        uint32_t at;
        switch (look(arg, at)) {
            case REG: push(FH_OUTPUT, 0, (uint8_t)at, 0, slot); break;
            case MEM: { const uint8_t ra = take_reg(); store(ra, at); push(FH_OUTPUT, 0, ra, 0, slot); bind(arg, ra); break; }
            case UNSET: { const uint8_t ra = take_reg(); push(FH_OUTPUT, 0, ra, 0, slot); bind(arg, ra); break; }
        }
    }
};
}  // namespace regtape_detail

// RegTape::new::<N> (reg_tape.rs:26-32) of the tape `t` (device format, evaluation order).  1 <= N <= 255.
static inline bool reg_tape(const HostTape& t, uint32_t N, RegTapeOut& out, std::string& err) {
    using namespace regtape_detail;
    if (N < 1 || N > 255) { err = "register count out of range (1..255)"; return false; }
    // the tape in SSA terms: a value per register write
    struct V { uint8_t op; uint32_t o, a, b, w; };
    std::vector<V> ssa;
    ssa.reserve(t.ops.size());
    std::vector<uint32_t> cur(FH_MAX_REGS, NONE);
    uint32_t next = 0;
    for (const uint64_t word : t.ops) {
        const uint32_t w0 = (uint32_t)word, w1 = (uint32_t)(word >> 32);
        const uint8_t op = (uint8_t)FH_W_OP(w0);
        const uint32_t ro = FH_W_OUT(w0), ra = FH_W_A(w0);
        V v{op, 0, 0, 0, w1};
        const bool has_a = op != FH_INPUT && op != FH_COPY_IMM;
        if (has_a) { if (ra >= FH_MAX_REGS || cur[ra] == NONE) { err = "operand read before it is written"; return false; } v.a = cur[ra]; }
        if (fh_is_rr(op)) { if (w1 >= FH_MAX_REGS || cur[w1] == NONE) { err = "operand read before it is written"; return false; } v.b = cur[w1]; }
        if (op != FH_OUTPUT) { v.o = next; cur[ro] = next++; }
        ssa.push_back(v);
    }
    Alloc al(N, next);
    bool ok = true;
    for (size_t i = ssa.size(); i-- > 0 && ok && !al.starved;) {      // root first, as SsaTape iterates
        const V& v = ssa[i];
        if (v.op == FH_OUTPUT) al.output(v.a, v.w);
        else if (v.op == FH_INPUT || v.op == FH_COPY_IMM) al.leaf(v.op, v.o, v.w, ok);
        else if (fh_is_rr(v.op)) al.binary(v.op, v.o, v.a, v.b, ok);
        else al.unary(v.op, v.o, v.a, v.w, ok);
    }
    if (al.starved) { err = "too few registers for this tape (the reference's allocator asserts)"; return false; }
    if (!ok) { err = "the tape holds an op whose value is never used"; return false; }
    out = std::move(al.out);
    return true;
}

// Bytecode::new::<N> (fidget-bytecode/src/lib.rs:203-332): evaluation order, registers renumbered by frequency, start and end
// markers.  false: the reserved register 255 would be in use (ReservedRegister).
static inline bool reg_tape_bytecode(const RegTapeOut& rt, uint32_t N, std::vector<uint32_t>& words, uint32_t& reg_count, uint32_t& mem_count) {
    static const uint8_t UN[] = {3, 4, 5, 6, 7, 8, 9, 10, 13, 14, 15, 16, 17, 18, 19, 20, 11, 12};      // FH_NEG .. FH_RAND -> BytecodeOp (lib.rs:69-104)
    auto regs_of = [](const RegOp& o, uint8_t r[3]) -> int {                        // RegOp::visit_regs (op.rs:393-468)
        if (o.op == FH_OUTPUT || o.op == FH_REG_STORE) { r[0] = o.a; return 1; }
        if (o.op == FH_INPUT || o.op == FH_COPY_IMM || o.op == FH_REG_LOAD) { r[0] = o.out; return 1; }
        r[0] = o.out; r[1] = o.a;
        if (fh_is_rr(o.op)) { r[2] = o.b; return 3; }
        return 2;
    };
    // repack_map (reg_tape.rs:46-61): most used register first, ties by register number
    size_t uses[256] = {};
    for (const RegOp& o : rt.ops) { uint8_t r[3]; const int n = regs_of(o, r); for (int k = 0; k < n; k++) uses[r[k]]++; }
    std::vector<uint32_t> order;
    for (uint32_t r = 0; r < 256; r++) if (uses[r]) order.push_back(r);
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return uses[x] != uses[y] ? uses[x] > uses[y] : x < y; });
    uint8_t map[256] = {};
    for (size_t i = 0; i < order.size(); i++) map[order[i]] = (uint8_t)i;
    words.assign({0xFFFFFFFFu, 0u});
    reg_count = mem_count = 0;
    bool reserved = false;
    auto reg = [&](uint8_t r) -> uint32_t { const uint8_t m = map[r]; if (m == 0xFF) reserved = true; reg_count = std::max<uint32_t>(reg_count, (uint32_t)m + 1); return m; };
    for (size_t i = rt.ops.size(); i-- > 0;) {
        const RegOp& o = rt.ops[i];
        uint32_t b0, b1 = 0xFF, b2 = 0xFF, b3 = 0xFF, imm = 0xFF000000u;
        if (o.op == FH_OUTPUT) { b0 = 0; b1 = reg(o.a); imm = o.w; }
        else if (o.op == FH_INPUT) { b0 = 1; b1 = reg(o.out); imm = o.w; }
        else if (o.op == FH_REG_LOAD) { b0 = 33; b1 = reg(o.out); mem_count = std::max(mem_count, o.w + 1 - N); imm = o.w - N; }
        else if (o.op == FH_REG_STORE) { b0 = 33; b2 = reg(o.a); mem_count = std::max(mem_count, o.w + 1 - N); imm = o.w - N; }
        else if (o.op == FH_COPY_IMM) { b0 = 2; b1 = reg(o.out); imm = o.w; }
        else if (o.op == FH_COPY_REG) { b0 = 2; b1 = reg(o.out); b2 = reg(o.a); }
        else if (fh_is_unary(o.op)) { b0 = UN[o.op - FH_NEG]; b1 = reg(o.out); b2 = reg(o.a); }
        else if (fh_is_rr(o.op)) { b0 = 21 + (o.op - FH_ADD_RR); b1 = reg(o.out); b2 = reg(o.a); b3 = reg(o.b); }
        else if (fh_is_ri(o.op)) { b0 = 21 + (o.op - FH_ADD_RI); b1 = reg(o.out); b2 = reg(o.a); imm = o.w; }
        else {      // imm (op) reg
            static const uint8_t IR[] = {22, 24, 25, 26, 27, 28};       // Sub, Div, Atan2, Compare, Mix, Mod
            b0 = IR[o.op - FH_SUB_IR]; b1 = reg(o.out); b3 = reg(o.a); imm = o.w;
        }
        words.push_back(b0 | (b1 << 8) | (b2 << 16) | (b3 << 24));
        words.push_back(imm);
    }
    words.push_back(0xFFFFFFFFu); words.push_back(0xFFFFFFFFu);
    return !reserved;
}

}  // namespace fh
