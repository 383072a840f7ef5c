// Host side of fidget-hip: the math-graph arena and the graph -> SSA -> device
// tape compiler.  In a Rust build of Fidget this role is played by fidget-core
// itself (`Context`, `SsaTape`, `VmData`), which hands `fidget_bytecode::Bytecode`
// words to fhip_tape_from_bytecode(); no Rust toolchain exists here, so the
// host mirror is C++ and reproduces the *observable* rules of the reference
// (node identity, operand order, tape order, variable numbering), because
// those define choice indices and which side of a min/max is `Left`:
//
//   Context ctor rules ......... fidget-core/src/context/mod.rs:188-780
//   .vm text format ............. fidget-core/src/context/mod.rs:878-941
//   graph -> SSA order ......... fidget-core/src/compiler/ssa_tape.rs:39-261
//   variable numbering ......... fidget-core/src/var/mod.rs:140-148
//
// Register allocation is NOT the reference's (compiler/alloc.rs); device tapes
// use a dense reverse-scan allocation shared with the on-device simplifier
// (see tape_format.h, kernels.hip: simplify_emit).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <map>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "tape_format.h"

namespace fh {

static inline uint32_t bits_of(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float float_of(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// ---- scalar semantics used for constant folding (context/op.rs:50-94) --------
static inline uint32_t pcg_hash(uint32_t v) {  // rng/mod.rs:8-13
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}
static inline float host_rem_euclid(float a, float b) {
    float r = fmodf(a, b);
    return r < 0.0f ? r + fabsf(b) : r;
}
static inline float fold_unary(int op, float a) {
    switch (op) {
        case FH_NEG: return -a;
        case FH_ABS: return fabsf(a);
        case FH_RECIP: return 1.0f / a;
        case FH_SQRT: return sqrtf(a);
        case FH_SQUARE: return a * a;
        case FH_FLOOR: return floorf(a);
        case FH_CEIL: return ceilf(a);
        case FH_ROUND: return roundf(a);
        case FH_SIN: return sinf(a);
        case FH_COS: return cosf(a);
        case FH_TAN: return tanf(a);
        case FH_ASIN: return asinf(a);
        case FH_ACOS: return acosf(a);
        case FH_ATAN: return atanf(a);
        case FH_EXP: return expf(a);
        case FH_LN: return logf(a);
        case FH_NOT: return a == 0.0f ? 1.0f : 0.0f;
        case FH_RAND: return float_of((pcg_hash(bits_of(a)) >> 9) | 0x3f800000u) - 1.0f;
    }
    return NAN;
}
static inline float fold_binary(int rr_op, float a, float b) {
    switch (rr_op) {
        case FH_ADD_RR: return a + b;
        case FH_SUB_RR: return a - b;
        case FH_MUL_RR: return a * b;
        case FH_DIV_RR: return a / b;
        case FH_ATAN2_RR: return atan2f(a, b);
        case FH_COMPARE_RR: return a < b ? -1.0f : (a == b ? 0.0f : (a > b ? 1.0f : NAN));
        case FH_MIX_RR: return float_of(pcg_hash(bits_of(a) + pcg_hash(bits_of(b))));
        case FH_MOD_RR: return host_rem_euclid(a, b);
        case FH_MIN_RR: return a < b ? a : (b < a ? b : ((isnan(a) || isnan(b)) ? NAN : b));
        case FH_MAX_RR: return a > b ? a : (b > a ? b : ((isnan(a) || isnan(b)) ? NAN : b));
        case FH_AND_RR: return a == 0.0f ? a : b;
        case FH_OR_RR: return a != 0.0f ? a : b;
    }
    return NAN;
}

// ---- graph -------------------------------------------------------------------
typedef uint32_t NodeId;
static const NodeId NO_NODE = 0xFFFFFFFFu;

struct GNode {
    uint8_t kind;   // 0 var, 1 const, 2 unary, 3 binary
    uint8_t op;     // FhOp (unary opcode, or the _RR opcode for binaries)
    uint8_t vkind;  // var: 0 X, 1 Y, 2 Z, 3 V
    float c;
    NodeId a, b;
    uint64_t vindex;
};

struct Graph {
    std::vector<GNode> nodes;
    struct Key {
        uint64_t k0, k1;
        bool operator==(const Key& o) const { return k0 == o.k0 && k1 == o.k1; }
    };
    struct KeyHash {
        size_t operator()(const Key& k) const { return (size_t)(k.k0 * 0x9E3779B97F4A7C15ull ^ (k.k1 + (k.k0 >> 29))); }
    };
    std::unordered_map<Key, NodeId, KeyHash> dedup;

    NodeId intern(const GNode& n) {
        Key k;
        switch (n.kind) {
            case 0: k = {(uint64_t)n.vkind << 8, n.vindex}; break;
            case 1: {
                // OrderedFloat identity: every NaN is one constant, -0 == +0
                uint32_t u = isnan(n.c) ? 0x7fc00000u : (n.c == 0.0f ? 0u : bits_of(n.c));
                k = {1, u};
                break;
            }
            case 2: k = {2 | ((uint64_t)n.op << 8), n.a}; break;
            default: k = {3 | ((uint64_t)n.op << 8), ((uint64_t)n.a << 32) | n.b}; break;
        }
        auto it = dedup.find(k);
        if (it != dedup.end()) return it->second;
        NodeId id = (NodeId)nodes.size();
        nodes.push_back(n);
        dedup.emplace(k, id);
        return id;
    }
    bool valid(NodeId n) const { return n < nodes.size(); }
    bool const_eq(NodeId n, float v) const { return valid(n) && nodes[n].kind == 1 && nodes[n].c == v; }

    NodeId var(uint8_t vkind, uint64_t index) { GNode n{}; n.kind = 0; n.vkind = vkind; n.vindex = index; return intern(n); }
    NodeId constant(float f) { GNode n{}; n.kind = 1; n.c = f; return intern(n); }

    NodeId raw_unary(int op, NodeId a) {
        if (!valid(a)) return NO_NODE;
        if (nodes[a].kind == 1) return constant(fold_unary(op, nodes[a].c));
        GNode n{}; n.kind = 2; n.op = (uint8_t)op; n.a = a;
        return intern(n);
    }
    NodeId raw_binary(int op, NodeId a, NodeId b) {
        if (!valid(a) || !valid(b)) return NO_NODE;
        if (nodes[a].kind == 1 && nodes[b].kind == 1) return constant(fold_binary(op, nodes[a].c, nodes[b].c));
        GNode n{}; n.kind = 3; n.op = (uint8_t)op; n.a = a; n.b = b;
        return intern(n);
    }
    NodeId sorted_binary(int op, NodeId a, NodeId b) { return a < b ? raw_binary(op, a, b) : raw_binary(op, b, a); }

    // Constructor identities.  `op` is a unary FhOp or the _RR flavour of a binary.
    NodeId unary(int op, NodeId a) { return raw_unary(op, a); }
    NodeId binary(int op, NodeId a, NodeId b) {
        if (!valid(a) || !valid(b)) return NO_NODE;
        switch (op) {
            case FH_ADD_RR:
                if (a == b) return binary(FH_MUL_RR, a, constant(2.0f));
                if (const_eq(a, 0.0f)) return b;
                if (const_eq(b, 0.0f)) return a;
                return sorted_binary(op, a, b);
            case FH_MUL_RR:
                if (a == b) return unary(FH_SQUARE, a);
                if (const_eq(a, 1.0f)) return b;
                if (const_eq(b, 1.0f)) return a;
                if (const_eq(a, 0.0f)) return a;
                if (const_eq(b, 0.0f)) return b;
                return sorted_binary(op, a, b);
            case FH_MIN_RR:
            case FH_MAX_RR:
                if (a == b) return a;
                return sorted_binary(op, a, b);
            case FH_SUB_RR:
                if (const_eq(a, 0.0f)) return unary(FH_NEG, b);
                if (const_eq(b, 0.0f)) return a;
                return raw_binary(op, a, b);
            case FH_DIV_RR:
                if (const_eq(a, 0.0f)) return a;
                if (const_eq(b, 1.0f)) return a;
                return raw_binary(op, a, b);
            case FH_AND_RR:
                if (nodes[a].kind == 1) return nodes[a].c == 0.0f ? a : b;
                return raw_binary(op, a, b);
            case FH_OR_RR:
                if (nodes[a].kind == 1) return nodes[a].c != 0.0f ? a : b;
                if (nodes[b].kind == 1 && nodes[b].c == 0.0f) return a;
                return raw_binary(op, a, b);
            default: return raw_binary(op, a, b);
        }
    }

    // `.vm` text (one "name opcode args..." per line; '#' comments).  Returns the
    // last node, or NO_NODE with `err` set.
    NodeId parse(const char* text, std::string& err) {
        static const std::map<std::string, int> un = {
            {"abs", FH_ABS}, {"neg", FH_NEG}, {"sqrt", FH_SQRT}, {"square", FH_SQUARE}, {"floor", FH_FLOOR},
            {"ceil", FH_CEIL}, {"round", FH_ROUND}, {"sin", FH_SIN}, {"cos", FH_COS}, {"tan", FH_TAN},
            {"asin", FH_ASIN}, {"acos", FH_ACOS}, {"atan", FH_ATAN}, {"ln", FH_LN}, {"not", FH_NOT},
            {"rand", FH_RAND}, {"exp", FH_EXP}};
        static const std::map<std::string, int> bin = {
            {"add", FH_ADD_RR}, {"mul", FH_MUL_RR}, {"min", FH_MIN_RR}, {"max", FH_MAX_RR}, {"div", FH_DIV_RR},
            {"atan2", FH_ATAN2_RR}, {"sub", FH_SUB_RR}, {"compare", FH_COMPARE_RR}, {"mod", FH_MOD_RR},
            {"and", FH_AND_RR}, {"or", FH_OR_RR}, {"mix", FH_MIX_RR}};
        std::unordered_map<std::string, NodeId> names;
        NodeId last = NO_NODE;
        std::istringstream in(text);
        std::string line, tok[4];
        while (std::getline(in, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty() || line[0] == '#') continue;
            std::istringstream ls(line);
            int nt = 0;
            while (nt < 4 && (ls >> tok[nt])) nt++;
            if (nt < 2) { err = "malformed line: " + line; return NO_NODE; }
            auto arg = [&](int i, NodeId* out) -> bool {
                if (i >= nt) { err = "missing argument: " + line; return false; }
                auto it = names.find(tok[i]);
                if (it == names.end()) { err = "unknown variable " + tok[i]; return false; }
                *out = it->second;
                return true;
            };
            const std::string& opc = tok[1];
            NodeId n = NO_NODE, a, b;
            if (opc == "const") {
                if (nt < 3) { err = "missing constant: " + line; return NO_NODE; }
                n = constant(strtof(tok[2].c_str(), nullptr));
            } else if (opc == "var-x") n = var(0, 0);
            else if (opc == "var-y") n = var(1, 0);
            else if (opc == "var-z") n = var(2, 0);
            else if (un.count(opc)) {
                if (!arg(2, &a)) return NO_NODE;
                n = unary(un.at(opc), a);
            } else if (bin.count(opc)) {
                if (!arg(2, &a) || !arg(3, &b)) return NO_NODE;
                n = binary(bin.at(opc), a, b);
            } else { err = "unknown opcode " + opc; return NO_NODE; }
            names[tok[0]] = n;
            last = n;
        }
        if (last == NO_NODE) err = "empty file";
        return last;
    }
};

// ---- SSA tape ------------------------------------------------------------------
struct SsaOp {
    uint8_t op;  // FhOp
    uint32_t out, a, b;
    uint32_t imm;  // f32 bits or slot
};
struct VarTable {
    int axis[3] = {-1, -1, -1};
    std::vector<std::pair<uint64_t, int>> named;  // Var::V(index) -> slot
    int count = 0;
    int slot_of(uint8_t vkind, uint64_t index) const {
        if (vkind < 3) return axis[vkind];
        for (auto& p : named) if (p.first == index) return p.second;
        return -1;
    }
    void touch(uint8_t vkind, uint64_t index) {
        if (slot_of(vkind, index) >= 0) return;
        if (vkind < 3) axis[vkind] = count++;
        else named.push_back({index, count++});
    }
};
struct SsaProgram {
    std::vector<SsaOp> ops;  // root first (i.e. reverse evaluation order)
    uint32_t n_values = 0;
    uint32_t n_choices = 0;
    uint32_t n_outputs = 0;
    VarTable vars;
};

static inline int rr_to_ri(int op) { return op - FH_ADD_RR + FH_ADD_RI; }
static inline int rr_to_ir(int op) {
    switch (op) {
        case FH_SUB_RR: return FH_SUB_IR;
        case FH_DIV_RR: return FH_DIV_IR;
        case FH_ATAN2_RR: return FH_ATAN2_IR;
        case FH_COMPARE_RR: return FH_COMPARE_IR;
        case FH_MIX_RR: return FH_MIX_IR;
        case FH_MOD_RR: return FH_MOD_IR;
    }
    return -1;
}

// Two depth-first sweeps with an explicit stack; the second emits a node only
// once every parent has been emitted, which yields the reference's tape order.
static inline bool flatten(const Graph& g, const std::vector<NodeId>& roots, SsaProgram& out, std::string& err) {
    const size_t n = g.nodes.size();
    std::vector<int64_t> value(n, -1);  // SSA value id for non-constants
    std::vector<uint32_t> pending(n, 0);
    std::vector<char> visited(n, 0);
    std::vector<NodeId> stack(roots);
    uint32_t next_value = 0;
    while (!stack.empty()) {
        NodeId id = stack.back();
        stack.pop_back();
        if (!g.valid(id)) { err = "bad node"; return false; }
        if (visited[id]) continue;
        visited[id] = 1;
        const GNode& nd = g.nodes[id];
        if (nd.kind != 1) {
            if (nd.kind == 0) out.vars.touch(nd.vkind, nd.vindex);
            value[id] = next_value++;
        }
        if (nd.kind >= 2) { pending[nd.a]++; stack.push_back(nd.a); }
        if (nd.kind == 3) { pending[nd.b]++; stack.push_back(nd.b); }
    }
    for (size_t i = 0; i < roots.size(); i++) {
        const GNode& r = g.nodes[roots[i]];
        if (r.kind == 1) {
            uint32_t v = next_value++;
            out.ops.push_back({FH_OUTPUT, 0, v, 0, (uint32_t)i});
            out.ops.push_back({FH_COPY_IMM, v, 0, 0, bits_of(r.c)});
        } else {
            out.ops.push_back({FH_OUTPUT, 0, (uint32_t)value[roots[i]], 0, (uint32_t)i});
        }
    }
    std::fill(visited.begin(), visited.end(), 0);
    stack.assign(roots.begin(), roots.end());
    while (!stack.empty()) {
        NodeId id = stack.back();
        stack.pop_back();
        if (pending[id] > 0 || visited[id]) continue;
        visited[id] = 1;
        const GNode& nd = g.nodes[id];
        if (nd.kind >= 2) { stack.push_back(nd.a); pending[nd.a]--; }
        if (nd.kind == 3) { stack.push_back(nd.b); pending[nd.b]--; }
        if (nd.kind == 1) continue;
        uint32_t me = (uint32_t)value[id];
        if (nd.kind == 0) {
            out.ops.push_back({FH_INPUT, me, 0, 0, (uint32_t)out.vars.slot_of(nd.vkind, nd.vindex)});
        } else if (nd.kind == 2) {
            if (g.nodes[nd.a].kind == 1) { err = "unary of constant survived folding"; return false; }
            out.ops.push_back({nd.op, me, (uint32_t)value[nd.a], 0, 0});
        } else {
            const GNode& l = g.nodes[nd.a];
            const GNode& r = g.nodes[nd.b];
            if (fh_is_choice(nd.op)) out.n_choices++;
            if (l.kind != 1 && r.kind != 1) {
                out.ops.push_back({nd.op, me, (uint32_t)value[nd.a], (uint32_t)value[nd.b], 0});
            } else if (r.kind == 1 && l.kind != 1) {
                out.ops.push_back({(uint8_t)rr_to_ri(nd.op), me, (uint32_t)value[nd.a], 0, bits_of(r.c)});
            } else if (l.kind == 1 && r.kind != 1) {
                int ir = rr_to_ir(nd.op);
                if (ir < 0) {
                    // add/mul/min/max commute into the (reg, imm) form; and/or never reach here
                    if (nd.op == FH_AND_RR || nd.op == FH_OR_RR) { err = "and/or(imm, reg) must be collapsed"; return false; }
                    out.ops.push_back({(uint8_t)rr_to_ri(nd.op), me, (uint32_t)value[nd.b], 0, bits_of(l.c)});
                } else {
                    out.ops.push_back({(uint8_t)ir, me, (uint32_t)value[nd.b], 0, bits_of(l.c)});
                }
            } else { err = "binary of two constants survived folding"; return false; }
        }
    }
    out.n_values = next_value;
    out.n_outputs = (uint32_t)roots.size();
    return true;
}

// ---- SSA -> device tape --------------------------------------------------------
struct HostTape {
    std::vector<uint64_t> ops;  // evaluation order
    uint32_t n_regs = 0, n_choices = 0, n_outputs = 0, n_vars = 0;
    VarTable vars;
    mutable uint32_t op_class = 0;   // what kinds of opcodes the tape holds, looked up once (capi_tapes.hpp tape_class; 0: not yet)
};

// Links of a register-allocated tape for the linked prune (prune2.hip): the tape's dependency graph with the register numbers
// taken out.  Per op 8 bytes: word 0 = opcode | class << 8 | choice ordinal << 16 (choice ops before this one), word 1 = fa | fb << 16
// - the op that produced operand a / b (the last writer of that register before this op, looking through register copies):
// its index, or 0x8000 | its choice ordinal when it is a min / max / and / or, 0xFFFF when there is no such operand.  Per
// choice 8 bytes: word 0 = the op's fa | fb << 16 again, word 1 = the index of the op | its class << 16 (a batch of ordinals: one load).  Returns false when the tape is outside what the linked prune handles (more than
// 16383 ops or 32767 choices - the field widths -, an OUTPUT that is not the one last op, an operand nobody wrote).
static inline bool compute_links(const HostTape& t, std::vector<uint64_t>& out, std::vector<uint64_t>& choice_ops) {
    const size_t n = t.ops.size();
    out.clear();
    choice_ops.clear();
    if (n == 0 || n > 0x3FFFu || t.n_choices > 0x7FFFu || FH_W_OP((uint32_t)t.ops[n - 1]) != FH_OUTPUT) return false;
    std::vector<int> last(FH_MAX_REGS, -1);
    std::vector<uint32_t> field(n, 0xFFFFu);     // what a consumer's link says when op i produced its operand
    out.reserve(n);
    for (size_t i = 0; i < n; i++) {
        const uint32_t w0 = (uint32_t)t.ops[i], w1 = (uint32_t)(t.ops[i] >> 32);
        const int op = (int)FH_W_OP(w0);
        const uint32_t ro = FH_W_OUT(w0), ra = FH_W_A(w0);
        const bool rr = fh_is_rr(op), choice = fh_is_choice(op);
        uint32_t kind = 2, fa = 0xFFFFu, fb = 0xFFFFu;     // (classes: prune2.hip FH_LK_*)
        if (op == FH_OUTPUT) { if (i + 1 != n) return false; kind = 0; }
        else if (op == FH_INPUT || op == FH_COPY_IMM) kind = 1;
        else if (op == FH_COPY_REG) kind = 4;
        else if (choice) kind = rr ? 5 : 6;
        else if (rr) kind = 3;
        if (kind != 1) { if (ra >= FH_MAX_REGS || last[ra] < 0) return false; fa = field[last[ra]]; }
        if (rr) { if (w1 >= FH_MAX_REGS || last[w1] < 0) return false; fb = field[last[w1]]; }
        const uint32_t ci = (uint32_t)choice_ops.size();
        out.push_back((uint64_t)((uint32_t)op | (kind << 8) | (ci << 16)) | ((uint64_t)(fa | (fb << 16)) << 32));
        if (op == FH_COPY_REG) field[i] = fa;                         // a copy is its source
        else if (choice) { field[i] = 0x8000u | ci; choice_ops.push_back((uint64_t)(fa | (fb << 16)) | ((uint64_t)((uint32_t)i | (kind << 16)) << 32)); }
        else field[i] = (uint32_t)i;
        if (op != FH_OUTPUT) last[ro] = (int)i;
    }
    return true;
}

struct RegPool {  // lowest-free-first over FH_MAX_REGS registers
    uint64_t free_[FH_MAX_REGS / 64];
    int high = 0;
    RegPool() { for (auto& w : free_) w = ~0ull; }
    int take() {
        for (uint32_t w = 0; w < FH_MAX_REGS / 64; w++)
            if (free_[w]) {
                int b = __builtin_ctzll(free_[w]);
                free_[w] &= free_[w] - 1;
                int r = (int)w * 64 + b;
                if (r + 1 > high) high = r + 1;
                return r;
            }
        return -1;
    }
    void give(int r) { free_[r >> 6] |= 1ull << (r & 63); }
};

// One reverse sweep (the SSA program is already root-first): a value gets a
// register at its last use and returns it at its definition, so the output
// register of an op is immediately reusable by that op's own operands.
static inline bool allocate(const SsaProgram& p, HostTape& t, std::string& err) {
    std::vector<int> reg(p.n_values, -1);
    RegPool pool;
    std::vector<uint64_t> rev;
    rev.reserve(p.ops.size());
    auto use = [&](uint32_t v) -> int {
        if (reg[v] < 0) reg[v] = pool.take();
        return reg[v];
    };
    for (const SsaOp& o : p.ops) {
        if (o.op == FH_OUTPUT) {
            int ra = use(o.a);
            if (ra < 0) { err = "more than 4096 live registers"; return false; }
            rev.push_back(fh_pack(o.op, 0, ra, 0, o.imm));
            continue;
        }
        int ro = reg[o.out];
        if (ro < 0) continue;  // dead value (cannot happen for graph-built programs)
        reg[o.out] = -1;
        pool.give(ro);
        int ra = 0, rb = 0;
        if (o.op != FH_INPUT && o.op != FH_COPY_IMM) ra = use(o.a);
        if (fh_is_rr(o.op)) rb = use(o.b);
        if (ra < 0 || rb < 0) { err = "more than 4096 live registers"; return false; }
        rev.push_back(fh_pack(o.op, ro, ra, rb, o.imm));
    }
    t.ops.assign(rev.rbegin(), rev.rend());
    t.n_regs = pool.high;
    t.n_choices = p.n_choices;
    t.n_outputs = p.n_outputs;
    t.n_vars = p.vars.count;
    t.vars = p.vars;
    return true;
}

// ---- reference bytecode -> SSA ------------------------------------------------
// fidget-bytecode/src/lib.rs:11-42, 203-332.  Registers (<= 254) and memory
// slots are value-numbered; Mem (load / store) becomes an alias, not an op.
static inline bool from_bytecode(const uint32_t* w, size_t n_words, SsaProgram& p, std::string& err) {
    static const int UN[] = {FH_NEG, FH_ABS, FH_RECIP, FH_SQRT, FH_SQUARE, FH_FLOOR, FH_CEIL, FH_ROUND, FH_NOT, FH_RAND,
                             FH_SIN, FH_COS, FH_TAN, FH_ASIN, FH_ACOS, FH_ATAN, FH_EXP, FH_LN};  // bytecode ops 3..20
    static const int BIN[] = {FH_ADD_RR, FH_SUB_RR, FH_MUL_RR, FH_DIV_RR, FH_ATAN2_RR, FH_COMPARE_RR, FH_MIX_RR,
                              FH_MOD_RR, FH_MIN_RR, FH_MAX_RR, FH_AND_RR, FH_OR_RR};  // bytecode ops 21..32
    if (n_words < 4 || (n_words & 1) || w[0] != 0xFFFFFFFFu || w[1] != 0) { err = "missing start marker"; return false; }
    std::vector<int64_t> cur(256, -1);
    std::map<uint32_t, int64_t> mem;
    std::vector<SsaOp> fwd;
    uint32_t next = 0, max_in = 0, n_out = 0;
    bool ended = false;
    for (size_t i = 2; i + 1 < n_words; i += 2) {
        uint32_t w0 = w[i], imm = w[i + 1];
        uint32_t op = w0 & 0xFF, r1 = (w0 >> 8) & 0xFF, r2 = (w0 >> 16) & 0xFF, r3 = (w0 >> 24) & 0xFF;
        if (w0 == 0xFFFFFFFFu) {
            if (imm == 0xFFFFFFFFu) { ended = true; break; }
            err = "user-defined jump ops are not supported";
            return false;
        }
        auto rd = [&](uint32_t r, uint32_t* v) -> bool {
            if (r == 0xFF || cur[r] < 0) { err = "read of undefined register"; return false; }
            *v = (uint32_t)cur[r];
            return true;
        };
        uint32_t a, b;
        if (op == 0) {  // Output
            if (!rd(r1, &a)) return false;
            if (imm >= 256) { err = "output slot out of range"; return false; }   // (slot + 1 sizes the output buffers)
            fwd.push_back({FH_OUTPUT, 0, a, 0, imm});
            if (imm + 1 > n_out) n_out = imm + 1;
        } else if (op == 1) {  // Input
            if (imm >= 16) { err = "input slot out of range (16 variables at most)"; return false; }   // FH_MAX_INPUTS
            cur[r1] = next;
            fwd.push_back({FH_INPUT, next++, 0, 0, imm});
            if (imm + 1 > max_in) max_in = imm + 1;
        } else if (op == 2) {  // Copy
            if (r2 == 0xFF) { cur[r1] = next; fwd.push_back({FH_COPY_IMM, next++, 0, 0, imm}); }
            else { if (!rd(r2, &a)) return false; cur[r1] = next; fwd.push_back({FH_COPY_REG, next++, a, 0, 0}); }
        } else if (op >= 3 && op <= 20) {
            if (!rd(r2, &a)) return false;
            cur[r1] = next;
            fwd.push_back({(uint8_t)UN[op - 3], next++, a, 0, 0});
        } else if (op >= 21 && op <= 32) {
            int rr = BIN[op - 21];
            if (r2 != 0xFF && r3 != 0xFF) {
                if (!rd(r2, &a) || !rd(r3, &b)) return false;
                cur[r1] = next;
                fwd.push_back({(uint8_t)rr, next++, a, b, 0});
            } else if (r3 == 0xFF && r2 != 0xFF) {
                if (!rd(r2, &a)) return false;
                cur[r1] = next;
                fwd.push_back({(uint8_t)rr_to_ri(rr), next++, a, 0, imm});
            } else if (r2 == 0xFF && r3 != 0xFF) {
                if (!rd(r3, &a)) return false;
                int ir = rr_to_ir(rr);
                if (ir < 0) { err = "imm,reg form of a commutative op"; return false; }
                cur[r1] = next;
                fwd.push_back({(uint8_t)ir, next++, a, 0, imm});
            } else { err = "binary op with two immediates"; return false; }
            if (fh_is_choice(rr)) p.n_choices++;
        } else if (op == 33) {  // Mem
            if (r2 == 0xFF && r1 != 0xFF) {  // load: reg <- mem
                auto it = mem.find(imm);
                if (it == mem.end()) { err = "load of undefined memory slot"; return false; }
                cur[r1] = it->second;
            } else if (r1 == 0xFF && r2 != 0xFF) {  // store: mem <- reg
                if (!rd(r2, &a)) return false;
                mem[imm] = a;
            } else { err = "malformed Mem op"; return false; }
        } else { err = "unknown bytecode op"; return false; }
    }
    if (!ended) { err = "missing end marker"; return false; }
    p.ops.assign(fwd.rbegin(), fwd.rend());
    p.n_values = next;
    p.n_outputs = n_out;
    p.vars.count = (int)max_in;  // slots only; the caller owns the Var -> slot map
    return true;
}

// Reverse liveness so that values made dead by Mem aliasing (or never used)
// do not reach `allocate` with a register.
static inline void drop_dead(SsaProgram& p) {
    std::vector<char> live(p.n_values, 0);
    std::vector<SsaOp> kept;
    uint32_t choices = 0;
    for (const SsaOp& o : p.ops) {
        if (o.op == FH_OUTPUT) { live[o.a] = 1; kept.push_back(o); continue; }
        if (!live[o.out]) continue;
        if (o.op != FH_INPUT && o.op != FH_COPY_IMM) live[o.a] = 1;
        if (fh_is_rr(o.op)) live[o.b] = 1;
        if (fh_is_choice(o.op)) choices++;
        kept.push_back(o);
    }
    // NB: dead choice ops would shift trace indices; the reference never emits them.
    p.ops.swap(kept);
    p.n_choices = choices;
}

// ---- tape parallelism: split a program at its root min / max tree ------------------------------
// Many shapes are a union (min) or intersection (max) of hundreds of small parts (prospero.vm:
// a min of 665 terms of <= 80 ops).  Evaluated as one tape that is a single dependency chain;
// split into `want` contiguous runs of terms it is `want` independent tapes whose outputs
// combine with the same op, in the same order - which keeps every tie (f32 min returns its
// second operand on a tie, types/float.rs:93-108) and every NaN exactly as in the original tree.
// Shared subexpressions are recomputed by each group that needs them.
// Returns the combining opcode (FH_MIN_RR / FH_MAX_RR) or -1 when the root does not split.
static inline int split_root(const SsaProgram& p, uint32_t want, uint32_t min_terms, std::vector<SsaProgram>& groups) {
    groups.clear();
    if (p.n_outputs != 1 || p.ops.empty() || p.ops[0].op != FH_OUTPUT || want < 2) return -1;
    std::vector<int> def(p.n_values, -1);
    for (size_t i = 0; i < p.ops.size(); i++)
        if (p.ops[i].op != FH_OUTPUT) def[p.ops[i].out] = (int)i;
    const uint32_t root = p.ops[0].a;
    if (def[root] < 0) return -1;
    const int rop = p.ops[def[root]].op;
    int rr, ri;
    if (rop == FH_MIN_RR || rop == FH_MIN_RI) { rr = FH_MIN_RR; ri = FH_MIN_RI; }
    else if (rop == FH_MAX_RR || rop == FH_MAX_RI) { rr = FH_MAX_RR; ri = FH_MAX_RI; }
    else return -1;
    // in-order terms of the maximal tree of that op hanging off the root
    struct Term { bool is_const; uint32_t v; };  // value id, or immediate bits
    std::vector<Term> terms;
    std::vector<std::pair<uint32_t, int>> stack{{root, 0}};  // (value, state); consts pushed as state 2
    while (!stack.empty()) {
        auto [v, st] = stack.back();
        stack.pop_back();
        if (st == 2) { terms.push_back({true, v}); continue; }
        const SsaOp& o = p.ops[def[v]];
        if (o.op == rr) { stack.push_back({o.b, 0}); stack.push_back({o.a, 0}); }        // a first, then b
        else if (o.op == ri) { stack.push_back({o.imm, 2}); stack.push_back({o.a, 0}); }  // a first, then the constant
        else terms.push_back({false, v});
    }
    if (terms.size() < min_terms) return -1;
    // weight of a term = ops it reaches (what its group will have to evaluate, roughly)
    std::vector<uint32_t> weight(terms.size(), 1);
    std::vector<uint32_t> mark(p.n_values, 0), work;
    uint64_t total = 0;
    for (size_t t = 0; t < terms.size(); t++) {
        if (terms[t].is_const) { total += 1; continue; }
        uint32_t n = 0;
        work.assign(1, terms[t].v);
        while (!work.empty()) {
            const uint32_t v = work.back();
            work.pop_back();
            if (mark[v] == t + 1) continue;
            mark[v] = (uint32_t)t + 1;
            n++;
            const SsaOp& o = p.ops[def[v]];
            if (o.op != FH_INPUT && o.op != FH_COPY_IMM) work.push_back(o.a);
            if (fh_is_rr(o.op)) work.push_back(o.b);
        }
        weight[t] = n;
        total += n;
    }
    // contiguous runs of about total / want ops
    const uint64_t per = (total + want - 1) / want;
    std::vector<std::pair<size_t, size_t>> runs;  // [first, last)
    size_t first = 0;
    uint64_t acc = 0;
    for (size_t t = 0; t < terms.size(); t++) {
        acc += weight[t];
        if (acc >= per && runs.size() + 1 < want) { runs.push_back({first, t + 1}); first = t + 1; acc = 0; }
    }
    if (first < terms.size()) runs.push_back({first, terms.size()});
    if (runs.size() < 2) return -1;
    for (auto& run : runs) {
        SsaProgram g;
        g.n_values = p.n_values;
        g.n_outputs = 1;
        g.vars = p.vars;
        // the chain acc = op(op(op(t0, t1), t2), ...) in evaluation order, stored root first
        std::vector<SsaOp> chain;
        uint32_t accv = 0;
        bool have = false;
        for (size_t t = run.first; t < run.second; t++) {
            const Term& tm = terms[t];
            if (!have) {
                if (tm.is_const) { accv = g.n_values++; chain.push_back(SsaOp{FH_COPY_IMM, accv, 0, 0, tm.v}); }
                else accv = tm.v;
                have = true;
                continue;
            }
            const uint32_t nv = g.n_values++;
            if (tm.is_const) chain.push_back(SsaOp{(uint8_t)ri, nv, accv, 0, tm.v});
            else chain.push_back(SsaOp{(uint8_t)rr, nv, accv, tm.v, 0});
            accv = nv;
        }
        g.ops.push_back(SsaOp{FH_OUTPUT, 0, accv, 0, 0});
        for (size_t i = chain.size(); i-- > 0;) g.ops.push_back(chain[i]);
        for (size_t i = 1; i < p.ops.size(); i++) g.ops.push_back(p.ops[i]);  // what is not reached is dropped below
        drop_dead(g);
        groups.push_back(std::move(g));
    }
    return rr;
}

// ---- tape parallelism for the renderer: terms of the root tree as separate outputs ------------------
// split_root's groups recompute shared subexpressions and their pruned tapes cannot simply be glued
// back together (the duplicates would stay).  For the renderer the split is therefore only used to
// EVALUATE: the groups hold the tree's terms as outputs (no combining chain), the tree itself - a
// few hundred min / max ops - becomes a small "top" program over those outputs, and every choice of
// the full tape is found either in a group's trace or in the top program's (`choice_src`).  The
// choices of the full tape are then exactly those its own forward pass would have recorded, and the
// prune works on the full tape as if nothing had been split.
struct TopOp {
    uint8_t op;              // FH_MIN_RR / FH_MAX_RR (b_kind 0, 1) or their reg,imm forms (b_kind 2)
    uint8_t out;             // top register
    uint8_t a_kind, b_kind;  // 0 top register, 1 term (group output), 2 immediate bits
    uint32_t a, b;
};
struct TermPlan {
    int op = -1;
    uint32_t n_terms = 0, top_regs = 0;
    bool chain = false;                // acc = op(acc, term or constant) all the way, starting from a term
    std::vector<SsaProgram> groups;    // OUTPUT imm = term index
    std::vector<TopOp> top;            // evaluation order; the last op writes the root
    std::vector<uint32_t> choice_src;  // per choice of the full tape, tape order: group << 24 | index; group 255: top op
};
static inline bool plan_terms(const SsaProgram& p, uint32_t want, uint32_t min_terms, uint32_t max_top_regs, TermPlan& plan) {
    plan = TermPlan();
    if (p.n_outputs != 1 || p.ops.empty() || p.ops[0].op != FH_OUTPUT || want < 2) return false;
    std::vector<int> def(p.n_values, -1);
    for (size_t i = 1; i < p.ops.size(); i++) {
        if (p.ops[i].op == FH_OUTPUT) return false;
        def[p.ops[i].out] = (int)i;
    }
    const uint32_t root = p.ops[0].a;
    if (def[root] < 0) return false;
    const int rop = p.ops[def[root]].op;
    int rr, ri;
    if (rop == FH_MIN_RR || rop == FH_MIN_RI) { rr = FH_MIN_RR; ri = FH_MIN_RI; }
    else if (rop == FH_MAX_RR || rop == FH_MAX_RI) { rr = FH_MAX_RR; ri = FH_MAX_RI; }
    else return false;
    // tree nodes and, in order of first appearance, the values hanging off the tree
    std::vector<char> is_node(p.n_values, 0);
    std::vector<int> term_of(p.n_values, -1);
    std::vector<uint32_t> terms;
    std::vector<uint32_t> stack{root};
    while (!stack.empty()) {
        const uint32_t v = stack.back();
        stack.pop_back();
        const SsaOp& o = p.ops[def[v]];
        if (o.op == rr || o.op == ri) {
            if (is_node[v]) continue;
            is_node[v] = 1;
            if (o.op == rr) stack.push_back(o.b);
            stack.push_back(o.a);
        } else if (term_of[v] < 0) {
            term_of[v] = (int)terms.size();
            terms.push_back(v);
        }
    }
    if (terms.size() < min_terms || terms.size() >= (1u << 20)) return false;
    // a tree node may also be used from inside a term; then it is a term as well (its value must be
    // an output of some group), evaluated a second time by the top program - same inputs, same result
    // the top program: the tree's ops in evaluation order (p.ops is root first)
    std::vector<int> top_index(p.n_values, -1), last_use(p.n_values, -1);
    std::vector<uint32_t> order;
    for (size_t i = p.ops.size(); i-- > 1;)
        if (is_node[p.ops[i].out]) { top_index[p.ops[i].out] = (int)order.size(); order.push_back((uint32_t)i); }
    for (size_t k = 0; k < order.size(); k++) {
        const SsaOp& o = p.ops[order[k]];
        if (is_node[o.a]) last_use[o.a] = (int)k;
        if (o.op == rr && is_node[o.b]) last_use[o.b] = (int)k;
    }
    std::vector<int> treg(p.n_values, -1);
    std::vector<int> free_regs;
    uint32_t high = 0;
    for (size_t k = 0; k < order.size(); k++) {
        const SsaOp& o = p.ops[order[k]];
        TopOp t{};
        t.op = o.op;
        auto operand = [&](uint32_t v, uint8_t& kind, uint32_t& ref) -> bool {
            if (is_node[v]) { if (treg[v] < 0) return false; kind = 0; ref = (uint32_t)treg[v]; }
            else { kind = 1; ref = (uint32_t)term_of[v]; }
            return true;
        };
        if (!operand(o.a, t.a_kind, t.a)) return false;
        if (o.op == rr) { if (!operand(o.b, t.b_kind, t.b)) return false; }
        else { t.b_kind = 2; t.b = o.imm; }
        // operands die here (a value used twice by the same op is released once)
        if (is_node[o.a] && last_use[o.a] == (int)k) { free_regs.push_back(treg[o.a]); }
        if (o.op == rr && is_node[o.b] && o.b != o.a && last_use[o.b] == (int)k) { free_regs.push_back(treg[o.b]); }
        int r;
        if (!free_regs.empty()) { r = free_regs.back(); free_regs.pop_back(); }
        else r = (int)high++;
        if (high > max_top_regs || r > 255) return false;
        treg[o.out] = r;
        t.out = (uint8_t)r;
        plan.top.push_back(t);
    }
    if (plan.top.empty() || p.ops[order.back()].out != root) return false;
    plan.top_regs = high;
    plan.chain = plan.top[0].a_kind == 1 && plan.top[0].b_kind != 0;
    for (size_t k = 1; k < plan.top.size() && plan.chain; k++) plan.chain = plan.top[k].a_kind == 0 && plan.top[k].b_kind != 0;
    plan.n_terms = (uint32_t)terms.size();
    plan.op = rr;
    // groups: contiguous runs of terms of about equal weight (ops a term reaches)
    std::vector<uint32_t> weight(terms.size(), 1), mark(p.n_values, 0), work;
    uint64_t total = 0;
    for (size_t t = 0; t < terms.size(); t++) {
        uint32_t n = 0;
        work.assign(1, terms[t]);
        while (!work.empty()) {
            const uint32_t v = work.back();
            work.pop_back();
            if (mark[v] == t + 1) continue;
            mark[v] = (uint32_t)t + 1;
            n++;
            const SsaOp& o = p.ops[def[v]];
            if (o.op != FH_INPUT && o.op != FH_COPY_IMM) work.push_back(o.a);
            if (fh_is_rr(o.op)) work.push_back(o.b);
        }
        weight[t] = n;
        total += n;
    }
    const uint64_t per = (total + want - 1) / want;
    std::vector<std::pair<size_t, size_t>> runs;
    size_t first = 0;
    uint64_t acc = 0;
    for (size_t t = 0; t < terms.size(); t++) {
        acc += weight[t];
        if (acc >= per && runs.size() + 1 < want) { runs.push_back({first, t + 1}); first = t + 1; acc = 0; }
    }
    if (first < terms.size()) runs.push_back({first, terms.size()});
    if (runs.size() < 2) return false;
    // per group: each term's OUTPUT right after (in evaluation order) the op that defines it, so that
    // a term's register is free again as soon as it is stored
    std::vector<std::vector<int>> local_choice(runs.size());
    for (size_t g = 0; g < runs.size(); g++) {
        SsaProgram gp;
        gp.n_values = p.n_values;
        gp.n_outputs = plan.n_terms;
        gp.vars = p.vars;
        std::vector<int> out_term(p.n_values, -1);
        for (size_t t = runs[g].first; t < runs[g].second; t++) out_term[terms[t]] = (int)t;
        for (size_t i = 1; i < p.ops.size(); i++) {
            const SsaOp& o = p.ops[i];
            if (out_term[o.out] >= 0) gp.ops.push_back(SsaOp{FH_OUTPUT, 0, o.out, 0, (uint32_t)out_term[o.out]});
            gp.ops.push_back(o);
        }
        drop_dead(gp);
        // choice index, in this group's tape order, of every choice op it holds
        local_choice[g].assign(p.n_values, -1);
        int rank = 0;
        for (size_t i = gp.ops.size(); i-- > 0;)
            if (gp.ops[i].op != FH_OUTPUT && fh_is_choice(gp.ops[i].op)) local_choice[g][gp.ops[i].out] = rank++;
        plan.groups.push_back(std::move(gp));
    }
    // where each choice of the full tape is recorded
    std::vector<char> live(p.n_values, 0);  // (same liveness rule as allocate / drop_dead)
    live[root] = 1;
    std::vector<char> live_op(p.ops.size(), 0);
    for (size_t i = 1; i < p.ops.size(); i++) {
        const SsaOp& o = p.ops[i];
        if (!live[o.out]) continue;
        live_op[i] = 1;
        if (o.op != FH_INPUT && o.op != FH_COPY_IMM) live[o.a] = 1;
        if (fh_is_rr(o.op)) live[o.b] = 1;
    }
    for (size_t i = p.ops.size(); i-- > 1;) {
        const SsaOp& o = p.ops[i];
        if (!live_op[i] || !fh_is_choice(o.op)) continue;
        uint32_t src = 0xFFFFFFFFu;
        if (is_node[o.out]) src = (255u << 24) | (uint32_t)top_index[o.out];
        else
            for (size_t g = 0; g < runs.size(); g++)
                if (local_choice[g][o.out] >= 0) { src = ((uint32_t)g << 24) | (uint32_t)local_choice[g][o.out]; break; }
        if (src == 0xFFFFFFFFu) return false;
        plan.choice_src.push_back(src);
    }
    return plan.choice_src.size() == p.n_choices;
}

}  // namespace fh
