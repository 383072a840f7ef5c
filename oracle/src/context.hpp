// ORACLE — TEST INFRASTRUCTURE ONLY (see types.hpp header).
//
// Restates the deduplicated math-graph arena and its constructor rules:
//   fidget-core/src/context/mod.rs      (Context: 49-51; ctor rules 188-780;
//                                        from_text 878-941)
//   fidget-core/src/context/op.rs       (Op, UnaryOpcode, BinaryOpcode, eval 50-94)
//   fidget-core/src/context/indexed.rs  (IndexMap insert = dedup, 58-64)
//   fidget-core/src/var/mod.rs          (Var)
#pragma once
#include <cstdio>
#include <cstdlib>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "types.hpp"

namespace orc {

enum UnaryOpcode : uint8_t {
    U_NEG, U_ABS, U_RECIP, U_SQRT, U_SQUARE, U_FLOOR, U_CEIL, U_ROUND, U_SIN, U_COS, U_TAN,
    U_ASIN, U_ACOS, U_ATAN, U_EXP, U_LN, U_NOT, U_RAND
};
enum BinaryOpcode : uint8_t {
    B_ADD, B_SUB, B_MUL, B_DIV, B_ATAN, B_MIN, B_MAX, B_COMPARE, B_MOD, B_AND, B_OR, B_MIX
};

// context/op.rs:50-94
static inline float eval_unary(UnaryOpcode op, float a) {
    switch (op) {
        case U_NEG: return -a;
        case U_ABS: return fabsf(a);
        case U_RECIP: return 1.0f / a;
        case U_SQRT: return sqrtf(a);
        case U_SQUARE: return a * a;
        case U_FLOOR: return floorf(a);
        case U_CEIL: return ceilf(a);
        case U_ROUND: return roundf(a);
        case U_SIN: return sinf(a);
        case U_COS: return cosf(a);
        case U_TAN: return tanf(a);
        case U_ASIN: return asinf(a);
        case U_ACOS: return acosf(a);
        case U_ATAN: return atanf(a);
        case U_EXP: return expf(a);
        case U_LN: return logf(a);
        case U_NOT: return f_not(a);
        case U_RAND: return f_rand(a);
    }
    return NANF;
}
static inline float eval_binary(BinaryOpcode op, float a, float b) {
    switch (op) {
        case B_ADD: return a + b;
        case B_SUB: return a - b;
        case B_MUL: return a * b;
        case B_DIV: return a / b;
        case B_ATAN: return atan2f(a, b);
        case B_MIN: return f_min_choice(a, b).v;
        case B_MAX: return f_max_choice(a, b).v;
        case B_COMPARE: return f_compare(a, b);
        case B_MOD: return rem_euclid(a, b);
        case B_AND: return f_and_choice(a, b).v;
        case B_OR: return f_or_choice(a, b).v;
        case B_MIX: return f_mix(a, b);
    }
    return NANF;
}

// var/mod.rs: Var::{X,Y,Z,V(u64)}
struct Var {
    uint8_t kind;  // 0=X 1=Y 2=Z 3=V
    uint64_t index;
    bool operator==(const Var& o) const { return kind == o.kind && index == o.index; }
};
static const Var VAR_X{0, 0}, VAR_Y{1, 0}, VAR_Z{2, 0};

enum NodeKind : uint8_t { N_INPUT, N_CONST, N_BINARY, N_UNARY };
struct NodeOp {
    NodeKind kind;
    uint8_t opcode;
    Var var;
    float c;
    uint32_t a, b;
};
typedef uint32_t Node;
static const Node BAD_NODE = 0xFFFFFFFFu;

// var/mod.rs:100-148 (VarMap: index assigned on first insert, tightly packed)
struct VarMap {
    int x = -1, y = -1, z = -1;
    std::map<uint64_t, int> v;
    std::vector<Var> order;  // insertion order == index order
    int len() const { return (int)order.size(); }
    void insert(Var var) {
        int next = len();
        switch (var.kind) {
            case 0: if (x < 0) { x = next; order.push_back(var); } break;
            case 1: if (y < 0) { y = next; order.push_back(var); } break;
            case 2: if (z < 0) { z = next; order.push_back(var); } break;
            default:
                if (!v.count(var.index)) { v[var.index] = next; order.push_back(var); }
        }
    }
    int get(Var var) const {
        switch (var.kind) {
            case 0: return x;
            case 1: return y;
            case 2: return z;
            default: {
                auto it = v.find(var.index);
                return it == v.end() ? -1 : it->second;
            }
        }
    }
};

struct Context {
    std::vector<NodeOp> ops;
    // dedup map: key is a canonical byte string of the op (indexed.rs:58-64)
    std::unordered_map<std::string, Node> map;

    static std::string key(const NodeOp& o) {
        char buf[32];
        std::memset(buf, 0, sizeof(buf));
        buf[0] = (char)o.kind;
        switch (o.kind) {
            case N_INPUT:
                buf[1] = (char)o.var.kind;
                std::memcpy(buf + 8, &o.var.index, 8);
                break;
            case N_CONST: {
                // ordered_float::OrderedFloat Eq/Hash: all NaN equal, -0 == +0
                uint32_t bits = f2u(o.c);
                if (std::isnan(o.c)) bits = 0x7fc00000u;
                else if (o.c == 0.0f) bits = 0;
                std::memcpy(buf + 8, &bits, 4);
                break;
            }
            case N_BINARY:
                buf[1] = (char)o.opcode;
                std::memcpy(buf + 8, &o.a, 4);
                std::memcpy(buf + 12, &o.b, 4);
                break;
            case N_UNARY:
                buf[1] = (char)o.opcode;
                std::memcpy(buf + 8, &o.a, 4);
                break;
        }
        return std::string(buf, 24);
    }
    Node insert(const NodeOp& o) {
        std::string k = key(o);
        auto it = map.find(k);
        if (it != map.end()) return it->second;
        Node n = (Node)ops.size();
        ops.push_back(o);
        map.emplace(std::move(k), n);
        return n;
    }
    size_t len() const { return ops.size(); }
    const NodeOp* get_op(Node n) const { return n < ops.size() ? &ops[n] : nullptr; }
    bool get_const(Node n, float* out) const {
        const NodeOp* o = get_op(n);
        if (o && o->kind == N_CONST) { *out = o->c; return true; }
        return false;
    }
    bool is_const(Node n, float v) const {  // pattern `Ok(v)` (float == compare)
        float c;
        return get_const(n, &c) && c == v;
    }

    Node var(Var v) { NodeOp o{}; o.kind = N_INPUT; o.var = v; return insert(o); }
    Node x() { return var(VAR_X); }
    Node y() { return var(VAR_Y); }
    Node z() { return var(VAR_Z); }
    Node constant(float f) { NodeOp o{}; o.kind = N_CONST; o.c = f; return insert(o); }

    // mod.rs:188-224
    Node op_unary(Node a, UnaryOpcode op) {
        const NodeOp* oa = get_op(a);
        if (!oa) return BAD_NODE;
        if (oa->kind == N_CONST) return constant(eval_unary(op, oa->c));
        NodeOp o{}; o.kind = N_UNARY; o.opcode = op; o.a = a;
        return insert(o);
    }
    Node op_binary(Node a, Node b, BinaryOpcode op) {
        const NodeOp* oa = get_op(a);
        const NodeOp* ob = get_op(b);
        if (!oa || !ob) return BAD_NODE;
        if (oa->kind == N_CONST && ob->kind == N_CONST) return constant(eval_binary(op, oa->c, ob->c));
        NodeOp o{}; o.kind = N_BINARY; o.opcode = op; o.a = a; o.b = b;
        return insert(o);
    }
    Node op_binary_commutative(Node a, Node b, BinaryOpcode op) {
        return op_binary(a < b ? a : b, a < b ? b : a, op);
    }
    // mod.rs:234-322
    Node add(Node a, Node b) {
        if (a == BAD_NODE || b == BAD_NODE) return BAD_NODE;
        if (a == b) return mul(a, constant(2.0f));
        if (is_const(a, 0.0f)) return b;
        if (is_const(b, 0.0f)) return a;
        return op_binary_commutative(a, b, B_ADD);
    }
    Node mul(Node a, Node b) {
        if (a == BAD_NODE || b == BAD_NODE) return BAD_NODE;
        if (a == b) return square(a);
        if (is_const(a, 1.0f)) return b;
        if (is_const(b, 1.0f)) return a;
        if (is_const(a, 0.0f)) return a;
        if (is_const(b, 0.0f)) return b;
        return op_binary_commutative(a, b, B_MUL);
    }
    Node min(Node a, Node b) {
        if (a == BAD_NODE || b == BAD_NODE) return BAD_NODE;
        if (a == b) return a;
        return op_binary_commutative(a, b, B_MIN);
    }
    Node max(Node a, Node b) {
        if (a == BAD_NODE || b == BAD_NODE) return BAD_NODE;
        if (a == b) return a;
        return op_binary_commutative(a, b, B_MAX);
    }
    // mod.rs:344-400
    Node and_(Node a, Node b) {
        const NodeOp* oa = get_op(a);
        if (!oa || !get_op(b)) return BAD_NODE;
        if (oa->kind == N_CONST) return (oa->c == 0.0f) ? a : b;
        return op_binary(a, b, B_AND);
    }
    Node or_(Node a, Node b) {
        const NodeOp* oa = get_op(a);
        const NodeOp* ob = get_op(b);
        if (!oa || !ob) return BAD_NODE;
        if (oa->kind == N_CONST) return (oa->c != 0.0f) ? a : b;
        if (ob->kind == N_CONST && ob->c == 0.0f) return a;
        return op_binary(a, b, B_OR);
    }
    Node not_(Node a) { return op_unary(a, U_NOT); }
    Node neg(Node a) { return op_unary(a, U_NEG); }
    Node recip(Node a) { return op_unary(a, U_RECIP); }
    Node abs(Node a) { return op_unary(a, U_ABS); }
    Node sqrt(Node a) { return op_unary(a, U_SQRT); }
    Node sin(Node a) { return op_unary(a, U_SIN); }
    Node cos(Node a) { return op_unary(a, U_COS); }
    Node tan(Node a) { return op_unary(a, U_TAN); }
    Node asin(Node a) { return op_unary(a, U_ASIN); }
    Node acos(Node a) { return op_unary(a, U_ACOS); }
    Node atan(Node a) { return op_unary(a, U_ATAN); }
    Node exp(Node a) { return op_unary(a, U_EXP); }
    Node ln(Node a) { return op_unary(a, U_LN); }
    Node square(Node a) { return op_unary(a, U_SQUARE); }
    Node floor(Node a) { return op_unary(a, U_FLOOR); }
    Node ceil(Node a) { return op_unary(a, U_CEIL); }
    Node round(Node a) { return op_unary(a, U_ROUND); }
    Node rand(Node a) { return op_unary(a, U_RAND); }
    // mod.rs:586-623
    Node sub(Node a, Node b) {
        if (a == BAD_NODE || b == BAD_NODE) return BAD_NODE;
        if (is_const(a, 0.0f)) return neg(b);
        if (is_const(b, 0.0f)) return a;
        return op_binary(a, b, B_SUB);
    }
    Node div(Node a, Node b) {
        if (a == BAD_NODE || b == BAD_NODE) return BAD_NODE;
        if (is_const(a, 0.0f)) return a;
        if (is_const(b, 1.0f)) return a;
        return op_binary(a, b, B_DIV);
    }
    Node atan2(Node y, Node x) { return op_binary(y, x, B_ATAN); }
    Node compare(Node a, Node b) { return op_binary(a, b, B_COMPARE); }
    Node mix(Node a, Node b) { return op_binary(a, b, B_MIX); }
    Node modulo(Node a, Node b) { return op_binary(a, b, B_MOD); }
    // mod.rs:701-736
    Node less_than(Node lhs, Node rhs) {
        Node cmp = op_binary(rhs, lhs, B_COMPARE);
        return max(cmp, constant(0.0f));
    }
    Node less_than_or_equal(Node lhs, Node rhs) {
        Node cmp = op_binary(rhs, lhs, B_COMPARE);
        Node shift = add(cmp, constant(1.0f));
        return min(shift, constant(1.0f));
    }
    // mod.rs:766-780
    Node if_nonzero_else(Node cond, Node a, Node b) {
        Node lhs = and_(cond, a);
        Node nc = not_(cond);
        Node rhs = and_(nc, b);
        return or_(lhs, rhs);
    }

    // mod.rs:878-941.  Returns the last node; throws on parse errors.
    static Node from_text(Context& ctx, const std::string& text) {
        std::map<std::string, Node> seen;
        Node last = BAD_NODE;
        std::istringstream in(text);
        std::string line;
        while (std::getline(in, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty() || line[0] == '#') continue;
            std::istringstream it(line);
            std::string name, opcode;
            if (!(it >> name >> opcode)) throw std::runtime_error("bad line: " + line);
            auto pop = [&]() -> Node {
                std::string t;
                if (!(it >> t)) throw std::runtime_error("missing arg: " + line);
                auto f = seen.find(t);
                if (f == seen.end()) throw std::runtime_error("unknown variable " + t);
                return f->second;
            };
            Node n;
            if (opcode == "const") {
                std::string t;
                it >> t;
                n = ctx.constant(strtof(t.c_str(), nullptr));
            } else if (opcode == "var-x") n = ctx.x();
            else if (opcode == "var-y") n = ctx.y();
            else if (opcode == "var-z") n = ctx.z();
            else if (opcode == "abs") n = ctx.abs(pop());
            else if (opcode == "neg") n = ctx.neg(pop());
            else if (opcode == "sqrt") n = ctx.sqrt(pop());
            else if (opcode == "square") n = ctx.square(pop());
            else if (opcode == "floor") n = ctx.floor(pop());
            else if (opcode == "ceil") n = ctx.ceil(pop());
            else if (opcode == "round") n = ctx.round(pop());
            else if (opcode == "sin") n = ctx.sin(pop());
            else if (opcode == "cos") n = ctx.cos(pop());
            else if (opcode == "tan") n = ctx.tan(pop());
            else if (opcode == "asin") n = ctx.asin(pop());
            else if (opcode == "acos") n = ctx.acos(pop());
            else if (opcode == "atan") n = ctx.atan(pop());
            else if (opcode == "ln") n = ctx.ln(pop());
            else if (opcode == "not") n = ctx.not_(pop());
            else if (opcode == "rand") n = ctx.rand(pop());
            else if (opcode == "exp") n = ctx.exp(pop());
            else {
                // two-argument forms: Rust evaluates `pop()?, pop()?` left-to-right
                Node a = pop();
                Node b = pop();
                if (opcode == "add") n = ctx.add(a, b);
                else if (opcode == "mul") n = ctx.mul(a, b);
                else if (opcode == "min") n = ctx.min(a, b);
                else if (opcode == "max") n = ctx.max(a, b);
                else if (opcode == "div") n = ctx.div(a, b);
                else if (opcode == "atan2") n = ctx.atan2(a, b);
                else if (opcode == "sub") n = ctx.sub(a, b);
                else if (opcode == "compare") n = ctx.compare(a, b);
                else if (opcode == "mod") n = ctx.modulo(a, b);
                else if (opcode == "and") n = ctx.and_(a, b);
                else if (opcode == "or") n = ctx.or_(a, b);
                else if (opcode == "mix") n = ctx.mix(a, b);
                else throw std::runtime_error("unknown opcode " + opcode);
            }
            seen[name] = n;
            last = n;
        }
        if (last == BAD_NODE) throw std::runtime_error("empty file");
        return last;
    }
};

}  // namespace orc
