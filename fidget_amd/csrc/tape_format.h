// fidget-hip device tape format (shared by host compiler and HIP kernels).
//
// A tape is a straight-line program in *evaluation order*, 8 bytes per
// operation, resident in HBM and fetched wave-uniformly:
//
//   word0 = opcode | out << 8 | a << 20               (12-bit register numbers)
//   word1 = b (reg,reg ops), or f32 immediate bits, or the input / output slot
//
// Unlike the reference's bytecode (fidget-bytecode/src/lib.rs:11-42), operand
// forms are folded into the opcode (no 0xFF "immediate" flag), there are no
// Load/Store ops (up to 256 registers, densely renumbered every time a tape is
// simplified) and no start/end sentinels (lengths travel with the tape
// descriptor).  The reference wire format is accepted at the C-ABI boundary
// and converted (fhip_tape_from_bytecode).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FH_HD __host__ __device__
#else
#define FH_HD
#endif

enum FhOp {
    FH_OUTPUT = 0,   // output[slot] = r[a]
    FH_INPUT = 1,    // r[out] = var[slot]
    FH_COPY_REG = 2, // r[out] = r[a]
    FH_COPY_IMM = 3, // r[out] = imm
    // unary, r[out] = f(r[a])
    FH_NEG = 4, FH_ABS, FH_RECIP, FH_SQRT, FH_SQUARE, FH_FLOOR, FH_CEIL, FH_ROUND, FH_SIN, FH_COS,
    FH_TAN, FH_ASIN, FH_ACOS, FH_ATAN, FH_EXP, FH_LN, FH_NOT, FH_RAND,  // .. 21
    // binary reg,reg: r[out] = f(r[a], r[b])
    FH_ADD_RR = 22, FH_SUB_RR, FH_MUL_RR, FH_DIV_RR, FH_ATAN2_RR, FH_COMPARE_RR, FH_MIX_RR, FH_MOD_RR,
    FH_MIN_RR, FH_MAX_RR, FH_AND_RR, FH_OR_RR,  // .. 33
    // binary reg,imm: r[out] = f(r[a], imm)
    FH_ADD_RI = 34, FH_SUB_RI, FH_MUL_RI, FH_DIV_RI, FH_ATAN2_RI, FH_COMPARE_RI, FH_MIX_RI, FH_MOD_RI,
    FH_MIN_RI, FH_MAX_RI, FH_AND_RI, FH_OR_RI,  // .. 45
    // binary imm,reg: r[out] = f(imm, r[a])   (non-commutative ops only)
    FH_SUB_IR = 46, FH_DIV_IR, FH_ATAN2_IR, FH_COMPARE_IR, FH_MIX_IR, FH_MOD_IR,  // .. 51
    FH_OP_COUNT = 52
};

#define FH_BIN_BASE(op) ((op) >= FH_SUB_IR ? -1 : ((op) >= FH_ADD_RI ? (op) - FH_ADD_RI : (op) - FH_ADD_RR))

FH_HD static inline int fh_is_unary(int op) { return op >= FH_NEG && op <= FH_RAND; }
FH_HD static inline int fh_is_rr(int op) { return op >= FH_ADD_RR && op <= FH_OR_RR; }
FH_HD static inline int fh_is_ri(int op) { return op >= FH_ADD_RI && op <= FH_OR_RI; }
FH_HD static inline int fh_is_ir(int op) { return op >= FH_SUB_IR && op <= FH_MOD_IR; }
// min / max / and / or in either register form: the ops that record a Choice
FH_HD static inline int fh_is_choice(int op) { return (op >= FH_MIN_RR && op <= FH_OR_RR) || (op >= FH_MIN_RI && op <= FH_OR_RI); }

// word0 = opcode[0:8] | out[8:20] | a[20:32]   (12-bit registers: up to 4096)
// word1 = b for reg,reg ops; otherwise the f32 immediate / input / output slot
FH_HD static inline uint64_t fh_pack(uint32_t op, uint32_t out, uint32_t a, uint32_t b, uint32_t imm) {
    const uint32_t w1 = fh_is_rr((int)op) ? b : imm;
    return (uint64_t)(op | (out << 8) | (a << 20)) | ((uint64_t)w1 << 32);
}
#define FH_W_OP(w0) ((w0) & 0xFFu)
#define FH_W_OUT(w0) (((w0) >> 8) & 0xFFFu)
#define FH_W_A(w0) ((w0) >> 20)
#define FH_MAX_REGS 4096u

// Choice values, identical to the reference (fidget-core/src/vm/choice.rs:15-29)
enum { FH_CHOICE_UNKNOWN = 0, FH_CHOICE_LEFT = 1, FH_CHOICE_RIGHT = 2, FH_CHOICE_BOTH = 3 };

// Descriptor of one tape inside the device arena
struct FhTapeRef {
    uint32_t off;        // first op, in 8-byte units from the arena base
    uint32_t len;        // number of ops
    uint16_t n_regs;     // registers used (max index + 1)
    uint16_t n_choices;  // choice ops in the tape
};
