// k_prune2 - the linked prune: VmData::simplify for ONE child tile per wavefront, visiting only the ops the child keeps.
//
// VmData::simplify (fidget-core/src/vm/data.rs:123-318, restated as prune_sweep in kernels.hip) walks the parent tape
// backwards and asks of every op whether its output register is wanted.  A child of the root tape keeps ~3 % of it (prospero:
// 150-210 of 6363 ops), so nearly all of that walk finds out that an op is dead or that a decided min / max merely passes its
// operand on - the scalar sweep fh_prune1 needs ~26-38 k instructions per child for it, 340 us for the root level, a third of
// the frame's longest chain.
//
// Here the tape comes with LINKS (host_graph.hpp compute_links, made once per tape): per op, which op produced each operand -
// in SSA terms, register numbers no longer matter - and for a producer that is itself a min / max / and / or, the ordinal
// of that choice instead.  Then
//   A  the child's choices turn every choice op into "kept", "is its left / right operand" or "is its immediate"; chains
//      of passed-on operands (prospero's root is a chain of 664 min ops of which a child keeps a handful) are followed for
//      ALL choice ops at once by pointer jumping, 64 ordinals per step in tape order: E[q] = the op whose value choice q's
//      output really is;
//   B  a walk over the kept ops only: a bit mask of wanted ops, the highest one visited next, its output register freed, its
//      operands' producers (through E) given registers and marked wanted - the reverse sweep of simplify restricted to live ops,
//      with the same lowest-free-first register pool, keyed by the producing op rather than by the parent's register number;
//   C  (emit_links) links of the child tape for the prune of ITS children.
// No register copies are ever emitted: a consumer of a decided choice reads the surviving operand's register directly.
// The tapes differ from fh_prune1's (which inserts a copy where an operand outlives the choice that passed it on) in
// register numbers and in those copies only; values are those of the parent tape on the child's region, op for op
// (tests/test_prune2.py evaluates both on points of the tile).
//
//   grid   : 64 / WPB workgroups of WPB waves per slot; wave = one child lane of the slot
//   LDS    : the parent's ops and links, 16 B per op, staged once per workgroup (its WPB children share the parent);
//            per wave: op -> new register (1 B per parent op), the wanted-op mask, E (2 B per choice)
//   limits : <= 8192 ops, <= 4096 choices, <= 255 registers in the CHILD (more: the child keeps the parent tape), one OUTPUT
//            op, the last one - capi.hip checks and keeps fh_prune1 otherwise
#pragma once
#include <hip/hip_runtime.h>

#include "render_state.h"

#define FH_P2_WPB 4
#define FH_P2_MAX_OPS 8192u
#define FH_P2_MAX_CHOICES 4096u
// op classes of a link
enum { FH_LK_OUT = 0, FH_LK_NONE = 1, FH_LK_A = 2, FH_LK_RR = 3, FH_LK_COPY = 4, FH_LK_CRR = 5, FH_LK_CRI = 6 };
// Link of an op, 8 bytes: word 0 = opcode | class << 8 | choice ordinal << 16, word 1 = fa | fb << 16: the producers of operands a and
// b as op indices, 0x8000 | ordinal when the producer is a choice op, 0xFFFF none.  Register copies are looked through.
// Per choice (second table, 2 bytes): the index of the op.
#define FH_LK_NONE16 0xFFFFu
#define FH_LK_CHOICE 0x8000u
#define FH_LK_IMM 0x4000u        // E: the choice's value is its immediate (reg,imm op decided Right): the op stays, as COPY_IMM

static inline __host__ __device__ size_t fh_p2_wave_lds(uint32_t n_ops, uint32_t n_choices) {
    return (((size_t)n_ops + 15) & ~(size_t)15) + 1024 + (((size_t)n_choices * 2 + 15) & ~(size_t)15) + 16;
}

namespace fhp2 {
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t rfl64(uint64_t v) { return (uint64_t)rfl((uint32_t)v) | ((uint64_t)rfl((uint32_t)(v >> 32)) << 32); }
}  // namespace fhp2

// mode 0: slots[big] of `level`, choice words S->chw[big] with `cw_stride` words per slot; mode 2 (tape groups, level 0): slot =
// block * n_tgroups, choice words S->chwr (k_tscatter3d).  links / ctab: the parent's links when it is the root tape (device
// copies made with the tape).
__global__ void __launch_bounds__(FH_P2_WPB * 64) k_prune2(FhRenderState* S, uint32_t level, uint32_t big, uint32_t mode, uint32_t cw_stride,
                                                           const uint2* __restrict__ links, const uint16_t* __restrict__ ctab, uint32_t emit_links) {
    using namespace fhp2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lane = threadIdx.x & 63, wave = rfl(threadIdx.x >> 6);      // (everything the walk branches on is made wave-uniform explicitly)
    const uint32_t per_slot = 64 / FH_P2_WPB;
    const uint32_t sidx = blockIdx.x / per_slot;
    const uint32_t G = mode == 2 ? rfl(S->n_tgroups) : 1u;
    if (sidx * G >= rfl(S->n_slots[big][level])) return;
    FhSlot& sl = S->slots[big][(size_t)sidx * G];
    if (sl.act == 0) return;
    const uint32_t c0 = (blockIdx.x % per_slot) * FH_P2_WPB, c = c0 + wave;      // this wave's child lane
    {   // any of this workgroup's children marked for the prune?  (c_len == ~0: k_tmark3d / the export mode of the forward kernels)
        bool any = false;
        for (uint32_t k = 0; k < FH_P2_WPB; k++) any |= sl.c_len[c0 + k] == 0xFFFFFFFFu;
        if (!any) return;
    }
    const uint32_t off = rfl(sl.tape.off), n = rfl(sl.tape.len), nch = rfl((uint32_t)sl.tape.n_choices);
    uint4* const recs = (uint4*)smem;
    {   // stage the parent: {link word 0, link word 1, op word 1 (immediate), op word 0} per op
        const uint2* const ops = (const uint2*)(S->arena + off);
        for (uint32_t i = threadIdx.x; i < n; i += FH_P2_WPB * 64) {
            const uint2 o = ops[i], l = links[i];
            recs[i] = make_uint4(l.x, l.y, o.y, o.x);
        }
    }
    char* const mine = smem + (size_t)n * 16 + (size_t)wave * fh_p2_wave_lds(n, nch);
    uint8_t* const map = (uint8_t*)mine;                                   // op index -> new register, 0xFF: none
    uint64_t* const mask = (uint64_t*)(mine + (((size_t)n + 15) & ~(size_t)15));    // wanted ops, 64 per word (128 words)
    uint16_t* const E = (uint16_t*)((char*)mask + 1024);                  // per choice: the op its value is (| FH_LK_IMM)
    const bool marked = rfl(sl.c_len[c]) == 0xFFFFFFFFu;
    const uint32_t n_words = (n + 63) >> 6;
    if (marked) {
        for (uint32_t k = lane; k < (n + 15) / 16; k += 64) ((uint4*)map)[k] = make_uint4(~0u, ~0u, ~0u, ~0u);
        for (uint32_t k = lane; k < n_words; k += 64) mask[k] = 0;
    }
    __syncthreads();
    if (!marked) return;

    // ---- A: what every choice op's value is -------------------------------------------------------------------------------------
    // 64 ordinals at a time, in tape order: an operand's producer has a lower ordinal, so a pointer out of the batch lands on a
    // final entry, and pointers inside the batch are followed by six rounds of jumping between lanes.
    {
        const uint32_t* const cws = mode == 2 ? S->chwr + (size_t)sidx * G * cw_stride * 64 : S->chw[big] + (size_t)sidx * cw_stride * 64;
        for (uint32_t q0 = 0; q0 < nch; q0 += 64) {
            const uint32_t q = q0 + lane;
            uint32_t e = 0;
            if (q < nch) {
                const uint32_t i = ctab[q];
                const uint4 r = recs[i];
                const uint32_t ch = (cws[(size_t)(q >> 4) * 64 + c] >> ((q & 15) * 2)) & 3u;
                const uint32_t kind = (r.x >> 8) & 0xFFu;
                if (ch == FH_CHOICE_LEFT) e = r.y & 0xFFFFu;
                else if (ch == FH_CHOICE_RIGHT) e = kind == FH_LK_CRR ? r.y >> 16 : (i | FH_LK_IMM);
                else e = i;
                if ((e & FH_LK_CHOICE) && (e & 0x7FFFu) < q0) e = E[e & 0x7FFFu];     // out of the batch: final
            }
#pragma unroll
            for (int round = 0; round < 6; round++) {
                const uint32_t t = __shfl(e, (int)((e & 0x7FFFu) - q0) & 63, 64);
                if (e & FH_LK_CHOICE) e = t;
            }
            if (q < nch) E[q] = (uint16_t)e;
        }
    }
    // ---- B: the walk ------------------------------------------------------------------------------------------------------------
    const uint32_t end = rfl(sl.c_off[c]);        // one past the child's last op (arena index)
    uint64_t* const dst = S->arena;
    uint64_t pool0 = ~0ull, pool1 = ~0ull, pool2 = ~0ull, pool3 = ~0ull;    // free new registers, 1 = free (lowest first)
    uint32_t high = 0, count = 0, kept = 0, overflow = 0;
    auto take = [&]() -> uint32_t {
        const bool a = pool0 != 0, b = pool1 != 0, cc = pool2 != 0;
        const uint64_t p = a ? pool0 : (b ? pool1 : (cc ? pool2 : pool3));
        const uint32_t base = a ? 0u : (b ? 64u : (cc ? 128u : 192u));
        const uint32_t r = base + (p ? (uint32_t)__builtin_ctzll(p) : 63u);
        const uint64_t q = p & (p - 1);
        pool0 = a ? q : pool0; pool1 = (!a && b) ? q : pool1; pool2 = (!a && !b && cc) ? q : pool2; pool3 = (!a && !b && !cc) ? q : pool3;
        overflow |= (r >= 255u) ? 1u : 0u;        // (255 = "none" in the map; a child of a <= 128-register parent never gets there)
        high = max(high, r + 1);
        return r;
    };
    auto give = [&](uint32_t r) {
        const uint64_t b = 1ull << (r & 63);
        const uint32_t w = r >> 6;
        pool0 |= w == 0 ? b : 0ull; pool1 |= w == 1 ? b : 0ull; pool2 |= w == 2 ? b : 0ull; pool3 |= w == 3 ? b : 0ull;
    };
    // the wanted-op mask: `s0` / `s1` say which of its words are non-zero, (cw, cb) is the word being walked (bits at and above
    // the last visit cleared).  Marks always go to lower ops than the one being visited.  (Written without branches between
    // the words: the optimiser otherwise turns them into one indexed update of a stack array.)
    uint64_t s0 = 0, s1 = 0;
    uint32_t cw = (n - 1) >> 6;
    uint64_t cb = 1ull << ((n - 1) & 63);       // the OUTPUT op, the last of the tape
    auto mark = [&](uint32_t t) {
        const uint32_t w = t >> 6;
        const uint64_t b = 1ull << (t & 63), sb = 1ull << (w & 63);
        const bool same = w == cw;
        cb |= same ? b : 0ull;
        s0 |= (!same && w < 64) ? sb : 0ull;
        s1 |= (!same && w >= 64) ? sb : 0ull;
        if (!same && lane == 0) atomicOr((unsigned long long*)&mask[w], (unsigned long long)b);
    };
    // the op an operand field stands for
    auto eff = [&](uint32_t f) -> uint32_t { return (f & FH_LK_CHOICE) ? (uint32_t)rfl((uint32_t)E[f & 0x7FFFu]) & 0x3FFFu : f; };
    // register of the value op `p` produces, given one (and the op marked wanted) at its last use = first visit
    auto use = [&](uint32_t p) -> uint32_t {
        uint32_t m = rfl((uint32_t)map[p]);
        if (m == 0xFFu) {
            m = take();
            if (lane == 0) map[p] = (uint8_t)m;
            mark(p);
        }
        return m;
    };
    for (;;) {
        if (cb == 0) {      // the next lower word with a wanted op
            const uint64_t m1 = cw >= 64 ? s1 & ((1ull << (cw & 63)) - 1) : 0ull;
            const uint64_t m0 = cw >= 64 ? s0 : s0 & ((1ull << (cw & 63)) - 1);
            if ((m0 | m1) == 0) break;
            cw = m1 ? 127u - (uint32_t)__builtin_clzll(m1) : 63u - (uint32_t)__builtin_clzll(m0);
            cb = rfl64(mask[cw]);
            continue;
        }
        const uint32_t bit = 63u - (uint32_t)__builtin_clzll(cb);
        cb &= ~(1ull << bit);
        const uint32_t i = (cw << 6) | bit;
        const uint4 r = recs[i];
        const uint32_t hdr = rfl(r.x), pf = rfl(r.y), w1 = rfl(r.z);
        const uint32_t op = hdr & 0xFFu, kind = (hdr >> 8) & 0xFFu;
        if (kind == FH_LK_OUT) {
            const uint32_t na = use(eff(pf & 0xFFFFu));
            count++;
            if (lane == 0) dst[end - count] = fh_pack(op, 0, na, 0, w1);
            continue;
        }
        const uint32_t no = rfl((uint32_t)map[i]);
        give(no);
        bool imm = false;
        if (kind >= FH_LK_CRR) imm = (rfl((uint32_t)E[hdr >> 16]) & FH_LK_IMM) != 0;
        uint64_t word;
        if (imm) word = fh_pack(FH_COPY_IMM, no, 0, 0, w1);
        else {
            uint32_t na = 0, nb = w1;
            if (kind != FH_LK_NONE) na = use(eff(pf & 0xFFFFu));
            if (kind == FH_LK_RR || kind == FH_LK_CRR) nb = use(eff(pf >> 16));
            kept += kind >= FH_LK_CRR ? 1u : 0u;
            word = (uint64_t)(op | (no << 8) | (na << 20)) | ((uint64_t)nb << 32);
        }
        count++;
        if (lane == 0) dst[end - count] = word;
    }
    if (overflow) {     // more than 255 registers: the child keeps the parent tape (what the arena-overflow path does too)
        if (lane == 0) { sl.c_off[c] = off; sl.c_len[c] = n; sl.c_rc[c] = (uint32_t)sl.tape.n_regs | ((uint32_t)sl.tape.n_choices << 16); }
        return;
    }
    const uint32_t start = end - count;
    if (lane == 0) { sl.c_off[c] = start; sl.c_len[c] = count; sl.c_rc[c] = high | (kept << 16); }
    (void)emit_links;
}
