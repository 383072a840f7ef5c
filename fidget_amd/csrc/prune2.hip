// k_prune2 - the linked prune: VmData::simplify for ONE child tile per wavefront, with the work that is sequential by nature
// reduced to the ops the child keeps, and everything else done 64 ops at a time.
//
// VmData::simplify (fidget-core/src/vm/data.rs:123-318, restated as prune_sweep in kernels.hip) walks the parent tape
// backwards and asks of every op whether its output register is wanted.  The scalar sweep fh_prune1 does that for a child of the
// root tape in ~115 instructions per kept op, one after the other: 343 us for prospero's root level (a child keeps 579 ops in
// the median, 1 011 at most, of 6 363), the longest kernel of the frame.  What is truly sequential in simplify is only the
// register allocation; the rest is data parallel once register numbers are out of the way.
//
// The tape comes with LINKS (host_graph.hpp compute_links, made once per tape): per op, which op produced each operand - in SSA
// terms - and for a producer that is itself a min / max / and / or, the ordinal of that choice instead.  Then, per child:
//   A  the choices turn every choice op into "kept", "is its left / right operand" or "is its immediate"; chains of passed-on
//      operands (prospero's root is a chain of 664 min ops) are followed for ALL choice ops at once by pointer jumping, 64
//      ordinals per step in tape order: E[q] = the op whose value choice q's output really is;
//   B1 liveness: a work queue from the OUTPUT op, up to 64 queued ops a round, lane = op: a lane sets the bits of the ops that really
//      produce its op's operands (through E) in a mask of wanted ops, and whoever finds a bit clear queues that op.  When the root is
//      a chain acc = min(acc, term) (plan.chain: most models) its kept ops are found without walking it and queued at once, so the
//      rounds are as many as the deepest TERM is deep (10 - 25), not as the chain is long;
//   B2 the kept ops' positions in the child tape (prefix population counts), per kept op its operands' positions and whether it
//      is their last use (an atomic maximum per value);
//   B3 the one sequential step, in tape order over the KEPT ops only: linear scan, lowest free register first (optimal for a
//      straight-line program).  The registers themselves hold the time they come back - a VGPR, lane = register, value = position of
//      the last use of what the register holds - so an op finds the free ones with ONE compare over all 64, takes the lowest bit, and
//      two v_writelane record the register's new last use and the op's register: 13 instructions per kept op in inline assembly
//      (p2_scan_batch2), no look-up, no memory, no branch but the loop's;
//   B4 the child's ops, 64 at a time.
// No register copies are ever emitted: a consumer of a decided choice reads the surviving operand's register directly.  The
// tapes differ from fh_prune1's (which re-uses the parent's structure and inserts a copy where an operand outlives the choice
// that passed it on) in register numbers and in those copies; values are those of the parent tape on the child's region, op for
// op (tests/test_prune2.py evaluates both on points of the tile).
//
//   grid   : FH_P2_WPC waves per child lane of a slot, FH_P2_WPB children per workgroup
//   limits : <= 8192 ops, <= 4096 choices in the parent, <= cap_kept kept ops and <= 64 registers in the child (more: the child
//            is left to the scalar sweep launched behind this kernel), one OUTPUT op, the last one - capi.hip checks and keeps fh_prune1 otherwise
//
// (Handing the links down - a child tape written here carrying its own links for a linked prune of level 1 - was built and measured in
// round 3: one wave per 32^3 child, 6 120 of them, 0.75 ms where the lockstep sweep of fh_tiles_v64 prunes a parent's children at once in
// 0.26; taken out in round 5, DESIGN_HISTORY.md.  With root tiles of 32^3 the children of THIS kernel are the 32^3 tiles.)
#pragma once
#include <hip/hip_runtime.h>

#include "render_state.h"

#define FH_P2_WPB 4                                             // children per workgroup (they share the root chain's table in LDS)
#define FH_P2_WPC 4                                             // waves per child (the phases that are parallel over the tape; B1 and B3 are one wave's)
#define FH_P2_PER_SLOT ((64 + FH_P2_WPB - 1) / FH_P2_WPB)      // ... workgroups per slot (the last one's spare waves idle)
#define FH_P2_MAX_OPS 8192u
#define FH_P2_MAX_CHOICES 4096u
#define FH_P2_MAX_KEPT 1280u                                    // (four children's areas: 104 KB of the CU's 160)
// op classes of a link
enum { FH_LK_OUT = 0, FH_LK_NONE = 1, FH_LK_A = 2, FH_LK_RR = 3, FH_LK_COPY = 4, FH_LK_CRR = 5, FH_LK_CRI = 6 };
// Link of an op, 8 bytes: word 0 = opcode | class << 8 | choice ordinal << 16, word 1 = fa | fb << 16: the producers of operands a and
// b as op indices, 0x8000 | ordinal when the producer is a choice op, 0xFFFF none.  Register copies are looked through.
// Per choice (second table, 8 bytes, so that a batch of ordinals is one coalesced load): word 0 = fa | fb << 16 of the op, word 1 =
// its index | class << 16.
#define FH_LK_CHOICE 0x8000u
#define FH_LK_IMM 0x4000u        // E: the choice's value is its immediate (reg,imm op decided Right): the op stays, as COPY_IMM

// bytes of LDS per wave: wanted-op mask (128 x 8), position prefixes (128 x 2), E (2 per choice), kept-op records (8 each), last
// uses (4 each), registers by position (1 each; 2 spare bytes each)
static inline __host__ __device__ size_t fh_p2_wave_lds(uint32_t n_choices, uint32_t cap_kept = FH_P2_MAX_KEPT) {
    return 1024 + 256 + (((size_t)n_choices * 2 + 15) & ~(size_t)15) + (size_t)cap_kept * (8 + 4 + 1 + 2) + 64;       // (the last 64: FH_P2_WPC waves' shared words)
}

namespace fhp2 {
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t rfl64(uint64_t v) { return (uint64_t)rfl((uint32_t)v) | ((uint64_t)rfl((uint32_t)(v >> 32)) << 32); }
__device__ __forceinline__ uint32_t excl_sum(uint32_t v, uint32_t lane, uint32_t& total) {
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if ((int)lane >= d) x += y;
    }
    total = __shfl(x, 63, 64);
    return x - v;
}
// B3's loop: the registers themselves keep the time they come back.  rel: lane r = the position of the last use
// of the value that register r holds (0: never used).  Op k of the batch, at position kabs, may take every register with rel <= kabs -
// a value that dies AT op k gives its register to op k's result, as linear scan does -: one compare for all 64 registers, the lowest set
// bit, two v_writelane (the register's new last use; the op's register).  u: lane k = last use of op k's value.  Nothing is posted
// anywhere, nothing is looked up, no branch but the loop's: 13 instructions per kept op.  Lowest free register first.
__device__ __forceinline__ void p2_scan_batch2(uint32_t u, uint32_t& rel, uint32_t& out, uint32_t& maxro, uint32_t cnt, uint32_t kabs0) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t k, su, ro, kabs = kabs0;
    asm volatile(
        "s_mov_b32 %[k], 0\n"
        "L_p2b_loop_%=:\n\t"
        "v_readlane_b32 %[su], %[u], %[k]\n\t"
        "v_cmp_ge_u32 vcc, %[kabs], %[rel]\n\t"
        "s_add_u32 %[kabs], %[kabs], 1\n\t"
        "s_nop 0\n\t"
        "s_ff1_i32_b64 %[ro], vcc\n\t"
        "s_mov_b32 m0, %[ro]\n\t"
        "s_max_u32 %[maxro], %[maxro], %[ro]\n\t"
        "s_nop 0\n\t"
        "v_writelane_b32 %[rel], %[su], m0\n\t"
        "s_mov_b32 m0, %[k]\n\t"
        "s_add_u32 %[k], %[k], 1\n\t"
        "v_writelane_b32 %[out], %[ro], m0\n\t"
        "s_cmp_lt_u32 %[k], %[cnt]\n\t"
        "s_cbranch_scc1 L_p2b_loop_%="
        : [rel] "+v"(rel), [out] "+v"(out), [maxro] "+s"(maxro), [kabs] "+s"(kabs), [k] "=&s"(k), [su] "=&s"(su), [ro] "=&s"(ro)
        : [u] "v"(u), [cnt] "s"(cnt)
        : "m0", "scc", "vcc", "memory");
#else
    (void)u; (void)rel; (void)out; (void)maxro; (void)cnt; (void)kabs0;
#endif
}
}  // namespace fhp2

// Tape groups, level 0: slot = block * n_tgroups, choice words S->chwr (k_tscatter3d) with `cw_stride` words per slot, links / ctab:
// the root tape's (device copies made with the tape).  cap_ops / cap_choices / cap_kept size the LDS areas (links, E, per kept op
// records).  `flags` bit 1: profiled frames record the slowest child's shader clocks per phase.
// One work item: the (up to) FH_P2_WPB children `blk` % per_slot of slot `blk` / per_slot, FH_P2_WPC waves each.
//
// Several waves per child (round 5).  A child used to be ONE wave, alone on its SIMD: 0.19 ms of instruction latency whatever the number
// of children, the longest kernel of a frame once level 1 was gone.  What of the work is parallel over the tape is now shared by the
// child's FH_P2_WPC waves (workgroup barriers in between; every wave of the workgroup passes every barrier, a child that is not
// marked - or drops out: too many kept ops, too many registers - idles through them):
//   A  in two passes: every wave resolves the batches of 64 ordinals it owns WITHOUT looking outside the batch (a pointer out of
//      the batch stays a pointer), then one wave walks the batches in order and replaces what is still a pointer by its target's
//      entry, final by then - one look-up per batch instead of the six jumping rounds;
//   B1 liveness: one wave (a sweep from the end of the tape: every batch needs the marks of the batches behind it);
//   B2 positions / operand positions / last uses: batches shared out; B3 the register scan: one wave; B4 emission: shared out.
__device__ __forceinline__ void p2_item(FhRenderState* S, uint32_t level, uint32_t big, uint32_t cw_stride, const uint2* __restrict__ links,
                                        const uint2* __restrict__ ctab, uint32_t flags, uint32_t cap_ops, uint32_t cap_choices, uint32_t cap_kept,
                                        const uint32_t* __restrict__ chain, uint32_t n_chain, uint32_t blk, char* smem) {
    using namespace fhp2;
    constexpr uint32_t W = FH_P2_WPC;
    const uint32_t lane = threadIdx.x & 63, wave = rfl(threadIdx.x >> 6);      // (everything the sequential step branches on is made wave-uniform explicitly)
    const uint32_t wpb = FH_P2_WPB, per_slot = (64 + wpb - 1) / wpb;
    const uint32_t kid = wave / W, sub = wave % W, tid = sub * 64 + lane;       // child of this workgroup, wave of that child, thread of that child
    // (the wave that runs a child's sequential phases: wave `kid` of the child, so that the workgroup's four leaders sit on four
    // different SIMDs - consecutive waves of a workgroup go round the SIMDs, and with wave 0 of every child as the leader all four shared
    // SIMD 0 while the others idled at the barrier: B3 took 1.5 x as long as with one wave per child)
    const bool lead = sub == kid % W;
    const uint32_t sidx = blk / per_slot;
    const uint32_t G = rfl(S->n_tgroups);
    if (sidx * G >= rfl(S->n_slots[big][level])) return;
    FhSlot& sl = S->slots[big][(size_t)sidx * G];
    if (sl.act == 0) return;
    const uint32_t c0 = (blk % per_slot) * wpb, c = c0 + kid;      // this wave's child lane
    {   // any of this workgroup's children marked for the prune?  (c_len == ~0: k_tmark3d / the export mode of the forward kernels)
        bool any = false;
        for (uint32_t k = 0; k < wpb; k++) any |= c0 + k < 64 && sl.c_len[c0 + k] == 0xFFFFFFFFu;
        if (!any) return;
    }
    const uint32_t off = rfl(sl.tape.off), n = rfl(sl.tape.len), nch = rfl((uint32_t)sl.tape.n_choices);
    if (n > cap_ops || nch > cap_choices) return;                       // (left marked)
    const uint2* const ops = (const uint2*)(S->arena + off);
    // the parent's links are read where they lie, through L2 (8 B per op, the same 51 KB for every workgroup).  Until the end of round 5
    // every workgroup staged them in LDS - 155 KB per workgroup with the four children's areas, a compute unit's LDS to itself - for the
    // liveness pass's sake, which then was a sweep over all of them; the queue reads the links of kept ops only, and with 104 KB other
    // streams' kernels fit beside this one: 0.139 -> 0.135 ms per frame (profiles/r05i/sweep10)
    const uint2* const lks = links;
    // (the root chain's ops, evaluation order: choice ordinal | op index << 16 - behind the children's areas, shared like the links)
    uint32_t* const chl = (uint32_t*)(smem + (size_t)FH_P2_WPB * fh_p2_wave_lds(cap_choices, cap_kept));
    for (uint32_t i = threadIdx.x; i < n_chain; i += blockDim.x) chl[i] = chain[i];
    char* const mine = smem + (size_t)kid * fh_p2_wave_lds(cap_choices, cap_kept);
    uint64_t* const mask = (uint64_t*)mine;                               // wanted ops, 64 per word (128 words)
    uint16_t* const pref = (uint16_t*)(mine + 1024);                    // kept ops before each word
    uint16_t* const E = (uint16_t*)(mine + 1280);                       // per choice: the op its value is (| FH_LK_IMM)
    uint2* const comp = (uint2*)(mine + 1280 + (((size_t)cap_choices * 2 + 15) & ~(size_t)15));       // per kept op: operand positions, op index | flags << 16
    uint32_t* const lastuse = (uint32_t*)((char*)comp + (size_t)cap_kept * 8);
    uint8_t* const regb = (uint8_t*)((char*)lastuse + (size_t)cap_kept * 4);
    uint32_t* const red = (uint32_t*)(regb + (size_t)cap_kept * 3);     // [0] kept ops m (B2), [1] highest register + 1, [2] kept choices (B4), [3] 0: the child goes on (the area's last 64 bytes)
    const uint32_t nw = (n + 63) >> 6;
    bool on = c < 64 && rfl(sl.c_len[min(c, 63u)]) == 0xFFFFFFFFu;    // marked for the prune; false: this wave only keeps the barriers company
    if (on && tid < 4) red[tid] = 0;
    const uint32_t* const cws = S->chwr + (size_t)sidx * G * cw_stride * 64;
    uint32_t* const cwl = (uint32_t*)comp;       // (the child's choice words, staged where the kept-op records go later: one load latency for all)
    if (on) for (uint32_t k = tid; k < (nch + 15) / 16; k += 64 * W) cwl[k] = cws[(size_t)k * 64 + c];
    __syncthreads();

    const bool probe = S->want_stats != 0 && (flags & 2u) != 0 && lead;      // (profiled frames: the slowest child's shader clocks per phase, leaf_stat[4..7])
    const uint64_t t_a = probe ? clock64() : 0;
    // ---- A: what every choice op's value is -------------------------------------------------------------------------------------
    // pass 1, batches of 64 ordinals shared out over the child's waves: pointers inside the batch are followed by jumping between lanes (a
    // producer has a lower ordinal: at most six rounds), pointers out of it are left standing
    if (on) {
        const uint32_t nb = (nch + 63) >> 6;
        uint2 tn = (sub * 64 + lane) < nch ? ctab[sub * 64 + lane] : make_uint2(0, 0);
        for (uint32_t bi = sub; bi < nb; bi += W) {
            const uint32_t q0 = bi << 6, q = q0 + lane;
            const uint2 t = tn;
            if (q + 64 * W < nch) tn = ctab[q + 64 * W];      // (the wave's next batch arrives while this one is resolved)
            uint32_t e = 0;
            if (q < nch) {
                const uint32_t i = t.y & 0xFFFFu, kind = t.y >> 16;
                const uint32_t ch = (cwl[q >> 4] >> ((q & 15) * 2)) & 3u;
                if (ch == FH_CHOICE_LEFT) e = t.x & 0xFFFFu;
                else if (ch == FH_CHOICE_RIGHT) e = kind == FH_LK_CRR ? t.x >> 16 : (i | FH_LK_IMM);
                else e = i;
            }
            // (0xFFFF - no operand - never comes out of a choice table entry that is taken: a choice's operands exist)
            for (int round = 0; round < 6; round++) {
                const bool inside = (e & FH_LK_CHOICE) != 0 && (e & 0x7FFFu) >= q0;
                if (__ballot(inside) == 0) break;
                const uint32_t t2 = __shfl(e, (int)((e & 0x7FFFu) - q0) & 63, 64);
                if (inside) e = t2;
            }
            if (q < nch) E[q] = (uint16_t)e;
        }
    }
    __syncthreads();
    // pass 2, one wave, batches in order: what still points out of its batch points at an entry that is final by now
    if (on && lead) {
        for (uint32_t q0 = 0; q0 < nch; q0 += 64) {
            const uint32_t q = q0 + lane;
            const uint32_t e = q < nch ? (uint32_t)E[q] : 0u;
            const bool ptr = (e & FH_LK_CHOICE) != 0;
            if (__ballot(ptr) == 0) continue;
            if (ptr) E[q] = E[e & 0x7FFFu];
        }
    }
    __syncthreads();
    // The ops a kept op's operands really come from (its link in l).  The three E entries it may need - its own choice's (a reg,imm
    // choice that became its immediate has no operands), its operands' producers' when those are choices - are loaded together.
    auto resolve = [&](uint2 l, bool& has_a, bool& has_b, bool& imm, uint32_t& ta, uint32_t& tb) {
        const uint32_t kind = (l.x >> 8) & 0xFFu, fa = l.y & 0xFFFFu, fb = l.y >> 16;
        const uint32_t e_own = E[kind == FH_LK_CRI ? (l.x >> 16) : 0u];
        // (0xFFFF = no operand, and the padding link 0xFFFFFFFF, have the choice bit set too: entry 0 for them, never index 0x7FFF)
        const uint32_t e_a = E[((fa & FH_LK_CHOICE) && fa != 0xFFFFu) ? (fa & 0x7FFFu) : 0u], e_b = E[((fb & FH_LK_CHOICE) && fb != 0xFFFFu) ? (fb & 0x7FFFu) : 0u];
        imm = kind == FH_LK_CRI && (e_own & FH_LK_IMM) != 0;
        has_a = !imm && kind != FH_LK_NONE;
        has_b = !imm && (kind == FH_LK_RR || kind == FH_LK_CRR);
        ta = (fa & FH_LK_CHOICE) ? (e_a & 0x3FFFu) : fa;
        tb = (fb & FH_LK_CHOICE) ? (e_b & 0x3FFFu) : fb;
    };

    const uint64_t t_b1 = probe ? clock64() : 0;
    // ---- B1: liveness (one wave) ---------------------------------------------------------------------------------------------------
    if (on && lead) {
        // A work queue from the OUTPUT op: up to 64 queued ops a round, lane = op; a lane looks up its op's producers (through E) and sets
        // their bits in the mask, and whoever finds a bit clear queues that op - once.  A child of the root tape keeps 100 - 230 of its
        // 6 363 ops.  (What this replaced: a sweep over the tape's 100 batches of 64 ops from its end, which met a wanted op in nearly
        // every batch and paid 1 300 cycles for each - 128 k cycles of the kernel's 360 k.  The queue alone was no faster, 126 k: the kept
        // ops of the root chain are one dependent path, a round each.  With the chain's head start below: 50 k.)
        // The queue (2 bytes per op, each kept op enters once) lies where the kept-op records go in B2.
        for (uint32_t k = lane; k < 128; k += 64) mask[k] = (k == ((n - 1) >> 6)) ? 1ull << ((n - 1) & 63) : 0ull;      // the OUTPUT op, the last of the tape
        // (the queue lies behind the child's staged choice words, which the head start below still reads)
        const uint32_t cw_bytes = (((nch + 15) / 16) * 4 + 15) & ~15u;
        uint16_t* const queue = (uint16_t*)((char*)comp + cw_bytes);
        const uint32_t qcap = (cap_kept * 8u - cw_bytes) / 2u;
        if (lane == 0) queue[0] = (uint16_t)(n - 1);
        uint32_t head = 0, tail = 1;
        bool over = false;
        const uint64_t below = (1ull << lane) - 1;
        // Head start: the root of most models is a chain, acc = min(acc, term) from the first term to the OUTPUT op (prospero: 664 ops),
        // and the kept ops of that chain are one dependent path - 60 - 100 rounds of this queue with one useful lane each.  But which of
        // them are wanted is known without walking: from the chain's end down, an op that kept both operands is wanted and the walk goes
        // on below it, one that took its left operand (the chain) is passed through, the first that took its right operand (its term) ends
        // the chain - everything below is dead from here.  So the chain is read 64 ops at a time from its end, its kept ops are marked
        // and queued at once, and the rounds that follow are as deep as the deepest TERM (10 - 25).
        for (uint32_t top = n_chain; top > 0;) {
            const uint32_t cnt = min(64u, top);
            uint32_t ch = 0, i = 0;
            if (lane < cnt) {
                const uint32_t e = chl[top - 1 - lane], q = e & 0xFFFFu;
                i = e >> 16;
                ch = (cwl[q >> 4] >> ((q & 15) * 2)) & 3u;
            }
            const uint64_t cut = __ballot(ch == FH_CHOICE_RIGHT);
            const bool seed = ch == FH_CHOICE_BOTH && (cut == 0 || lane < (uint32_t)__builtin_ctzll(cut));
            if (seed) atomicOr((unsigned long long*)&mask[i >> 6], 1ull << (i & 63));
            const uint64_t ms = __ballot(seed);
            if (tail + (uint32_t)__popcll(ms) > qcap) { over = true; break; }
            if (seed) queue[tail + (uint32_t)__popcll(ms & below)] = (uint16_t)i;
            tail += (uint32_t)__popcll(ms);
            if (cut) break;
            top -= cnt;
        }
        while (!over && head < tail) {
            const uint32_t cnt = min(64u, tail - head);
            bool has_a = false, has_b = false, imm = false;
            uint32_t ta = 0, tb = 0;
            if (lane < cnt) resolve(lks[queue[head + lane]], has_a, has_b, imm, ta, tb);
            head += cnt;
            bool new_a = false, new_b = false;
            if (has_a) { const uint64_t bit = 1ull << (ta & 63); new_a = (atomicOr((unsigned long long*)&mask[ta >> 6], bit) & bit) == 0; }
            if (has_b) { const uint64_t bit = 1ull << (tb & 63); new_b = (atomicOr((unsigned long long*)&mask[tb >> 6], bit) & bit) == 0; }
            const uint64_t ma = __ballot(new_a), mb = __ballot(new_b);
            const uint32_t na = (uint32_t)__popcll(ma), nb = (uint32_t)__popcll(mb);
            if (tail + na + nb > qcap) { over = true; break; }       // (more than the areas hold: the scalar sweep takes this child)
            if (new_a) queue[tail + (uint32_t)__popcll(ma & below)] = (uint16_t)ta;
            if (new_b) queue[tail + na + (uint32_t)__popcll(mb & below)] = (uint16_t)tb;
            tail += na + nb;
        }
        if (over && lane == 0) red[3] = 1;
    }
    if (on && lead) {
        // positions of the words' first kept ops (B2 needs them all)
        const uint32_t k0 = lane < nw ? (uint32_t)__popcll(mask[lane]) : 0u, k1 = lane + 64 < nw ? (uint32_t)__popcll(mask[lane + 64]) : 0u;
        uint32_t t0, t1;
        const uint32_t e0 = excl_sum(k0, lane, t0), e1 = excl_sum(k1, lane, t1);
        pref[lane] = (uint16_t)e0; pref[lane + 64] = (uint16_t)(t0 + e1);
        if (lane == 0) { red[0] = t0 + t1; if (t0 + t1 > cap_kept) red[3] = 1; }
    }
    __syncthreads();
    const uint64_t t_b2 = probe ? clock64() : 0;
    // ---- B2: positions, operand positions, last uses (batches shared out) -----------------------------------------------------------
    const uint32_t m = on ? rfl(red[0]) : 0u;
    if (on && rfl(red[3]) != 0) {          // (more kept ops than the areas hold: left marked, the scalar sweep launched behind this kernel takes it)
        if (probe && lane == 0) atomicAdd(&S->leaf_stat[5], 1ull << 32);
        on = false;
    }
    if (on) for (uint32_t k = tid; k < m; k += 64 * W) lastuse[k] = 0;
    __syncthreads();
    auto pos_of = [&](uint32_t t) -> uint32_t { return (uint32_t)pref[t >> 6] + (uint32_t)__popcll(mask[t >> 6] & ((1ull << (t & 63)) - 1)); };
    if (on) {
        for (uint32_t b = sub; b < nw; b += W) {
            const uint64_t word = rfl64(mask[b]);
            if (word == 0) continue;
            if ((word >> lane) & 1) {
                const uint32_t i = (b << 6) | lane;
                const uint32_t p = (uint32_t)pref[b] + (uint32_t)__popcll(word & ((1ull << lane) - 1));
                const uint2 l = lks[i];
                bool has_a, has_b, imm;
                uint32_t ta, tb;
                resolve(l, has_a, has_b, imm, ta, tb);
                const uint32_t kind = (l.x >> 8) & 0xFFu;
                uint32_t pa = 0, pb = 0;
                if (has_a) { pa = pos_of(ta); atomicMax(&lastuse[pa], p); }
                if (has_b) { pb = pos_of(tb); atomicMax(&lastuse[pb], p); }
                // flags: 0 has a, 1 has b, 2 became its immediate, 3 a kept choice, 4 the OUTPUT op
                const uint32_t fl = (has_a ? 1u : 0u) | (has_b ? 2u : 0u) | (imm ? 4u : 0u) | ((kind >= FH_LK_CRR && !imm) ? 8u : 0u) | (kind == FH_LK_OUT ? 16u : 0u);
                comp[p] = make_uint2(pa | (pb << 16), i | (fl << 16));
            }
        }
    }
    __syncthreads();
    const uint64_t t_b3 = probe ? clock64() : 0;
    // ---- B3: registers, in tape order over the kept ops (one wave) ------------------------------------------------------------------
    // Linear scan: a value's register is not looked up when an op reads it (that is done for all ops at once in B4) - what the sequential
    // step needs is only WHEN registers come back, and that is kept in the registers' own lanes (p2_scan_batch2).  Registers 0 .. 63 only:
    // a child that wants more is left to the scalar sweep.  (Until late in round 5 a value posted "free r" to the op where it dies - the
    // op's record in LDS, or a VGPR copy for an op of the same batch - and every op began by collecting what was posted to it: 33
    // instructions and up to three taken branches per kept op, 113 k cycles for the slowest child where this loop takes 53 k.)
    if (on && lead) {
        uint32_t maxro = 0, rel = 0;
        for (uint32_t base = 0; base < m; base += 64) {
            const uint32_t pl = base + lane;
            const uint32_t u = pl < m ? lastuse[pl] : 0u;          // (0 for the OUTPUT op, the last one: it takes the register its operand gives back, which nobody reads)
            uint32_t outv = 0;
            p2_scan_batch2(u, rel, outv, maxro, min(64u, m - base), base);
            if (pl < m) regb[pl] = (uint8_t)outv;
        }
        if (maxro >= 64u) {          // more than 64 registers: left marked for the scalar sweep
            if (probe && lane == 0) { atomicAdd(&S->leaf_stat[4], 1ull << 32); atomicMax(&S->leaf_stat[6], (unsigned long long)m << 32); }
            if (lane == 0) red[3] = 1;
        }
    }
    __syncthreads();
    if (on && rfl(red[3]) != 0) on = false;
    // ---- B4: the child's ops (shared out) ----------------------------------------------------------------------------------------------
    const uint32_t end = on ? rfl(sl.c_off[c]) : 0u;        // one past the child's last op (arena index); the child's slot is [end - n, end)
    if (on) {
        uint64_t* const dst = S->arena + (end - m);
        uint32_t high = 0, kept = 0;
        for (uint32_t pl = tid; pl < m; pl += 64 * W) {
            const uint2 cr = comp[pl];
            const uint32_t fl = cr.y >> 16, i = cr.y & 0xFFFFu, pa = cr.x & 0xFFFFu, pb = cr.x >> 16;
            const uint2 opw = ops[i];
            const uint32_t ro = regb[pl], ra = (fl & 1u) ? regb[pa] : 0u, rb = (fl & 2u) ? regb[pb] : 0u;
            uint64_t word;
            if (fl & 4u) word = fh_pack(FH_COPY_IMM, ro, 0, 0, opw.y);
            else if (fl & 16u) word = fh_pack(FH_OUTPUT, 0, ra, 0, opw.y);
            else word = (uint64_t)((opw.x & 0xFFu) | (ro << 8) | (ra << 20)) | ((uint64_t)((fl & 2u) ? rb : opw.y) << 32);
            dst[pl] = word;
            high = max(high, (fl & 16u) ? 0u : ro + 1u);
            kept += (fl >> 3) & 1u;
        }
#pragma unroll
        for (int dlt = 32; dlt > 0; dlt >>= 1) { high = max(high, (uint32_t)__shfl_xor(high, dlt, 64)); kept += (uint32_t)__shfl_xor(kept, dlt, 64); }
        if (lane == 0) { atomicMax(&red[1], high); atomicAdd(&red[2], kept); }
    }
    __syncthreads();
    if (probe && on && lane == 0) {
        const uint64_t t_e = clock64();
        atomicMax(&S->leaf_stat[4], (unsigned long long)(t_b1 - t_a)); atomicMax(&S->leaf_stat[5], (unsigned long long)(t_b2 - t_b1));
        atomicMax(&S->leaf_stat[6], (unsigned long long)(t_b3 - t_b2)); atomicMax(&S->leaf_stat[7], (unsigned long long)(t_e - t_b3));
    }
    if (on && lead && lane == 0) { sl.c_off[c] = end - m; sl.c_len[c] = m; sl.c_rc[c] = red[1] | (red[2] << 16); }
}

// grid: one workgroup per item (FH_P2_WPB children of one slot)
__global__ void __launch_bounds__(FH_P2_WPB * FH_P2_WPC * 64) k_prune2(FhRenderState* S, uint32_t level, uint32_t big, uint32_t cw_stride, const uint2* __restrict__ links,
                                                const uint2* __restrict__ ctab, uint32_t flags, uint32_t cap_ops, uint32_t cap_choices, uint32_t cap_kept,
                                                const uint32_t* __restrict__ chain, uint32_t n_chain) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    p2_item(S, level, big, cw_stride, links, ctab, flags, cap_ops, cap_choices, cap_kept, chain, n_chain, blockIdx.x, smem);
}
