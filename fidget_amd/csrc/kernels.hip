// fidget-hip: CDNA4 (gfx950) kernels for tape evaluation and tile rendering.
//
// Execution model (MI355X-first, not a port of the reference's CPU recursion):
//   * a tape is a read-only array of 8-byte ops in HBM; every lane of a wave walks the
//     SAME tape, so op words are fetched wave-uniformly with scalar loads (s_load through
//     the constant address space, 4 ops = 32 B per request, double buffered) and decoded
//     once per wave on the scalar unit;
//   * the interpreter's per-lane register file lives in VGPRs for the tapes that dominate
//     the run time (<= 32 registers after pruning; uniform dynamic indexing lowers to
//     s_set_gpr_idx_on + v_mov, no memory traffic at all) and in LDS, regs[reg][lane],
//     for larger tapes (bank == lane, conflict free);
//   * interval evaluation maps one TILE per lane; the lanes of a wave are the sibling tiles
//     that share their parent's tape.  The wave then prunes the tape for each child (two
//     reverse sweeps: count, then emit with dense register renumbering) into a bump
//     allocated HBM arena.  Wave ballots / prefix sums allocate arena and queue space with
//     one atomic per wave;
//   * point evaluation maps voxels to lanes, one 8x8 pixel footprint per wave and ZB
//     consecutive z per lane; the leaves of a footprint are walked front to back inside
//     the wave so occluded voxels are never evaluated;
//   * work flows level by level through device-side queues consumed by persistent
//     workgroups; there is no host round trip inside a frame.
//
// Results are order independent: the 3D depth buffer is a 64-bit atomicMax of
// (depth << 32 | leaf id), which reproduces the reference's sequential front-to-back
// "first writer wins" semantics (fidget-raster/src/voxel.rs:275-484).
#include <hip/hip_runtime.h>

#include "dev_ops.hpp"
#include "render_state.h"

using namespace fhd;

#define WAVE 64
#define AS4 __attribute__((address_space(4)))
typedef const AS4 uint64_t* ctape_t;  // constant address space => scalar (SMEM) loads

// 4 tape ops (32 B) per scalar load request; the arena keeps slack past its last tape
typedef unsigned long long Q4 __attribute__((ext_vector_type(4), aligned(8)));
FH_DEV uint64_t q4_pick(Q4 q, int u) { return u == 0 ? q.x : (u == 1 ? q.y : (u == 2 ? q.z : q.w)); }

FH_DEV uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// Arena reservation of `n` ops by one lane.  The bump pointer only ever grows, except that a reservation that does not
// fit clamps it back to the capacity (atomicMin): it can then never fall below a range that was granted (a give-back by
// subtraction could: two failures and a success in between leave it inside the successful range), and it cannot creep
// past 2^32 and wrap under a long run of failures either (arena_cap <= 2^31 ops, at most a few thousand waves in flight
// add <= 2^18 ops each before they clamp).  Returns ~0u on failure.
FH_DEV uint32_t arena_reserve(FhRenderState* S, uint32_t n) {
    const uint32_t base = atomicAdd(&S->arena_head, n);
    if (base + n < base || base + n > S->arena_cap) { atomicMin(&S->arena_head, S->arena_cap); return 0xFFFFFFFFu; }
    return base;
}
FH_DEV uint64_t ballot(bool p) { return __ballot(p); }

// Is `op` outside the "basic" arithmetic set?  Kernels are built twice: BASIC variants keep
// the f64 transcendental code out of the register budget (prospero, colonnade, hi use none).
FH_DEV bool op_is_heavy(uint32_t op) {
    return (op >= FH_SIN && op <= FH_LN) || op == FH_ATAN2_RR || op == FH_ATAN2_RI || op == FH_ATAN2_IR ||
           op == FH_MOD_RR || op == FH_MOD_RI || op == FH_MOD_IR;
}

// Where a root-sized kernel variant keeps its register file: LDS, or this workgroup's region of S->gscratch when the tape's
// file does not fit LDS (render_state.h gscratch: the slow path for tapes of several hundred registers)
FH_DEV char* big_file(FhRenderState* S, char* smem) {
    const uint32_t stride = S->gscratch_stride;
    return stride ? S->gscratch + (size_t)blockIdx.x * stride : smem;
}

// --------------------------------------------------------------------------------------
// LDS register file.  LANES is the lane stride (64 for point kernels, 16 for the tile kernel).
template <class T, int LANES>
struct Regs {
    T* base;
    int lane;
    FH_DEV T get(uint32_t r) const { return base[r * LANES + lane]; }
    FH_DEV void set(uint32_t r, T v) const { base[r * LANES + lane] = v; }
};

// One interpreter step (scalar domains F32 / IVAL / GRAD, LDS or global register file).
template <class D, int LANES, bool FULL, class InFn, class OutFn, class ChoiceFn>
FH_DEV void step(uint64_t w, const Regs<typename D::V, LANES>& R, InFn in, OutFn on_out, ChoiceFn on_choice) {
    typedef typename D::V V;
    const uint32_t w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
    const uint32_t op = FH_W_OP(w0), ro = FH_W_OUT(w0), ra = FH_W_A(w0), rb = w1;
    if (op >= FH_ADD_RR) {
        V a, b;
        int base;
        bool swap_mul = false;
        if (op < FH_ADD_RI) { base = op - FH_ADD_RR; a = R.get(ra); b = R.get(rb); }
        else if (op < FH_SUB_IR) {
            base = op - FH_ADD_RI; a = R.get(ra); b = D::imm(u2f(w1));
            swap_mul = (base == 2);
        } else {
            // imm,reg forms: sub div atan2 compare mix mod -> bases 1 3 4 5 6 7
            const int irb = op - FH_SUB_IR;
            base = irb == 0 ? 1 : irb + 2;
            a = D::imm(u2f(w1)); b = R.get(ra);
        }
        int c = FH_CHOICE_BOTH;
        V r = swap_mul ? D::mul_imm(a, u2f(w1)) : D::template binary<FULL>(base, a, b, c);
        R.set(ro, r);
        if (base >= 8) on_choice(c);
    } else if (op >= FH_NEG) {
        R.set(ro, D::template unary<FULL>(op, R.get(ra)));
    } else if (op == FH_INPUT) {
        R.set(ro, in(w1));
    } else if (op == FH_COPY_REG) {
        R.set(ro, R.get(ra));
    } else if (op == FH_COPY_IMM) {
        R.set(ro, D::imm(u2f(w1)));
    } else {
        on_out(w1, R.get(ra));
    }
}

// ======================================================================================
// Trait-surface kernels (fhip_float_eval / fhip_point_eval / fhip_interval_eval /
// fhip_grad_eval): one sample per lane, one wave per workgroup.
// GREGS: the register file of tapes too large for LDS lives in a global scratch slab.
// ======================================================================================
template <bool GREGS>
__global__ void __launch_bounds__(WAVE)
k_eval_f32(const uint64_t* __restrict__ tape_g, uint32_t len, const float* __restrict__ vars, uint32_t n,
           float* __restrict__ out, uint8_t* __restrict__ choices, uint8_t* __restrict__ simplify, uint32_t n_choices,
           float* gregs, uint32_t n_regs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ctape_t tape = (ctape_t)tape_g;
    const int lane = threadIdx.x;
    const uint32_t i = blockIdx.x * WAVE + lane;
    const bool act = i < n;
    const uint32_t ii = act ? i : 0;
    Regs<float, WAVE> R{GREGS ? gregs + (size_t)blockIdx.x * n_regs * WAVE : (float*)smem, lane};
    uint32_t ci = 0;
    bool simp = false;
    for (uint32_t k = 0; k < len; k++) {
        const uint64_t w = tape[k];
        step<F32, WAVE, true>(
            w, R, [&](uint32_t slot) { return vars[(size_t)slot * n + ii]; },
            [&](uint32_t slot, float v) { if (act) out[(size_t)slot * n + ii] = v; },
            [&](int c) {
                if (choices && act) choices[(size_t)ii * n_choices + ci] = (uint8_t)c;
                simp |= (c != FH_CHOICE_BOTH);
                ci++;
            });
    }
    if (simplify && act) simplify[ii] = simp ? 1 : 0;
}

template <bool GREGS>
__global__ void __launch_bounds__(WAVE)
k_eval_interval(const uint64_t* __restrict__ tape_g, uint32_t len, const float2* __restrict__ vars, uint32_t n_vars,
                uint32_t n, float2* __restrict__ out, uint32_t n_out, uint8_t* __restrict__ choices,
                uint8_t* __restrict__ simplify, uint32_t n_choices, IV* gregs, uint32_t n_regs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ctape_t tape = (ctape_t)tape_g;
    const int lane = threadIdx.x;
    const uint32_t i = blockIdx.x * WAVE + lane;
    const bool act = i < n;
    const uint32_t ii = act ? i : 0;
    Regs<IV, WAVE> R{GREGS ? gregs + (size_t)blockIdx.x * n_regs * WAVE : (IV*)smem, lane};
    uint32_t ci = 0;
    bool simp = false;
    for (uint32_t k = 0; k < len; k++) {
        const uint64_t w = tape[k];
        step<IVAL, WAVE, true>(
            w, R,
            [&](uint32_t slot) { float2 v = vars[(size_t)ii * n_vars + slot]; return iv(v.x, v.y); },
            [&](uint32_t slot, IV v) { if (act) out[(size_t)ii * n_out + slot] = make_float2(v.lo, v.hi); },
            [&](int c) {
                if (choices && act) choices[(size_t)ii * n_choices + ci] = (uint8_t)c;
                simp |= (c != FH_CHOICE_BOTH);
                ci++;
            });
    }
    if (simplify && act) simplify[ii] = simp ? 1 : 0;
}

template <bool GREGS>
__global__ void __launch_bounds__(WAVE)
k_eval_grad(const uint64_t* __restrict__ tape_g, uint32_t len, const float4* __restrict__ vars, uint32_t n,
            float4* __restrict__ out, GR* gregs, uint32_t n_regs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ctape_t tape = (ctape_t)tape_g;
    const int lane = threadIdx.x;
    const uint32_t i = blockIdx.x * WAVE + lane;
    const bool act = i < n;
    const uint32_t ii = act ? i : 0;
    Regs<GR, WAVE> R{GREGS ? gregs + (size_t)blockIdx.x * n_regs * WAVE : (GR*)smem, lane};
    for (uint32_t k = 0; k < len; k++) {
        const uint64_t w = tape[k];
        step<GRAD, WAVE, true>(
            w, R,
            [&](uint32_t slot) { float4 v = vars[(size_t)slot * n + ii]; return gr(v.x, v.y, v.z, v.w); },
            [&](uint32_t slot, GR v) { if (act) out[(size_t)slot * n + ii] = make_float4(v.v, v.dx, v.dy, v.dz); },
            [&](int) {});
    }
}

// ======================================================================================
// Rendering: shared helpers
// ======================================================================================
// Wave-wide exclusive prefix sum over lanes (and total) of a per-lane count
FH_DEV uint32_t wave_excl_sum(uint32_t v, uint32_t& total) {
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint32_t y = __shfl_up(x, d, WAVE);
        if ((int)(threadIdx.x & 63) >= d) x += y;
    }
    total = __shfl(x, WAVE - 1, WAVE);
    return x - v;
}

// 256-bit free-register pool held in VGPRs (lowest free first)
struct Pool {
    uint64_t f0, f1, f2, f3;
    int high;
    bool over;      // more than 255 registers were wanted at once: the tape being written is void (its child keeps the parent's)
    FH_DEV void init() { f0 = f1 = f2 = ~0ull; f3 = ~0ull >> 1; high = 0; over = false; }      // (255 is DEAD in the byte maps)
    FH_DEV int take() {
        int r;
        if (f0) { r = __builtin_ctzll(f0); f0 &= f0 - 1; }
        else if (f1) { r = 64 + __builtin_ctzll(f1); f1 &= f1 - 1; }
        else if (f2) { r = 128 + __builtin_ctzll(f2); f2 &= f2 - 1; }
        else if (f3) { r = 192 + __builtin_ctzll(f3); f3 &= f3 - 1; }
        else { over = true; return 0; }
        high = max(high, r + 1);
        return r;
    }
    FH_DEV void give(int r) {
        const uint64_t b = 1ull << (r & 63);
        const int w = r >> 6;
        if (w == 0) f0 |= b; else if (w == 1) f1 |= b; else if (w == 2) f2 |= b; else f3 |= b;
    }
};

// TL = lane stride of the tile kernel: sibling tiles per wave (16 or 64, template parameter)
#define DEAD 0xFFu

// Reverse sweep over the parent tape for one child tile (one lane):
//   EMIT = false: count surviving ops
//   EMIT = true : write them (back to front) with densely renumbered registers
// This is the device form of VmData::simplify (fidget-core/src/vm/data.rs:123-318):
// ops whose value is never used are dropped, a min/max/and/or whose trace says
// Left/Right is replaced by its surviving operand (aliased when that operand is
// not otherwise live yet, copied when it is).
template <bool EMIT, int TL>
FH_DEV void prune_sweep(ctape_t tape, uint32_t len, uint32_t n_choices, const uint32_t* chbits, uint8_t* map,
                        int lane16, bool act, uint64_t* dst /*one past the last op*/, uint32_t& out_len,
                        uint32_t& out_regs, uint32_t& out_choices) {
    Pool pool;
    pool.init();
    uint32_t ci = n_choices;
    uint32_t cw = 0;
    uint32_t count = 0, kept_choices = 0;
    auto M = [&](uint32_t r) -> uint8_t& { return map[r * TL + lane16]; };
    auto use = [&](uint32_t r) -> uint32_t {  // register holding old value r before this op
        uint32_t m = M(r);
        if (m == DEAD) { m = EMIT ? (uint32_t)pool.take() : 0u; M(r) = (uint8_t)m; }
        return m;
    };
    const AS4 Q4* q = (const AS4 Q4*)tape;
    Q4 cur = q[(len - 1) >> 2];
    for (uint32_t k = len; k-- > 0;) {
        const uint64_t w = q4_pick(cur, (int)(k & 3));
        if ((k & 3) == 0 && k) cur = q[(k >> 2) - 1];
        const uint32_t w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
        const uint32_t op = FH_W_OP(w0), ro = FH_W_OUT(w0), ra = FH_W_A(w0), rb = w1;
        const bool is_choice = fh_is_choice(op);
        uint32_t c = FH_CHOICE_BOTH;
        if (is_choice) {
            ci--;
            if ((ci & 15) == 15 || ci == n_choices - 1) cw = chbits[(ci >> 4) * TL + lane16];
            c = (cw >> ((ci & 15) * 2)) & 3;
        }
        if (!act) continue;
        if (op == FH_OUTPUT) {
            const uint32_t na = use(ra);
            if (EMIT) *--dst = fh_pack(op, 0, na, 0, w1);
            count++;
            continue;
        }
        const uint32_t no = M(ro);
        if (no == DEAD) continue;  // value never used
        M(ro) = DEAD;
        // which operand survives a decided choice
        int alias = -1;  // old register aliased by `out`, or -1
        bool copy_imm = false;
        if (op == FH_COPY_REG) alias = (int)ra;
        else if (is_choice && c == FH_CHOICE_LEFT) alias = (int)ra;
        else if (is_choice && c == FH_CHOICE_RIGHT) {
            if (fh_is_rr(op)) alias = (int)rb; else copy_imm = true;
        }
        if (alias >= 0) {
            if (M(alias) == DEAD) { M(alias) = (uint8_t)no; continue; }  // hand the register over, no op
            if (EMIT) { pool.give(no); *--dst = fh_pack(FH_COPY_REG, no, M(alias), 0, 0); }
            count++;
            continue;
        }
        if (EMIT) pool.give(no);
        if (copy_imm) {
            if (EMIT) *--dst = fh_pack(FH_COPY_IMM, no, 0, 0, w1);
            count++;
            continue;
        }
        uint32_t na = 0, nb = 0;
        if (op != FH_INPUT && op != FH_COPY_IMM) na = use(ra);
        if (fh_is_rr(op)) nb = use(rb);
        if (is_choice) kept_choices++;
        if (EMIT) *--dst = fh_pack(op, no, na, nb, w1);
        count++;
    }
    out_len = (EMIT && pool.over) ? 0xFFFFFFFFu : count;      // ~0: more than 255 registers, the child keeps the parent tape
    out_regs = (uint32_t)pool.high;
    out_choices = kept_choices;
}

// Small tapes (the overwhelming majority below the root levels) run with a small LDS
// budget so that many waves fit per CU; the rest use the root tape's bounds.
#define SMALL_REGS 32u
#define SMALL_CHOICES 256u
FH_DEV bool tape_is_small(const FhTapeRef& t) { return t.n_regs <= SMALL_REGS && t.n_choices <= SMALL_CHOICES; }

// The tile kernel: interval-evaluate the children of one parent tile per wave, classify
// them, fill / discard the decided ones and emit pruned tapes + next-level work for the
// ambiguous ones.  Persistent workgroups pull parents from queue[level]; BIG selects the
// half of the queue whose tapes need the large LDS layout.
// Diagnostics: busy time of each wave (100 MHz wall clock) and units of work, per kernel kind
// k: stat[4k] = sum of busy ticks, [4k+1] = max, [4k+2] = waves that found work, [4k+3] = work units
struct WaveProbe {
    FhRenderState* S;
    int k;
    uint64_t t0;
    uint32_t units = 0;
    __device__ WaveProbe(FhRenderState* S_, int k_) : S(S_), k(k_), t0(wall_clock64()) {}
    __device__ void done(int lane) {
        if (lane != 0 || units == 0) return;
        const unsigned long long dt = wall_clock64() - t0;
        atomicAdd(&S->stat[4 * k], dt);
        atomicMax(&S->stat[4 * k + 1], dt);
        atomicAdd(&S->stat[4 * k + 2], 1ull);
        atomicAdd(&S->stat[4 * k + 3], (unsigned long long)units);
    }
};

template <bool IS3D, bool FULL, bool BIG, int TL>
__global__ void __launch_bounds__(WAVE) k_tiles(FhRenderState* S, int level) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const AS4 FhRender& P = *(const AS4 FhRender*)&S->P;
    const int lane = threadIdx.x;
    const int lane16 = lane & (TL - 1);
    const uint32_t max_regs = BIG ? P.max_regs : SMALL_REGS;
    const uint32_t max_choices = BIG ? P.max_choices : SMALL_CHOICES;
    char* const file = BIG ? big_file(S, smem) : smem;
    IV* regs = (IV*)file;                                                       // [max_regs][TL]
    uint32_t* chbits = (uint32_t*)(file + (size_t)max_regs * TL * sizeof(IV)); // [(max_choices+15)/16][TL]
    uint8_t* map = (uint8_t*)(chbits + (size_t)((max_choices + 15) / 16) * TL);  // [max_regs][TL]
    const uint32_t T = P.tiles[level];
    const bool last_level = (level + 1 == (int)P.n_levels);
    const uint32_t ntx = (P.width + T - 1) / T;
    const uint32_t* mind = S->mind[level];
    Mat4 mat;
#pragma unroll
    for (int i = 0; i < 16; i++) mat.m[i] = P.mat[i];
    WaveProbe probe(S, IS3D ? level : 7);

    // static round robin over the queued parents (an atomic cursor serialises at ~15 ns per parent)
    const uint32_t n_groups = BIG ? S->count_big[level] : S->count[level];
    for (uint32_t gi = blockIdx.x; gi < n_groups; gi += gridDim.x) {
        probe.units++;
        // small groups fill the queue from the front, big ones from the back
        const AS4 FhGroup& g = *(const AS4 FhGroup*)&S->queue[level][BIG ? S->qcap[level] - 1 - gi : gi];
        const ctape_t tape = (ctape_t)(S->arena + g.tape.off);
        const uint32_t len = g.tape.len, n_choices = g.tape.n_choices, n_regs = g.tape.n_regs;

        // the whole parent may have been occluded since it was queued (pre-pass levels)
        if (IS3D && level > 0) {
            const uint32_t Tp = P.tiles[level - 1], ntxp = (P.width + Tp - 1) / Tp;
            if (S->mind[level - 1][(g.y / Tp) * ntxp + g.x / Tp] >= g.z + Tp + 1) continue;
        }

        // ---- enumerate children ------------------------------------------------
        uint32_t nchild, cx, cy, cz;
        if (level == 0) {
            nchild = g.n;
            const uint32_t ri = g.first + lane16 * g.stride;  // root index, x-major (lib.rs:116-123)
            cx = (ri / P.roots_y) * T; cy = (ri % P.roots_y) * T; cz = g.z;
        } else {
            const uint32_t n = P.tiles[level - 1] / T;
            nchild = IS3D ? n * n * n : n * n;
            const uint32_t ci = lane16 % n, cj = (lane16 / n) % n, ck = IS3D ? lane16 / (n * n) : 0;
            cx = g.x + ci * T; cy = g.y + cj * T; cz = g.z + ck * T;
        }
        // tiles entirely outside the image never reach the output (the reference evaluates them
        // into its root-tile scratch and clips at the end, pixel.rs:478-490 / voxel.rs:529-533)
        bool act = lane < (int)nchild && cx < P.width && cy < P.height;
        const uint32_t fill_z = cz + T + 1;

        // ---- occlusion (voxel.rs:283-289; never changes results, only skips work): the
        // min-depth pyramid holds, per tile footprint, the smallest depth as of the last slab
        if (IS3D && act && mind[(cy / T) * ntx + cx / T] >= fill_z) act = false;
        if (ballot(act) == 0) continue;

        // ---- forward interval pass, trace packed 2 bits per choice ------------------
        IV X, Y, Z;
        {
            const IV sx = iv((float)cx, (float)cx + (float)T), sy = iv((float)cy, (float)cy + (float)T);
            const IV sz = IS3D ? iv((float)cz, (float)cz + (float)T) : iv(P.z, P.z);
            xf_interval(mat, sx, sy, sz, X, Y, Z);
        }
        Regs<IV, TL> R{regs, lane16};
        IV result = iv_nan();
        const uint64_t t_fwd0 = wall_clock64();
        uint32_t ci = 0, cw = 0;
        bool any_decided = false;
        if (lane < TL) {
            const AS4 Q4* q = (const AS4 Q4*)tape;
            Q4 cur = q[0];
            for (uint32_t k = 0; k < len; k++) {
                const uint64_t w = q4_pick(cur, (int)(k & 3));
                if ((k & 3) == 3) cur = q[(k >> 2) + 1];
                step<IVAL, TL, FULL>(
                    w, R,
                    [&](uint32_t slot) {
                        const uint32_t kd = P.in_kind[slot];
                        return kd == 0 ? X : (kd == 1 ? Y : (kd == 2 ? Z : iv1(P.in_value[slot])));
                    },
                    [&](uint32_t, IV v) { result = v; },
                    [&](int c) {
                        cw |= (uint32_t)c << ((ci & 15) * 2);
                        any_decided |= (c != FH_CHOICE_BOTH);
                        if ((ci & 15) == 15) { chbits[(ci >> 4) * TL + lane16] = cw; cw = 0; }
                        ci++;
                    });
            }
            if (ci & 15) chbits[(ci >> 4) * TL + lane16] = cw;
        }

        if (IS3D && lane == 0) { atomicAdd(&S->stat[32 + level], (unsigned long long)(wall_clock64() - t_fwd0)); atomicAdd(&S->stat[48 + level], (unsigned long long)len); }
        const uint64_t t_prune0 = wall_clock64();
        // ---- classify (voxel.rs:310-320, pixel.rs:345-368) -----------------------------
        const bool full = act && (IS3D || !P.pixel_perfect) && result.hi < 0.0f;
        const bool empty = act && (IS3D || !P.pixel_perfect) && !full && result.lo > 0.0f;
        const bool amb = act && !full && !empty;

        // fills, cooperatively over the 64 lanes
        {
            const uint64_t fullm = ballot(full);
            uint64_t fm = fullm | (IS3D ? 0ull : ballot(empty));
            while (fm) {
                const int c = __builtin_ctzll(fm);
                fm &= fm - 1;
                const uint32_t ccx = __shfl(cx, c, WAVE), ccy = __shfl(cy, c, WAVE);
                if (IS3D) {
                    const uint64_t v = (uint64_t)__shfl(fill_z, c, WAVE) << 32;
                    for (uint32_t p = lane; p < T * T; p += WAVE) {
                        const uint32_t x = ccx + (p % T), y = ccy + (p / T);
                        if (x < P.width && y < P.height) atomicMax((unsigned long long*)&S->zbuf[(size_t)y * P.width + x], (unsigned long long)v);
                    }
                } else {
                    const bool inside = (fullm >> c) & 1;
                    const float f = u2f(0x7FC00000u | (P.tag[level] << 1) | (inside ? 1u : 0u) | (0xF6u << 9));
                    for (uint32_t p = lane; p < T * T; p += WAVE) {
                        const uint32_t x = ccx + (p % T), y = ccy + (p / T);
                        if (x < P.width && y < P.height) S->image2d[(size_t)y * P.width + x] = f;
                    }
                }
            }
        }
        if (ballot(amb) == 0) continue;

        // ---- prune the tape for every ambiguous child whose trace decided something ----
        const bool prune = amb && any_decided;
        FhTapeRef child;
        child.off = g.tape.off; child.len = len; child.n_regs = (uint16_t)n_regs; child.n_choices = (uint16_t)n_choices;
        if (ballot(prune)) {
            // One reverse sweep: every pruned child reserves a slot as long as its parent and
            // writes its ops back to front from the slot's end, so no counting pass is needed
            // (the unused head of the slot is arena slack, recycled at the next slab).
            uint32_t clen = 0, cregs = 0, cch = 0;
            const uint32_t nprune = (uint32_t)__popcll(ballot(prune));
            const uint32_t rank = (uint32_t)__popcll(ballot(prune) & ((1ull << lane) - 1));
            uint32_t base = 0;
            if (lane == 0) base = arena_reserve(S, nprune * len);
            base = uni(base);
            if (base != 0xFFFFFFFFu) {
                if (lane < TL) {
                    for (uint32_t r = 0; r < n_regs; r++) map[r * TL + lane16] = DEAD;
                    uint64_t* dst = S->arena + base + (rank + 1) * len;
                    prune_sweep<true, TL>(tape, len, n_choices, chbits, map, lane16, prune, dst, clen, cregs, cch);
                }
                if (prune && clen != 0xFFFFFFFFu) {
                    child.off = base + (rank + 1) * len - clen; child.len = clen;
                    child.n_regs = (uint16_t)cregs; child.n_choices = (uint16_t)cch;
                }
                if (IS3D) {  // ops written, for the algorithmic-bytes accounting
                    uint32_t wsum;
                    wave_excl_sum(prune ? clen : 0u, wsum);
                    if (lane == 0) atomicAdd(&S->stat[56 + level], (unsigned long long)wsum);
                }
            } else if (lane == 0) {
                atomicAdd(&S->arena_overflow, 1u);  // children fall back to the parent tape
            }
        }

        if (IS3D && lane == 0) atomicAdd(&S->stat[40 + level], (unsigned long long)(wall_clock64() - t_prune0));

        // ---- hand ambiguous children to the next stage ------------------------------------
        const bool small = tape_is_small(child);
        if (!last_level) {
            uint32_t ns, nb;
            const uint32_t slot_s = wave_excl_sum((amb && small) ? 1u : 0u, ns);
            const uint32_t slot_b = wave_excl_sum((amb && !small) ? 1u : 0u, nb);
            // the last pre-pass level parks its output in the queue of the children's z-slab
            const bool park = IS3D && S->pre_levels > 0 && (uint32_t)(level + 1) == S->pre_levels;
            const uint32_t slab = park ? g.z / P.slab : 0;
            FhGroup* const qdst = park ? S->squeue + (size_t)slab * S->squeue_cap : S->queue[level + 1];
            const uint32_t qcap = park ? S->squeue_cap : S->qcap[level + 1];
            uint32_t qs = 0, qb = 0;
            if (lane == 0) {
                if (ns) qs = atomicAdd(park ? &S->scount[slab] : &S->count[level + 1], ns);
                if (nb) qb = atomicAdd(park ? &S->scount_big[slab] : &S->count_big[level + 1], nb);
            }
            qs = uni(qs); qb = uni(qb);
            if (amb) {
                FhGroup o;
                o.tape = child; o.x = cx; o.y = cy; o.z = cz; o.first = 0; o.n = 0; o.stride = 0;
                // the two halves cannot collide: their total is bounded by the capacity
                if (small) qdst[qs + slot_s] = o;
                else qdst[qcap - 1 - (qb + slot_b)] = o;
            }
        } else {
            uint32_t namb;
            const uint32_t slot = wave_excl_sum(amb ? 1u : 0u, namb);
            uint32_t lb = 0;
            if (lane == 0) lb = atomicAdd(&S->n_leaves, namb);
            lb = uni(lb);
            if (amb && lb + slot < S->leaf_cap) {
                FhLeaf lf;
                lf.tape = child; lf.x = cx; lf.y = cy; lf.z = cz;
                S->leaves[lb + slot] = lf;
                if (child.n_regs > S->norm_asm_regs) S->rare_seen = 1u;
                if (child.n_regs > S->leaf_asm_regs) atomicAdd(&S->n_leaves_lds, 1u);  // rare: lets k_leaves3d<2> return at once otherwise
                if (IS3D) {
                    S->leaf_table[(size_t)((cz % P.slab) / T) * (ntx * ((P.height + T - 1) / T)) + (size_t)(cy / T) * ntx + cx / T] =
                        FhLeafRef{lb + slot + 1, child.off, child.len | (min((uint32_t)child.n_regs, 255u) << 24), cx | (cy << 16)};  // [layer][footprint]
                }
            } else if (amb) atomicAdd(&S->queue_overflow, 1u);
        }
    }
    probe.done(lane);
}

// Which queued parents have a tape that reads no input changing along z (`depmask`: bit per input slot, from the camera
// matrix - capi.hip)?  Their children repeat along z (tsetup_body / tpush_body).  One wave per queue entry, the tape scanned
// 64 ops at a time; parked = 0: the queue of `level`; 1: the per-slab parking queues of the first per-slab level.
// Second slab context for the two-stream pipeline over the z-slabs: a copy of the state after the
// pre-pass with its own leaves, leaf table and footprint lists and the upper half of the free arena
struct FhFork {
    uint32_t n, mark;       // contexts (0: nothing to do); also k_mark_frame's work
    FhLeaf* leaves; FhLeafRef* leaf_table; uint32_t* fp_lists;
    size_t leaf_cap, n_footprints, hit_words;
};
FH_DEV void fork_state_body(FhRenderState* A, const FhFork& f) {
    if (f.mark) A->arena_frame_end = min(A->arena_head, A->arena_cap);      // (k_mark_frame's work in the same launch: one kernel boundary less on the coarse chain)
    // contexts A[1] .. A[n-1]: copies of A[0] with their own leaves, leaf table, footprint lists and 1/n of the free arena
    const uint32_t n = f.n;
    const uint32_t lo = min(A->pre_levels ? A->arena_frame_end : A->arena_root_end, A->arena_cap);
    const uint32_t part = (A->arena_cap - lo) / n;
    for (uint32_t k = 1; k < n; k++) {
        FhRenderState* B = A + k;
        *B = *A;
        B->leaves = f.leaves + (k - 1) * f.leaf_cap; B->leaf_table = f.leaf_table + (k - 1) * f.leaf_cap;
        uint32_t* fp = f.fp_lists + (k - 1) * (3 * f.n_footprints + f.hit_words);
        B->fp_list[0] = fp; B->fp_list[1] = fp + f.n_footprints; B->fp_list[2] = fp + 2 * f.n_footprints; B->hit_list = fp + 3 * f.n_footprints;
        B->arena_frame_end = lo + k * part; B->arena_root_end = lo + k * part;
        B->arena_cap = lo + (k + 1) * part;
    }
    A->arena_cap = lo + part;
}
__global__ void k_fork_state(FhRenderState* A, FhFork f) { fork_state_body(A, f); }

// (fork.n != 0: the launch's last block forks the slab contexts - k_fork_state's work, which touches nothing this kernel reads or writes: the
// state's arena bounds and pointers against the queue entries' flags - one launch less on the frame's coarse chain)
__global__ void __launch_bounds__(WAVE) k_tape_flags(FhRenderState* S, int level, uint32_t depmask, int parked, FhFork fork) {
    const uint32_t n_blocks = gridDim.x - (fork.n ? 1u : 0u);
    if (blockIdx.x >= n_blocks) {
        if (threadIdx.x == 0) fork_state_body(S, fork);
        return;
    }
    const int lane = threadIdx.x;
    const uint32_t nq = parked ? S->n_slabs : 1u;
    for (uint32_t q = 0; q < nq; q++) {
        FhGroup* const base = parked ? S->squeue + (size_t)q * S->squeue_cap : S->queue[level];
        const uint32_t cap = parked ? S->squeue_cap : S->qcap[level];
        const uint32_t ns = parked ? S->scount[q] : S->count[level], nb = parked ? S->scount_big[q] : S->count_big[level];
        for (uint32_t gi = blockIdx.x; gi < ns + nb; gi += n_blocks) {
            FhGroup& g = base[gi < ns ? gi : cap - 1 - (gi - ns)];
            const ctape_t tape = (ctape_t)(S->arena + g.tape.off);
            const uint32_t len = uni(g.tape.len);
            bool dep = false;
            for (uint32_t k = lane; k < len; k += WAVE) {
                const uint64_t w = tape[k];
                if ((uint32_t)(w & 0xFFu) == FH_INPUT && ((depmask >> ((uint32_t)(w >> 32) & 31u)) & 1u)) dep = true;
            }
            const bool any = ballot(dep) != 0;
            if (lane == 0) g.stride = any ? 0u : 0x80000000u;
        }
    }
}

// ======================================================================================
// Split 3D tile stage (64 children per parent): k_tsetup3d -> evaluate + prune (fh_tiles in
// assembly, or k_teval3d below for tapes outside its opcode set) -> k_tpush3d.  Same
// algorithm as k_tiles; the parent travels between the kernels in an FhSlot.
// ======================================================================================
// `split`: the children of ONE parent spread over up to `split` slots (0: chosen from the number of parents; 1: off).  The
// pre-pass levels below the root have a few hundred parents with tapes of hundreds of ops, one wave each, on a machine of 1024
// SIMDs, and the level takes as long as its slowest parent: forward pass + a lockstep prune whose cost grows with the
// number of children that keep different ops.  Several waves per parent, each evaluating the parent's tape for ALL lanes
// but classifying and pruning only its share of the children, repeat the forward pass on SIMDs that would be idle and
// shorten the prune.  Slots are independent for everything downstream (the push stage sees F small parents).
template <bool IS3D>
FH_DEV void tsetup_body(FhRenderState* S, int level) {
    const AS4 FhRender& P = *(const AS4 FhRender*)&S->P;
    const int lane = threadIdx.x;
    const uint32_t T = P.tiles[level];
    const uint32_t ntx = (P.width + T - 1) / T;
    const uint32_t* mind = S->mind[level];
    Mat4 mat;
#pragma unroll
    for (int i = 0; i < 16; i++) mat.m[i] = P.mat[i];
    // Parents map to slots one to one and waves to parents round robin: no atomic cursors (they
    // serialise in L2 and cost more than this kernel's work)
    // Tape parallelism at level 0: every block of root tiles gets one slot per independent tape
    // group (slots g0 .. g0 + G - 1, same children, different tapes); k_ttop3d merges them.
    const uint32_t TG = (level == 0 && S->n_tgroups) ? S->n_tgroups : 1;
    // (sharing a level-1 parent's children out over 2 / 4 / 8 slots was measured in round 3 - the kernel 412 -> 377 / 371 / 453 us, the frame
    // unchanged: the lockstep prune's cost is the ops SOME child keeps, and neighbours keep much the same - and taken out)
    const uint32_t G = TG;      // slots per parent (tape groups at level 0)
    const uint32_t ns = min(S->count[level], S->slot_cap[0] / G), nb = min(S->count_big[level], S->slot_cap[1] / G);
    if (blockIdx.x == 0 && lane == 0) {
        S->n_slots[0][level] = ns * G; S->n_slots[1][level] = nb * G;
        if (IS3D && nb && S->pre_levels > 0 && (uint32_t)level >= S->pre_levels) S->rare_seen = 1u;      // (a per-slab level's parent outside the small list)
    }
    // (root level with tape groups: a parent has one slot per group - 32 for prospero.vm - and a frame of few root tiles is a handful of
    // parents: the slots of a parent are shared out over `parts` waves, each repeating the little arithmetic and writing its slots)
    const uint32_t parts = TG > 1 ? max(1u, min(TG, gridDim.x / max(ns + nb, 1u))) : 1u;
    for (uint32_t wi = blockIdx.x; wi < (ns + nb) * parts; wi += gridDim.x) {
        const uint32_t gi = wi / parts, part = wi % parts;
        const bool big = gi >= ns;
        const AS4 FhGroup& g = *(const AS4 FhGroup*)&S->queue[level][big ? S->qcap[level] - 1 - (gi - ns) : gi];
        FhSlot* const slg = &S->slots[big ? 1 : 0][(big ? (gi - ns) : gi) * G];
        FhSlot& sl = *slg;
        // Column-invariant parents (k_tape_flags: the tape reads no input that changes along z, axis-aligned camera): the
        // children of a z-layer are those of every other layer, and the copies of the parent stacked along z (g.n tiles one
        // tile apart, this one included: the push stage queued one entry for them) are the same tile again - only the first
        // layer is evaluated, the push stage hands its results to all the instances.  Occlusion is tested for the instance
        // nearest the camera.
        const uint32_t zrep = (IS3D && level > 0 && g.n > 1) ? g.n : 1u;
        const bool inv = IS3D && level > 0 && ((g.stride >> 31) != 0 || zrep > 1);   // (a copy-carrying entry's tape is invariant: pruning only removes ops)
        if (IS3D && level > 0) {  // the whole parent may have been occluded since it was queued
            const uint32_t Tp = P.tiles[level - 1], ntxp = (P.width + Tp - 1) / Tp;
            if (S->mind[level - 1][(g.y / Tp) * ntxp + g.x / Tp] >= g.z + (zrep - 1) * Tp + Tp + 1) { if (lane < (int)G && part == 0) slg[lane].act = 0; continue; }
        }
        uint32_t nchild, cx, cy, cz;
        // (level 0 with a column-invariant root tape, capi_render.hpp root_zrep: the group's root tiles stand for g.x layers stacked on them)
        const uint32_t copies0 = (IS3D && level == 0 && g.x > 1) ? g.x : 1u;
        if (level == 0) {
            nchild = g.n;
            const uint32_t ri = g.first + lane * g.stride;  // root index, x-major (lib.rs:116-123)
            cx = (ri / P.roots_y) * T; cy = (ri % P.roots_y) * T; cz = g.z;
        } else {
            const uint32_t n = P.tiles[level - 1] / T;
            nchild = IS3D ? n * n * n : n * n;
            cx = g.x + (lane % n) * T; cy = g.y + ((lane / n) % n) * T; cz = IS3D ? g.z + (lane / (n * n)) * T : 0u;
        }
        bool act = lane < (int)nchild && cx < P.width && cy < P.height;
        uint32_t cz_top = cz + (copies0 - 1) * T;       // z of the instance nearest the camera among those this lane stands for
        if (inv) {
            const uint32_t n = P.tiles[level - 1] / T;
            if (lane / (n * n) != 0) act = false;
            cz_top = g.z + (zrep - 1) * P.tiles[level - 1] + (n - 1) * T;
        }
        if (IS3D && act && mind[(cy / T) * ntx + cx / T] >= cz_top + T + 1) act = false;  // voxel.rs:283-289
        const uint64_t actm = ballot(act);
        if (actm == 0) { if (lane < (int)G && part == 0) slg[lane].act = 0; continue; }
        IV X, Y, Z;
        xf_interval(mat, iv((float)cx, (float)cx + (float)T), iv((float)cy, (float)cy + (float)T),
                    IS3D ? iv((float)cz, (float)cz + (float)T) : iv(P.z, P.z), X, Y, Z);     // (2D: pixel.rs:325-333)
        for (uint32_t k = part; k < G; k += parts) {
            FhSlot& so = slg[k];
            const uint64_t share = actm;
            if (lane == 0) {
                const FhTapeRef tr = (TG > 1) ? S->tgroup[k] : FhTapeRef{g.tape.off, g.tape.len, g.tape.n_regs, g.tape.n_choices};
                so.tape = tr;
                so.level = (uint32_t)level | (copies0 << 8); so.act = share; so.base = 0; so.overflow = 0;     // (bits 8..: layers a root tile stands for)
                // (levels >= 1 have no term values: the field carries inv | zrep << 8 to the push stage)
                so.tvals = (TG > 1) ? S->tvals + (size_t)(gi - ns) * S->n_terms * WAVE * 2 : (float*)(uintptr_t)((inv ? 1u : 0u) | (zrep << 8));
            }
            if (share == 0) continue;
            so.xyz[0][lane] = X.lo; so.xyz[1][lane] = X.hi; so.xyz[2][lane] = Y.lo; so.xyz[3][lane] = Y.hi;
            so.xyz[4][lane] = Z.lo; so.xyz[5][lane] = Z.hi;
            so.corner[0][lane] = cx; so.corner[1][lane] = cy; so.corner[2][lane] = cz;
        }
    }
}

__global__ void __launch_bounds__(WAVE) k_tsetup3d(FhRenderState* S, int level) { tsetup_body<true>(S, level); }
// ... and of the 2D renderer when its tile sizes give 64 children per parent (the 128 / 16 hint): same slots, same
// evaluate + prune kernels; children are the n x n sub-tiles at the slice height z
__global__ void __launch_bounds__(WAVE) k_tsetup2d(FhRenderState* S, int level) { tsetup_body<false>(S, level); }

// Tape parallelism at level 0 (host_graph.hpp plan_terms), after the groups' forward passes left the
// terms' intervals in S->tvals: the root min / max tree over those terms, with the Choice every op
// of the tree records (vm/mod.rs:436-471 via iv_min / iv_max) in topch[block][child][op].
//
// The usual tree is a chain, acc = op(acc, term): its accumulator is a prefix min / max of the terms
// (associative and exact, NaN absorbing), so one wave per child scans 64 ops at a time.
// grid (64 children, blocks), lane = op within the chunk.
FH_DEV IV top_comb(bool is_min, IV a, IV b) {
    if (iv_has_nan(a) || iv_has_nan(b)) return iv_nan();
    return is_min ? iv(rmin(a.lo, b.lo), rmin(a.hi, b.hi)) : iv(rmax(a.lo, b.lo), rmax(a.hi, b.hi));
}
// The scan, one wave per child.  Lane l takes the contiguous segment of ceil(n_top / 64) ops starting at l * that: it folds its
// segment (the loads of its terms are independent and issued together), the 64 segment totals are scanned across the lanes once, and the
// lane walks its segment again with the accumulator that reaches it, recording the choices: per + 6 + per dependent steps (28 for
// prospero's 664 ops) where scanning chunk after chunk of 64 ops took 11 x (6 cross-lane steps + a round of loads).  min / max of the
// bounds with NaN absorbing is associative and exact: any bracketing gives the chunked scan's accumulators bit for bit.
#define FH_CHAIN_SEG 16       // ops per lane held in registers: chains of up to 1 024 ops (longer ones: k_tchain3d_chunks below)
__global__ void __launch_bounds__(WAVE) k_tchain3d(FhRenderState* S) {
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x, b = blockIdx.y, G = S->n_tgroups, n_top = S->n_top;
    if (b >= S->n_slots[1][0] / G) return;
    FhSlot& p = S->slots[1][(size_t)b * G];
    if (((p.act >> c) & 1) == 0) return;
    const IV* const tv = (const IV*)S->tvals + (size_t)b * S->n_terms * WAVE;
    uint8_t* const tc = S->topch + ((size_t)b * WAVE + c) * n_top;
    const uint32_t* const top = (const uint32_t*)S->ttop;
    const bool is_min = (top[0] & 0xFF) == FH_MIN_RR || (top[0] & 0xFF) == FH_MIN_RI;
    const float ident = is_min ? u2f(0x7f800000u) : u2f(0xff800000u);
    const uint32_t per = (n_top + WAVE - 1) / WAVE, j0 = (uint32_t)lane * per;
    IV e[FH_CHAIN_SEG];
#pragma unroll
    for (int k = 0; k < FH_CHAIN_SEG; k++) {
        const uint32_t j = j0 + k;
        e[k] = iv(ident, ident);
        if ((uint32_t)k < per && j < n_top) {
            const uint32_t bk = top[3 * j] >> 24, bv = top[3 * j + 2];
            e[k] = bk == 1 ? tv[(size_t)bv * WAVE + c] : iv1(u2f(bv));
        }
    }
    IV incl = e[0];
#pragma unroll
    for (int k = 1; k < FH_CHAIN_SEG; k++) incl = top_comb(is_min, incl, e[k]);      // (past the segment's end: the identity)
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const IV o = iv(__shfl_up(incl.lo, d, WAVE), __shfl_up(incl.hi, d, WAVE));
        if (lane >= d) incl = top_comb(is_min, o, incl);
    }
    const IV start = tv[(size_t)top[1] * WAVE + c];   // the chain starts from the first op's `a`, a term
    IV before = iv(__shfl_up(incl.lo, 1, WAVE), __shfl_up(incl.hi, 1, WAVE));
    before = lane == 0 ? start : top_comb(is_min, start, before);
#pragma unroll
    for (int k = 0; k < FH_CHAIN_SEG; k++) {
        int ch;
        if (is_min) (void)iv_min(before, e[k], ch); else (void)iv_max(before, e[k], ch);
        if ((uint32_t)k < per && j0 + k < n_top) tc[j0 + k] = (uint8_t)ch;
        before = top_comb(is_min, before, e[k]);
    }
    if (lane == WAVE - 1) { p.res[0][c] = before.lo; p.res[1][c] = before.hi; }
}
// (chains of more than 64 x FH_CHAIN_SEG ops: chunk after chunk of 64 ops, a cross-lane scan each)
__global__ void __launch_bounds__(WAVE) k_tchain3d_chunks(FhRenderState* S) {
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x, b = blockIdx.y, G = S->n_tgroups, n_top = S->n_top;
    if (b >= S->n_slots[1][0] / G) return;
    FhSlot& p = S->slots[1][(size_t)b * G];
    if (((p.act >> c) & 1) == 0) return;
    const IV* const tv = (const IV*)S->tvals + (size_t)b * S->n_terms * WAVE;
    uint8_t* const tc = S->topch + ((size_t)b * WAVE + c) * n_top;
    const uint32_t* const top = (const uint32_t*)S->ttop;
    const bool is_min = (top[0] & 0xFF) == FH_MIN_RR || (top[0] & 0xFF) == FH_MIN_RI;
    const float ident = is_min ? u2f(0x7f800000u) : u2f(0xff800000u);
    IV carry = tv[(size_t)top[1] * WAVE + c];   // the chain starts from the first op's `a`, a term
    for (uint32_t j0 = 0; j0 < n_top; j0 += WAVE) {
        const uint32_t j = j0 + lane;
        const bool valid = j < n_top;
        IV e = iv(ident, ident);
        if (valid) {
            const uint32_t bk = top[3 * j] >> 24, bv = top[3 * j + 2];
            e = bk == 1 ? tv[(size_t)bv * WAVE + c] : iv1(u2f(bv));
        }
        IV incl = e;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const IV o = iv(__shfl_up(incl.lo, d, WAVE), __shfl_up(incl.hi, d, WAVE));
            if (lane >= d) incl = top_comb(is_min, o, incl);
        }
        IV before = iv(__shfl_up(incl.lo, 1, WAVE), __shfl_up(incl.hi, 1, WAVE));
        before = lane == 0 ? carry : top_comb(is_min, carry, before);
        int ch;
        if (is_min) (void)iv_min(before, e, ch); else (void)iv_max(before, e, ch);
        if (valid) tc[j] = (uint8_t)ch;
        carry = top_comb(is_min, carry, iv(__shfl(incl.lo, WAVE - 1, WAVE), __shfl(incl.hi, WAVE - 1, WAVE)));
    }
    if (lane == 0) { p.res[0][c] = carry.lo; p.res[1][c] = carry.hi; }
}

// Any other tree: op by op, one child per lane (registers of the tree in LDS).
#define FH_TOP_REGS 16
__global__ void __launch_bounds__(WAVE) k_ttop3d(FhRenderState* S) {
    __shared__ IV regs[FH_TOP_REGS][WAVE];
    const int lane = threadIdx.x;
    const uint32_t G = S->n_tgroups, n_top = S->n_top, n_terms = S->n_terms;
    const uint32_t nblk = S->n_slots[1][0] / G;
    const AS4 uint32_t* const top = (const AS4 uint32_t*)S->ttop;   // 3 words per op
    for (uint32_t b = blockIdx.x; b < nblk; b += gridDim.x) {
        FhSlot& p = S->slots[1][(size_t)b * G];
        if (p.act == 0) continue;
        const IV* const tv = (const IV*)S->tvals + (size_t)b * n_terms * WAVE;
        uint8_t* const tc = S->topch + ((size_t)b * WAVE + lane) * n_top;
        IV r = iv_nan();
        for (uint32_t j0 = 0; j0 < n_top; j0 += 8) {  // the terms of 8 ops are fetched together (independent loads)
            FhTopOp o[8];
            IV ta[8], tb[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t j = min(j0 + u, n_top - 1), w0 = top[3 * j];
                o[u].op = (uint8_t)w0; o[u].out = (uint8_t)(w0 >> 8); o[u].a_kind = (uint8_t)(w0 >> 16); o[u].b_kind = (uint8_t)(w0 >> 24);
                o[u].a = top[3 * j + 1]; o[u].b = top[3 * j + 2];
                ta[u] = o[u].a_kind == 1 ? tv[(size_t)o[u].a * WAVE + lane] : iv_nan();
                tb[u] = o[u].b_kind == 1 ? tv[(size_t)o[u].b * WAVE + lane] : iv1(__uint_as_float(o[u].b));
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (j0 + u >= n_top) break;
                const IV a = o[u].a_kind == 0 ? regs[o[u].a][lane] : ta[u];
                const IV bb = o[u].b_kind == 0 ? regs[o[u].b][lane] : tb[u];
                int c;
                r = (o[u].op == FH_MIN_RR || o[u].op == FH_MIN_RI) ? iv_min(a, bb, c) : iv_max(a, bb, c);
                regs[o[u].out][lane] = r;
                tc[j0 + u] = (uint8_t)c;
            }
        }
        p.res[0][lane] = r.lo; p.res[1][lane] = r.hi;
    }
}

// The tree's result is the root tape's result: ambiguous children reserve arena space for their
// pruned tape and are marked for fh_prune1, exactly as fh_tiles does in export mode.
FH_DEV void tmark_body(FhRenderState* S, uint32_t first, uint32_t stride) {
    const int lane = threadIdx.x;
    const uint32_t G = S->n_tgroups;
    const uint32_t nblk = S->n_slots[1][0] / G;
    const FhTapeRef root = FhTapeRef{0, S->troot_len, (uint16_t)S->troot_regs, (uint16_t)S->troot_choices};
    for (uint32_t b = first; b < nblk; b += stride) {
        FhSlot* const slg = &S->slots[1][(size_t)b * G];
        const uint64_t actm = slg[0].act;
        if (lane >= 1 && lane < (int)G) slg[lane].act = 0;   // only the primary slot is pushed
        if (actm == 0) continue;
        const bool act = (actm >> lane) & 1;
        FhSlot& p = slg[0];
        const IV r = iv(p.res[0][lane], p.res[1][lane]);
        const bool full = act && r.hi < 0.0f, empty = act && !full && r.lo > 0.0f;
        const bool amb = act && !full && !empty;
        const uint64_t am = ballot(amb);
        const uint32_t n = (uint32_t)__popcll(am), rank = (uint32_t)__popcll(am & ((1ull << lane) - 1));
        uint32_t base = 0;
        if (lane == 0 && n) base = arena_reserve(S, n * root.len);
        base = uni(base);
        const bool ok = n && base != 0xFFFFFFFFu;
        if (n && !ok && lane == 0) atomicAdd(&S->arena_overflow, 1u);  // the children keep the root tape
        p.c_off[lane] = (amb && ok) ? base + (rank + 1) * root.len : root.off;   // end of this child's arena slot
        p.c_len[lane] = (amb && ok) ? 0xFFFFFFFFu : root.len;                    // ~0: to be written by fh_prune1
        p.c_rc[lane] = (uint32_t)root.n_regs | ((uint32_t)root.n_choices << 16);
        if (lane == 0) { p.tape = root; if (ok) p.base = base; else if (n) p.overflow = 1; }
    }
}

__global__ void __launch_bounds__(WAVE) k_tmark3d(FhRenderState* S) { tmark_body(S, blockIdx.x, gridDim.x); }

// ... and the choice words of the root tape, gathered from the groups' traces and the tree's:
// grid (word, block), lane = child; the blocks with word == root_words are the marks above (they read the primary slots' results and
// write what the gather does not read: one launch for both)
__global__ void __launch_bounds__(WAVE) k_tscatter3d(FhRenderState* S, uint32_t group_words, uint32_t root_words) {
    const int lane = threadIdx.x;
    if (blockIdx.x == root_words) { tmark_body(S, blockIdx.y, gridDim.y); return; }
    const uint32_t w = blockIdx.x, b = blockIdx.y, G = S->n_tgroups;
    if (b >= S->n_slots[1][0] / G) return;
    if (S->slots[1][(size_t)b * G].act == 0) return;
    const uint32_t nch = S->troot_choices;
    const AS4 uint32_t* const src = (const AS4 uint32_t*)S->chsrc;
    const uint8_t* const tc = S->topch + ((size_t)b * WAVE + lane) * S->n_top;
    const uint32_t* const gw = S->chw[1] + (size_t)b * G * group_words * WAVE;
    uint32_t out = 0;
    for (uint32_t i = 0; i < 16 && w * 16 + i < nch; i++) {
        const uint32_t e = src[w * 16 + i], g = e >> 24, j = e & 0xFFFFFFu;
        uint32_t c;
        if (g == 255) c = tc[j];
        else c = (gw[((size_t)g * group_words + (j >> 4)) * WAVE + lane] >> ((j & 15) * 2)) & 3u;
        out |= c << (2 * i);
    }
    S->chwr[((size_t)b * G * root_words + w) * WAVE + lane] = out;
}

// Step 2 in C++ (reference implementation of fh_tiles; used for tapes outside the assembly
// opcode set): forward interval pass, then one reverse prune sweep per ambiguous child.
template <bool FULL, bool BIG>
FH_DEV void teval_slots(FhRenderState* S, int level, char* smem, uint32_t first, uint32_t stride) {
    constexpr int TL = 64;
    const AS4 FhRender& P = *(const AS4 FhRender*)&S->P;
    const int lane = threadIdx.x;
    const uint32_t max_regs = BIG ? P.max_regs : SMALL_REGS;
    const uint32_t max_choices = BIG ? P.max_choices : SMALL_CHOICES;
    IV* regs = (IV*)smem;
    uint32_t* chbits = (uint32_t*)(smem + (size_t)max_regs * TL * sizeof(IV));
    uint8_t* map = (uint8_t*)(chbits + (size_t)((max_choices + 15) / 16) * TL);
    const uint32_t n_slots = S->n_slots[BIG ? 1 : 0][level];
    for (uint32_t si = first; si < n_slots; si += stride) {
        FhSlot& sl = S->slots[BIG ? 1 : 0][si];
        if (uni((uint32_t)(sl.act != 0)) == 0) continue;
        const uint32_t off = uni(sl.tape.off), len = uni(sl.tape.len);
        const uint32_t n_choices = uni((uint32_t)sl.tape.n_choices), n_regs = uni((uint32_t)sl.tape.n_regs);
        const ctape_t tape = (ctape_t)(S->arena + off);
        const bool act = (sl.act >> lane) & 1;
        const IV X = iv(sl.xyz[0][lane], sl.xyz[1][lane]), Y = iv(sl.xyz[2][lane], sl.xyz[3][lane]),
                 Z = iv(sl.xyz[4][lane], sl.xyz[5][lane]);
        Regs<IV, TL> R{regs, lane};
        IV result = iv_nan();
        uint32_t ci = 0, cw = 0;
        bool any_decided = false;
        {
            const AS4 Q4* q = (const AS4 Q4*)tape;
            Q4 cur = q[0];
            for (uint32_t k = 0; k < len; k++) {
                const uint64_t w = q4_pick(cur, (int)(k & 3));
                if ((k & 3) == 3) cur = q[(k >> 2) + 1];
                step<IVAL, TL, FULL>(
                    w, R,
                    [&](uint32_t slot) {
                        const uint32_t kd = P.in_kind[slot];
                        return kd == 0 ? X : (kd == 1 ? Y : (kd == 2 ? Z : iv1(P.in_value[slot])));
                    },
                    [&](uint32_t, IV v) { result = v; },
                    [&](int c) {
                        cw |= (uint32_t)c << ((ci & 15) * 2);
                        any_decided |= (c != FH_CHOICE_BOTH);
                        if ((ci & 15) == 15) { chbits[(ci >> 4) * TL + lane] = cw; cw = 0; }
                        ci++;
                    });
            }
            if (ci & 15) chbits[(ci >> 4) * TL + lane] = cw;
        }
        sl.res[0][lane] = result.lo; sl.res[1][lane] = result.hi;
        const bool full = act && result.hi < 0.0f, empty = act && !full && result.lo > 0.0f;
        const bool amb = act && !full && !empty;
        const bool prune = amb && any_decided;
        uint32_t coff = off, clen = len, cregs = n_regs, cch = n_choices;
        const uint64_t pm = ballot(prune);
        if (pm) {
            const uint32_t nprune = (uint32_t)__popcll(pm);
            const uint32_t rank = (uint32_t)__popcll(pm & ((1ull << lane) - 1));
            uint32_t base = 0;
            if (lane == 0) base = arena_reserve(S, nprune * len);
            base = uni(base);
            if (base != 0xFFFFFFFFu) {
                for (uint32_t r = 0; r < n_regs; r++) map[r * TL + lane] = DEAD;
                uint32_t l2 = 0, r2 = 0, c2 = 0;
                prune_sweep<true, TL>(tape, len, n_choices, chbits, map, lane, prune, S->arena + base + (rank + 1) * len, l2, r2, c2);
                if (prune && l2 != 0xFFFFFFFFu) { coff = base + (rank + 1) * len - l2; clen = l2; cregs = r2; cch = c2; }
                if (lane == 0) sl.base = base;
            } else if (lane == 0) {
                sl.overflow = 1;
                atomicAdd(&S->arena_overflow, 1u);  // children fall back to the parent tape
            }
        }
        sl.c_off[lane] = coff; sl.c_len[lane] = clen; sl.c_rc[lane] = cregs | (cch << 16);
    }
}
template <bool FULL, bool BIG>
__global__ void __launch_bounds__(WAVE) k_teval3d(FhRenderState* S, int level) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    teval_slots<FULL, BIG>(S, level, BIG ? big_file(S, smem) : smem, blockIdx.x, gridDim.x);
}

// Step 3: fills of the decided children, queue / leaf entries for the ambiguous ones
// which: 0 every slot, 1 the small list only, 2 the other list only (rare mode: k_tpush3d)
template <bool IS3D>
FH_DEV void tpush_body(FhRenderState* S, int level, uint32_t first, uint32_t stride, int which = 0) {
    const AS4 FhRender& P = *(const AS4 FhRender*)&S->P;
    const int lane = threadIdx.x;
    const uint32_t T = P.tiles[level];
    const uint32_t ntx = (P.width + T - 1) / T;
    const bool last_level = (level + 1 == (int)P.n_levels);
    const uint32_t n0 = S->n_slots[0][level], n1 = S->n_slots[1][level];
    const uint32_t s_lo = which == 2 ? n0 : 0u, s_hi = which == 1 ? n0 : n0 + n1;
    // Leaves: ONE reservation per wave for all its parents (a first pass counts them).  One atomic per
    // parent on the same counter serialises in L2, ~15 ns each: with thousands of parents that was
    // nearly all of this kernel's time.
    uint32_t leaf_base = 0;
    if (last_level) {
        uint32_t mine = 0;
        for (uint32_t si = s_lo + first; si < s_hi; si += stride) {
            const FhSlot& sl = si < n0 ? S->slots[0][si] : S->slots[1][si - n0];
            const uint64_t actm = sl.act;
            if (uni((uint32_t)(actm != 0)) == 0) continue;
            const float lo = sl.res[0][lane], hi = sl.res[1][lane];
            mine += (uint32_t)__popcll(ballot(((actm >> lane) & 1) && ((!IS3D && P.pixel_perfect) || (!(hi < 0.0f) && !(lo > 0.0f)))));
        }
        if (lane == 0 && mine) leaf_base = atomicAdd(&S->n_leaves, mine);
        leaf_base = uni(leaf_base);
    }
    for (uint32_t si = s_lo + first; si < s_hi; si += stride) {
        const FhSlot& sl = si < n0 ? S->slots[0][si] : S->slots[1][si - n0];
        if (uni((uint32_t)(sl.act != 0)) == 0) continue;
        const bool act = (sl.act >> lane) & 1;
        const float lo = sl.res[0][lane], hi = sl.res[1][lane];
        const uint32_t cx = sl.corner[0][lane], cy = sl.corner[1][lane], cz = sl.corner[2][lane];
        // column-invariant parent (tsetup_body): every active lane stands for `ninst` tiles stacked along z, one T apart
        // (root level: a tile of a column-invariant root tape stands for the layers stacked on it - tsetup_body copies0 - one T apart too)
        const uint32_t copies0 = (IS3D && level == 0) ? uni(sl.level >> 8) : 1u;
        const uint32_t zi = (IS3D && level > 0) ? uni((uint32_t)(uintptr_t)sl.tvals) : 0u;
        const bool inv = (zi & 1) != 0 || copies0 > 1;
        const uint32_t ninst = level == 0 ? copies0 : (inv ? (zi >> 8) * (P.tiles[level - 1] / T) : 1u);
        const bool fills = IS3D || !P.pixel_perfect;                                       // pixel.rs:345-368
        const bool full = act && fills && hi < 0.0f, empty = act && fills && !full && lo > 0.0f;  // voxel.rs:310-320
        const bool amb = act && !full && !empty;
        const uint64_t fullm = ballot(full);
        // (2D: k_tfill2d, one workgroup per decided tile - the root level's 128 x 128 fills by 16 waves took 1.9 ms; 3D above the leaf
        // level: the launch's fill waves, tfill3d_body - blockIdx.y > 0)
        uint64_t fm = (IS3D && (gridDim.y == 1 || which == 2)) ? fullm : 0ull;
        while (fm) {  // interval-full tiles write fill_z = corner_z + T + 1 (voxel.rs:283, 310-317)
            const int c = __builtin_ctzll(fm);
            fm &= fm - 1;
            const uint32_t ccx = __shfl(cx, c, WAVE), ccy = __shfl(cy, c, WAVE);
            if (IS3D) {
                // (a full child behind another full child of this parent - same x, y, smaller z - fills with the smaller number of an
                // atomic max: skipped.  A solid interior is stacks of four such children)
                if (ballot(full && cx == ccx && cy == ccy && cz > __shfl(cz, c, WAVE))) continue;
                // (of the instances stacked along z the nearest one's fill is the maximum)
                const uint64_t v = (uint64_t)(__shfl(cz, c, WAVE) + (ninst - 1) * T + T + 1) << 32;
                for (uint32_t p = lane; p < T * T; p += WAVE) {
                    const uint32_t x = ccx + (p % T), y = ccy + (p / T);
                    if (x < P.width && y < P.height) atomicMax((unsigned long long*)&S->zbuf[(size_t)y * P.width + x], (unsigned long long)v);
                }
            } else {
                const bool inside = (fullm >> c) & 1;
                const float f = u2f(0x7FC00000u | (P.tag[level] << 1) | (inside ? 1u : 0u) | (0xF6u << 9));   // pixel.rs:225-229
                for (uint32_t p = lane; p < T * T; p += WAVE) {
                    const uint32_t x = ccx + (p % T), y = ccy + (p / T);
                    if (x < P.width && y < P.height) S->image2d[(size_t)y * P.width + x] = f;
                }
            }
        }
        if (S->want_stats) {   // algorithmic-bytes accounting: tape ops read by this parent, ops of pruned tapes written
            uint32_t wsum;
            wave_excl_sum((amb && sl.c_off[lane] != sl.tape.off) ? sl.c_len[lane] : 0u, wsum);
            if (lane == 0) { atomicAdd(&S->stat[48 + level], (unsigned long long)sl.tape.len); atomicAdd(&S->stat[56 + level], (unsigned long long)wsum); }
        }
        const uint64_t am = ballot(amb);
        if (am == 0) continue;
        FhTapeRef child;
        child.off = sl.c_off[lane]; child.len = sl.c_len[lane];
        child.n_regs = (uint16_t)(sl.c_rc[lane] & 0xFFFFu); child.n_choices = (uint16_t)(sl.c_rc[lane] >> 16);
        const bool small = tape_is_small(child);
        if (!last_level) {
            uint32_t ns, nb;
            const uint32_t slot_s = wave_excl_sum((amb && small) ? 1u : 0u, ns);
            const uint32_t slot_b = wave_excl_sum((amb && !small) ? 1u : 0u, nb);
            // the last pre-pass level parks its output in the queue of the children's z-slab
            const bool park = IS3D && S->pre_levels > 0 && (uint32_t)(level + 1) == S->pre_levels;
            const uint32_t slab = park ? uni(__shfl(cz, __builtin_ctzll(am), WAVE)) / P.slab : 0;
            FhGroup* const qdst = park ? S->squeue + (size_t)slab * S->squeue_cap : S->queue[level + 1];
            const uint32_t qcap = park ? S->squeue_cap : S->qcap[level + 1];
            uint32_t qs = 0, qb = 0;
            if (lane == 0) {
                if (ns) qs = atomicAdd(park ? &S->scount[slab] : &S->count[level + 1], ns);
                if (nb) qb = atomicAdd(park ? &S->scount_big[slab] : &S->count_big[level + 1], nb);
            }
            qs = uni(qs); qb = uni(qb);
            if (amb) {
                FhGroup o;
                // (a child of a column-invariant parent is queued once for its ninst instances, one T apart: as a parent it is the
                // first of `n` copies one tile apart; stride: k_tape_flags)
                o.tape = child; o.x = cx; o.y = cy; o.z = cz; o.first = 0; o.n = inv ? ninst : 0u; o.stride = 0;
                if (small) qdst[qs + slot_s] = o;
                else qdst[qcap - 1 - (qb + slot_b)] = o;
            }
        } else {
            const uint32_t namb = (uint32_t)__popcll(am), slot = (uint32_t)__popcll(am & ((1ull << lane) - 1));
            // Leaves of a column-invariant parent: the ninst leaves stacked along z hold the same value in every voxel of a pixel's
            // column, so a pixel is hit in the nearest of them or in none - ONE leaf, the nearest, stands for the stack.
            if (S->want_stats) {    // algorithmic accounting of the leaf stage (render_state.h leaf_stat)
                const uint32_t passes = inv ? 1u : (child.n_regs <= 8 ? 1u : (child.n_regs <= 16 ? 2u : (child.n_regs <= 32 ? 4u : 8u)));
                uint32_t s1, s2;
                wave_excl_sum(amb ? child.len : 0u, s1);
                wave_excl_sum(amb ? child.len * passes : 0u, s2);
                if (lane == 0) {
                    atomicAdd(&S->leaf_stat[0], (unsigned long long)namb); atomicAdd(&S->leaf_stat[1], (unsigned long long)s1);
                    atomicAdd(&S->leaf_stat[2], (unsigned long long)s2); atomicAdd(&S->leaf_stat[3], (unsigned long long)s1 * (inv ? 64u : 512u));
                }
            }
            {
                const uint32_t lb = leaf_base, iz = cz + (ninst - 1) * T;
                leaf_base += namb;
                if (amb && lb + slot < S->leaf_cap) {
                    FhLeaf lf;
                    lf.tape = child; lf.x = cx; lf.y = cy; lf.z = iz;
                    S->leaves[lb + slot] = lf;
                    if (child.n_regs > S->norm_asm_regs) S->rare_seen = 1u;
                    if (child.n_regs > S->leaf_asm_regs) atomicAdd(&S->n_leaves_lds, 1u);  // rare: lets k_leaves3d<2> return at once otherwise
                    if (IS3D)
                        S->leaf_table[(size_t)((iz % P.slab) / T) * (ntx * ((P.height + T - 1) / T)) + (size_t)(cy / T) * ntx + cx / T] =
                            FhLeafRef{lb + slot + 1, child.off, child.len | (min((uint32_t)child.n_regs, 255u) << 24), cx | (cy << 16)};  // [layer][footprint]
                } else if (amb) atomicAdd(&S->queue_overflow, 1u);
            }
        }
    }
}
// 3D fills of the levels above the leaves by waves of their own (k_tpush3d's blockIdx.y = 1 .. parts): a parent's wave took its
// interval-full children one after the other - T x T atomics each, 1024 at T = 32 - and a model with a solid interior has most of its
// few dozen pre-pass parents full of them: bear.vm 512^3, 187 us of a 1 ms frame in ONE launch of 64 busy waves.  Here `parts` waves
// share a parent's children, and a full child behind another full child of the same parent (same x, y, smaller z: its fill is the
// smaller number of an atomic max) is skipped.
FH_DEV void tfill3d_body(FhRenderState* S, int level, uint32_t first, uint32_t stride, uint32_t part, uint32_t parts, bool small_only) {
    const AS4 FhRender& P = *(const AS4 FhRender*)&S->P;
    const int lane = threadIdx.x;
    const uint32_t T = P.tiles[level];
    const uint32_t n0 = S->n_slots[0][level], n1 = small_only ? 0u : S->n_slots[1][level];
    for (uint32_t si = first; si < n0 + n1; si += stride) {
        const FhSlot& sl = si < n0 ? S->slots[0][si] : S->slots[1][si - n0];
        if (uni((uint32_t)(sl.act != 0)) == 0) continue;
        const bool full = ((sl.act >> lane) & 1) && sl.res[1][lane] < 0.0f;   // voxel.rs:310-317
        uint64_t fm = ballot(full);
        if (fm == 0) continue;
        const uint32_t cx = sl.corner[0][lane], cy = sl.corner[1][lane], cz = sl.corner[2][lane];
        // (instances stacked along z by a column-invariant parent: tpush_body)
        const uint32_t copies0 = level == 0 ? uni(sl.level >> 8) : 1u;
        const uint32_t zi = level > 0 ? uni((uint32_t)(uintptr_t)sl.tvals) : 0u;
        const bool inv = (zi & 1) != 0 || copies0 > 1;
        const uint32_t ninst = level == 0 ? copies0 : (inv ? (zi >> 8) * (P.tiles[level - 1] / T) : 1u);
        while (fm) {
            const int c = __builtin_ctzll(fm);
            fm &= fm - 1;
            if ((uint32_t)c % parts != part) continue;
            const uint32_t ccx = __shfl(cx, c, WAVE), ccy = __shfl(cy, c, WAVE), ccz = __shfl(cz, c, WAVE);
            if (ballot(full && cx == ccx && cy == ccy && cz > ccz)) continue;
            const uint64_t v = (uint64_t)(ccz + (ninst - 1) * T + T + 1) << 32;   // fill_z = corner_z + T + 1 (voxel.rs:283), of the nearest instance
            for (uint32_t p = lane; p < T * T; p += WAVE) {
                const uint32_t x = ccx + (p % T), y = ccy + (p / T);
                if (x < P.width && y < P.height) atomicMax((unsigned long long*)&S->zbuf[(size_t)y * P.width + x], (unsigned long long)v);
            }
        }
    }
}
// Rare mode (capi_render.hpp): the last `rare_blocks` blocks are the level's launches for the parents outside the small slot list - none,
// nearly always: each evaluates and prunes its share of them in C++ (teval_slots, the interval file in HBM: `rare_stride` bytes per block from
// `rare_file`) and pushes their children itself; the other blocks keep to the small list.
__global__ void __launch_bounds__(WAVE) k_tpush3d(FhRenderState* S, int level, uint32_t rare_blocks, char* rare_file, uint32_t rare_stride) {
    const uint32_t nb = gridDim.x - rare_blocks;
    if (blockIdx.x >= nb) {
        if (blockIdx.y) return;
        const uint32_t e = blockIdx.x - nb;
        if (S->n_slots[1][level] == 0) return;
        teval_slots<true, true>(S, level, rare_file + (size_t)e * rare_stride, e, rare_blocks);
        __threadfence_block();
        tpush_body<true>(S, level, e, rare_blocks, 2);
        return;
    }
    if (blockIdx.y == 0) tpush_body<true>(S, level, blockIdx.x, nb, rare_blocks ? 1 : 0);
    else tfill3d_body(S, level, blockIdx.x, nb, blockIdx.y - 1, gridDim.y - 1, rare_blocks != 0);
}
__global__ void __launch_bounds__(WAVE) k_tpush2d(FhRenderState* S, int level) { tpush_body<false>(S, level, blockIdx.x, gridDim.x); }
// 2D fills (pixel.rs:345-368, 225-229): a tile whose interval is decided becomes a NaN-boxed fill carrying the level it was
// decided at and whether it is inside.  grid (64 children, slots of the level's upper bound), one workgroup per tile.
__global__ void __launch_bounds__(256) k_tfill2d(FhRenderState* S, int level) {
    const AS4 FhRender& P = *(const AS4 FhRender*)&S->P;
    if (P.pixel_perfect) return;
    const uint32_t c = blockIdx.x, si = blockIdx.y;
    const uint32_t n0 = S->n_slots[0][level], n1 = S->n_slots[1][level];
    if (si >= n0 + n1) return;
    const FhSlot& sl = si < n0 ? S->slots[0][si] : S->slots[1][si - n0];
    if (((sl.act >> c) & 1) == 0) return;
    const float lo = sl.res[0][c], hi = sl.res[1][c];
    const bool full = hi < 0.0f, empty = !full && lo > 0.0f;
    if (!full && !empty) return;
    const uint32_t T = P.tiles[level], cx = sl.corner[0][c], cy = sl.corner[1][c];
    const float f = u2f(0x7FC00000u | (P.tag[level] << 1) | (full ? 1u : 0u) | (0xF6u << 9));
    for (uint32_t p = threadIdx.x; p < T * T; p += blockDim.x) {
        const uint32_t x = cx + (p % T), y = cy + (p / T);
        if (x < P.width && y < P.height) S->image2d[(size_t)y * P.width + x] = f;
    }
}

// ======================================================================================
// Point evaluation with a VGPR register file: NR registers x ZB values per lane.
// One array per z so that each stays within the 32-dword limit of uniform dynamic VGPR
// indexing (s_set_gpr_idx_on); larger tapes take the LDS path below.
// ======================================================================================
#define FOR_Z for (int j = 0; j < ZB; j++)

// Evaluate `tape` at ZB points per lane; x/y/z are model-space coordinates.
template <int NR, int ZB, bool FULL>
FH_DEV void run_points(ctape_t tape, uint32_t len, const AS4 FhRender& P, const float (&x)[ZB], const float (&y)[ZB],
                       const float (&z)[ZB], float (&res)[ZB]) {
    // one array per z: each must stay within 32 dwords to be kept in VGPRs
    float r0[NR], r1[ZB > 1 ? NR : 1], r2[ZB > 2 ? NR : 1], r3[ZB > 2 ? NR : 1];
#define RGET(i, v) do { v[0] = r0[i]; if (ZB > 1) v[1] = r1[i]; if (ZB > 2) { v[2] = r2[i]; v[3] = r3[i]; } } while (0)
#define RSET(i, v) do { r0[i] = v[0]; if (ZB > 1) r1[i] = v[1]; if (ZB > 2) { r2[i] = v[2]; r3[i] = v[3]; } } while (0)
    const AS4 Q4* q = (const AS4 Q4*)tape;
    Q4 cur = q[0];
    for (uint32_t k0 = 0; k0 < len; k0 += 4) {
        const Q4 nxt = q[(k0 >> 2) + 1];  // the arena keeps 8 ops of slack past its last tape
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (k0 + u >= len) break;
            const uint64_t w = cur[u];
            const uint32_t w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
            const uint32_t op = FH_W_OP(w0), ro = FH_W_OUT(w0), ra = FH_W_A(w0);
            float a[ZB], b[ZB], c[ZB];
            if (op >= FH_ADD_RR) {
                int base;
                if (op < FH_ADD_RI) { base = op - FH_ADD_RR; RGET(ra, a); RGET(w1, b); }
                else if (op < FH_SUB_IR) { base = op - FH_ADD_RI; RGET(ra, a); FOR_Z b[j] = u2f(w1); }
                else { const int irb = op - FH_SUB_IR; base = irb == 0 ? 1 : irb + 2; FOR_Z a[j] = u2f(w1); RGET(ra, b); }
                int ch;
                switch (base) {
                    case 0: FOR_Z c[j] = a[j] + b[j]; break;
                    case 1: FOR_Z c[j] = a[j] - b[j]; break;
                    case 2: FOR_Z c[j] = a[j] * b[j]; break;
                    case 3: FOR_Z c[j] = a[j] / b[j]; break;
                    case 4: if (FULL) { FOR_Z c[j] = t_atan2(a[j], b[j]); } break;
                    case 5: FOR_Z c[j] = f_compare(a[j], b[j]); break;
                    case 6: FOR_Z c[j] = f_mix(a[j], b[j]); break;
                    case 7: if (FULL) { FOR_Z c[j] = rem_euclid(a[j], b[j]); } break;
                    case 8: FOR_Z c[j] = f_min(a[j], b[j], ch); break;
                    case 9: FOR_Z c[j] = f_max(a[j], b[j], ch); break;
                    case 10: FOR_Z c[j] = f_and(a[j], b[j], ch); break;
                    default: FOR_Z c[j] = f_or(a[j], b[j], ch); break;
                }
                RSET(ro, c);
            } else if (op >= FH_NEG) {
                RGET(ra, a);
                switch (op) {
                    case FH_NEG: FOR_Z c[j] = -a[j]; break;
                    case FH_ABS: FOR_Z c[j] = fabsf(a[j]); break;
                    case FH_RECIP: FOR_Z c[j] = 1.0f / a[j]; break;
                    case FH_SQRT: FOR_Z c[j] = sqrtf(a[j]); break;
                    case FH_SQUARE: FOR_Z c[j] = a[j] * a[j]; break;
                    case FH_FLOOR: FOR_Z c[j] = floorf(a[j]); break;
                    case FH_CEIL: FOR_Z c[j] = ceilf(a[j]); break;
                    case FH_ROUND: FOR_Z c[j] = roundf(a[j]); break;
                    case FH_NOT: FOR_Z c[j] = a[j] == 0.0f ? 1.0f : 0.0f; break;
                    case FH_RAND: FOR_Z c[j] = f_rand(a[j]); break;
                    default:
                        if (FULL) { FOR_Z c[j] = F32::unary<true>((int)op, a[j]); }
                        break;
                }
                RSET(ro, c);
            } else if (op == FH_INPUT) {
                const uint32_t kd = P.in_kind[w1];
                if (kd == 0) { FOR_Z c[j] = x[j]; } else if (kd == 1) { FOR_Z c[j] = y[j]; }
                else if (kd == 2) { FOR_Z c[j] = z[j]; } else { const float v = P.in_value[w1]; FOR_Z c[j] = v; }
                RSET(ro, c);
            } else if (op == FH_COPY_REG) {
                RGET(ra, a);
                RSET(ro, a);
            } else if (op == FH_COPY_IMM) {
                FOR_Z c[j] = u2f(w1);
                RSET(ro, c);
            } else {
                RGET(ra, a);
                FOR_Z res[j] = a[j];
            }
        }
        cur = nxt;
    }
}
#undef RGET
#undef RSET

// Same, LDS register file, one value per lane (tapes with more than 32 registers)
template <bool FULL>
FH_DEV float run_points_lds(ctape_t tape, uint32_t len, const AS4 FhRender& P, float* lds, int lane, float x, float y, float z) {
    Regs<float, WAVE> R{lds, lane};
    float res = 0.0f;
    for (uint32_t k = 0; k < len; k++) {
        const uint64_t w = tape[k];
        step<F32, WAVE, FULL>(
            w, R,
            [&](uint32_t slot) {
                const uint32_t kd = P.in_kind[slot];
                return kd == 0 ? x : (kd == 1 ? y : (kd == 2 ? z : (float)P.in_value[slot]));
            },
            [&](uint32_t, float v) { res = v; }, [&](int) {});
    }
    return res;
}

// 2D leaves: one T x T pixel tile (T*T <= 64) per wave (pixel.rs:400-441).
// NR == 0 selects the LDS register file (big tapes, queued from the back of `leaves`).
template <int NR, bool FULL>
__global__ void __launch_bounds__(WAVE) k_pixels2d(FhRenderState* S) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const AS4 FhRender& P = *(const AS4 FhRender*)&S->P;
    const int lane = threadIdx.x;
    const uint32_t T = P.tiles[P.n_levels - 1];
    Mat4 mat;
#pragma unroll
    for (int i = 0; i < 16; i++) mat.m[i] = P.mat[i];
    const uint32_t n_leaves = min(S->n_leaves, S->leaf_cap);
    for (uint32_t li = blockIdx.x; li < n_leaves; li += gridDim.x) {
        const AS4 FhLeaf& lf = *(const AS4 FhLeaf*)&S->leaves[li];
        const bool fits = lf.tape.n_regs <= 32;
        if (NR ? !fits : fits) continue;  // the other variant renders this leaf
        const ctape_t tape = (ctape_t)(S->arena + lf.tape.off);
        for (uint32_t p0 = 0; p0 < T * T; p0 += WAVE) {
            const uint32_t p = p0 + lane;
            const uint32_t px = lf.x + (p % T), py = lf.y + (p / T);
            float x[1], y[1], z[1], res[1] = {0.0f};
            xf_point(mat, (float)px, (float)py, P.z, x[0], y[0], z[0]);
            if (NR) run_points<(NR ? NR : 1), 1, FULL>(tape, lf.tape.len, P, x, y, z, res);
            else res[0] = run_points_lds<FULL>(tape, lf.tape.len, P, (float*)big_file(S, smem), lane, x[0], y[0], z[0]);
            if (p < T * T && px < P.width && py < P.height)
                S->image2d[(size_t)py * P.width + px] = isnan_(res[0]) ? u2f(0x7FC00000u) : res[0];  // pixel.rs:235-241
        }
    }
}

// Footprint classification for the 3D column kernels: a footprint goes to the variant that
// fits the largest register count among its leaves (class 0: <= 16, 1: <= 32, 2: LDS).
// grid: blocks of 256 threads = 32 footprints x 8 layer groups: a thread reads every eighth layer of its footprint's column - up to eight
// independent loads, all in flight at once - and the eight partial maxima meet in LDS.  (One thread per footprint walking its 64
// layers took 26 us at 1024^2, on the caller's stream of every frame since the tail stream carries root levels: 64 loads, eight in
// flight.)
#define FH_CLASSIFY_FP 32
template <int CLS, int NR, int ZB, bool FULL>
FH_DEV void leaves3d_body(FhRenderState* S, char* file, uint32_t first, uint32_t stride);
// (rare mode, capi_render.hpp: the blocks behind `class_blocks` are the slab's launch for leaves of more than 32 registers - none, nearly
// always - with their register files in HBM, `rare_stride` bytes each from `rare_file`: a wave per block)
// (workgroups of one wave - two layer groups per footprint - for slabs of up to 16 layers: beside a leaf kernel that fills the machine a workgroup
// of four waves waits for four free slots of one compute unit; deep slabs - the 128 layers of a frame without z - keep eight layer groups)
__global__ void __launch_bounds__(256) k_classify3d(FhRenderState* S, int merge01, uint32_t class_blocks, char* rare_file, uint32_t rare_stride) {
    if (blockIdx.x >= class_blocks) {
        if (threadIdx.x < WAVE) leaves3d_body<2, 0, 1, true>(S, rare_file + (size_t)(blockIdx.x - class_blocks) * rare_stride, blockIdx.x - class_blocks, gridDim.x - class_blocks);
        return;
    }
    __shared__ uint32_t mx_s[FH_CLASSIFY_FP], any_s[FH_CLASSIFY_FP];
    const AS4 FhRender& P = *(const AS4 FhRender*)&S->P;
    const uint32_t T = P.tiles[P.n_levels - 1];
    const uint32_t fw = (P.width + T - 1) / T, fh = (P.height + T - 1) / T;
    const uint32_t layers = P.slab / T;
    const uint32_t fl = threadIdx.x % FH_CLASSIFY_FP, lg = threadIdx.x / FH_CLASSIFY_FP, n_lg = blockDim.x / FH_CLASSIFY_FP;
    const uint32_t fi = blockIdx.x * FH_CLASSIFY_FP + fl;
    if (threadIdx.x < FH_CLASSIFY_FP) { mx_s[threadIdx.x] = 0; any_s[threadIdx.x] = 0; }
    __syncthreads();
    const FhLeafRef* col = S->leaf_table + fi;  // [layer][footprint]
    uint32_t mx = 0;
    bool any = false;
    // (the table entry carries the leaf's register count beside its number: no dependent load of the leaf record per layer)
    if (fi < fw * fh) {
#pragma unroll 8
        for (uint32_t l = lg; l < layers; l += n_lg) {
            const FhLeafRef e = col[(size_t)l * fw * fh];
            if (e.id) { any = true; mx = max(mx, e.len_regs >> 24); }
        }
    }
    if (any) { atomicMax(&mx_s[fl], mx); any_s[fl] = 1; }
    __syncthreads();
    if (threadIdx.x >= FH_CLASSIFY_FP) return;
    mx = mx_s[fl];
    any = any_s[fl] != 0;
    // one atomic per block and class instead of one per footprint
    // merge01: the assembly leaf kernel picks the register-file shape per leaf, one list for <= 32 registers
    const int cls = !any ? -1 : (mx <= 16 ? 0 : (mx <= (merge01 ? S->norm_asm_regs : 32u) ? (merge01 ? 0 : 1) : 2));
    const int lane = threadIdx.x & (WAVE - 1);
    for (int c = 0; c < 3; c++) {
        const uint64_t m = ballot(cls == c);
        if (m == 0) continue;
        uint32_t base = 0;
        if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&S->fp_count[c], (uint32_t)__popcll(m));
        base = __shfl(base, __builtin_ctzll(m), WAVE);
        if (cls == c) S->fp_list[c][base + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = ((fi / fw) << 16) | (fi % fw);  // fy, fx
    }
}

// 3D leaves in HIP (tapes outside the assembly opcode set, or needing the LDS register file): ONE
// leaf (8x8x8 voxels, one pixel column per lane) per wave pass, static round robin over the slab's
// leaves; the voxels of a column front to back, ZB per lane at a time (voxel.rs:359-463).  Hits go
// to the z-buffer with a 64-bit atomic max (depth << 32 | leaf), so the order in which waves reach
// the leaves of one column does not matter; a leaf whose pixels are all hit in front of it retires
// after one load.  CLS: leaves of <= 16 registers, 17..32, more (LDS register file).
// (file: the register file of class 2 - LDS, or a region of HBM; first, stride: this wave's leaves)
template <int CLS, int NR, int ZB, bool FULL>
FH_DEV void leaves3d_body(FhRenderState* S, char* file, uint32_t first, uint32_t stride) {
    const AS4 FhRender& P = *(const AS4 FhRender*)&S->P;
    const int lane = threadIdx.x & (WAVE - 1);
    const uint32_t T = P.tiles[P.n_levels - 1];  // T*T == 64 lanes
    Mat4 mat;
#pragma unroll
    for (int i = 0; i < 16; i++) mat.m[i] = P.mat[i];
    const uint32_t n_leaves = min(S->n_leaves, S->leaf_cap);
    if (CLS == 2 && S->n_leaves_lds == 0) return;
    for (uint32_t li = first; li < n_leaves; li += stride) {
        const AS4 FhLeaf& lf = *(const AS4 FhLeaf*)&S->leaves[li];
        const uint32_t regs = lf.tape.n_regs;
        if (CLS == 0 ? regs > 16 : (CLS == 1 ? (regs <= 16 || regs > 32) : regs <= S->leaf_asm_regs)) continue;
        const uint32_t px = lf.x + (lane % T), py = lf.y + (lane / T), lz = lf.z;
        const bool inimg = px < P.width && py < P.height;
        const size_t pix = (size_t)py * P.width + px;
        uint32_t depth = inimg ? (uint32_t)(S->zbuf[pix] >> 32) : 0xFFFFFFFFu;
        bool pending = depth < lz + T;  // voxel.rs:377-381
        if (ballot(pending) == 0) continue;
        const ctape_t tape = (ctape_t)(S->arena + lf.tape.off);
        const uint32_t len = lf.tape.len;
        bool hit = false;
        for (int k = (int)T - 1; k >= 0; k -= ZB) {
            float x[ZB], y[ZB], z[ZB], res[ZB];
            FOR_Z { xf_point(mat, (float)px, (float)py, (float)(lz + k - j), x[j], y[j], z[j]); res[j] = 0.0f; }
            if (NR) run_points<(NR ? NR : 1), ZB, FULL>(tape, len, P, x, y, z, res);
            else res[0] = run_points_lds<FULL>(tape, len, P, (float*)file, lane, x[0], y[0], z[0]);
            FOR_Z {
                if (pending && res[j] < 0.0f) {  // first voxel inside, front to back
                    depth = lz + (uint32_t)(k - j) + 1;
                    hit = true;
                    pending = false;
                }
            }
            if (ballot(pending) == 0) break;
        }
        if (hit) atomicMax((unsigned long long*)&S->zbuf[pix], ((unsigned long long)depth << 32) | (li + 1));
    }
}
template <int CLS, int NR, int ZB, bool FULL>
__global__ void __launch_bounds__(WAVE) k_leaves3d(FhRenderState* S) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    leaves3d_body<CLS, NR, ZB, FULL>(S, CLS == 2 ? big_file(S, smem) : smem, blockIdx.x, gridDim.x);
}

// Normals for the hits of this slab: gradient of the winning leaf's tape at the voxel one
// above the hit (voxel.rs:447-482).  Lanes of a footprint may have been hit in different
// leaves; the wave loops over the distinct ones.
// BIG: footprints of class 2 (LDS sized by the root tape); otherwise classes 0 and 1
// (<= 32 registers, small LDS so that many waves fit).
// z_lo, z_hi: only hits of this slab (z_lo < depth <= z_hi) are this launch's - the leaf kernel of the slab behind may already
// be running (its hits lie below z_lo and carry leaf numbers of the other slab context).
template <bool FULL, bool BIG>
FH_DEV void normals3d_body(FhRenderState* S, uint32_t z_lo, uint32_t z_hi, char* file, uint32_t first, uint32_t stride) {
    const AS4 FhRender& P = *(const AS4 FhRender*)&S->P;
    const int lane = threadIdx.x & (WAVE - 1);
    const uint32_t T = P.tiles[P.n_levels - 1];
    const uint32_t fw = (P.width + T - 1) / T;
    Regs<GR, WAVE> R{(GR*)file, lane};
    Mat4 mat;
#pragma unroll
    for (int i = 0; i < 16; i++) mat.m[i] = P.mat[i];
    const uint32_t n_all = BIG ? S->fp_count[2] : S->fp_count[0] + S->fp_count[1];
    // static round robin: an atomic cursor per footprint would cost more than the work (most
    // footprints have no pending hit)
    for (uint32_t wi = first; wi < n_all; wi += stride) {
        uint32_t fi;
        if (BIG) fi = S->fp_list[2][wi];
        else if (wi < S->fp_count[0]) fi = S->fp_list[0][wi];
        else fi = S->fp_list[1][wi - S->fp_count[0]];
        fi = uni(fi);
        const uint32_t px = (fi & 0xFFFFu) * T + (lane % T), py = (fi >> 16) * T + (lane / T);
        const bool inimg = px < P.width && py < P.height;
        const size_t pix = (size_t)py * P.width + px;
        const uint64_t zb = inimg ? S->zbuf[pix] : 0;
        uint32_t id = (uint32_t)zb;
        const uint32_t depth = (uint32_t)(zb >> 32);
        if (depth <= z_lo || depth > z_hi) id = 0;
        uint64_t todo = ballot(id != 0);
        while (todo) {
            const uint32_t cur = uni(__shfl(id, __builtin_ctzll(todo), WAVE));
            const bool mine = (id == cur);
            const AS4 FhLeaf& lf = *(const AS4 FhLeaf*)&S->leaves[cur - 1];
            const ctape_t tape = (ctape_t)(S->arena + lf.tape.off);
            const uint32_t len = lf.tape.len;
            GR gx, gy, gz, res = gr1(0.0f);
            xf_grad(mat, gr((float)px, 1, 0, 0), gr((float)py, 0, 1, 0), gr((float)(depth - 1), 0, 0, 1), gx, gy, gz);
            for (uint32_t q = 0; q < len; q++) {
                const uint64_t w = tape[q];
                step<GRAD, WAVE, FULL>(
                    w, R,
                    [&](uint32_t slot) {
                        const uint32_t kd = P.in_kind[slot];
                        return kd == 0 ? gx : (kd == 1 ? gy : (kd == 2 ? gz : gr1(P.in_value[slot])));
                    },
                    [&](uint32_t, GR v) { res = v; }, [&](int) {});
            }
            if (mine) {
                S->normals[pix * 3 + 0] = res.dx; S->normals[pix * 3 + 1] = res.dy; S->normals[pix * 3 + 2] = res.dz;
                S->zbuf[pix] = zb & 0xFFFFFFFF00000000ull;  // normal done
                id = 0;
            }
            todo &= ~ballot(mine);
        }
    }
}
template <bool FULL, bool BIG>
__global__ void __launch_bounds__(WAVE) k_normals3d(FhRenderState* S, uint32_t z_lo, uint32_t z_hi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    normals3d_body<FULL, BIG>(S, z_lo, z_hi, BIG ? big_file(S, smem) : smem, blockIdx.x, gridDim.x);
}

// The work list of the assembly normals kernel (fh_normals, gen_normals.py): every leaf that owns a hit of this slab's depth range in the
// finished z-buffer, once.  A wave per footprint of the classes the assembly kernel takes (lists 0 and 1 of k_classify3d) reads the 64
// z-buffer words and appends the distinct leaf numbers among them - one atomic per footprint.  The normals kernel then runs ONE tape per
// wave pass: by footprints (a wave = all the leaves its footprints' pixels were hit by, one after the other) a launch lasted as long as its
// busiest wave - bear.vm 512^3, 350-430 ops per leaf tape, 2-6 leaves per footprint: 414 us per launch for ~100 us of work per wave slot.
// The list is FH_HIT_BUCKETS lists (hit_list: the counters, 256 bytes apart, then the buckets of `bucket_cap` entries): footprint wi of the
// class lists appends to bucket wi % 64, and wave w of the normals kernel walks bucket w % 64.  (One counter for all: 5.5 k atomics that
// return a value on ONE address took 82 us on prospero.vm 1024^3 - 15 ns each, one after the other.)
// (rare mode: the blocks behind `hit_blocks` are the slab's normals launch for the footprints with a leaf of more than 32 registers, as in
// k_classify3d)
// (workgroups of ONE wave: beside a leaf kernel that fills the machine a workgroup of four waits for four free slots of one compute unit -
// 195 us per launch on the general path against 4 alone)
__global__ void __launch_bounds__(WAVE) k_hits3d(FhRenderState* S, uint32_t z_lo, uint32_t z_hi, uint32_t bucket_cap, uint32_t hit_blocks, char* rare_file, uint32_t rare_stride) {
    if (blockIdx.x >= hit_blocks) {
        normals3d_body<true, true>(S, z_lo, z_hi, rare_file + (size_t)(blockIdx.x - hit_blocks) * rare_stride, blockIdx.x - hit_blocks, gridDim.x - hit_blocks);
        return;
    }
    const AS4 FhRender& P = *(const AS4 FhRender*)&S->P;
    const uint32_t T = P.tiles[P.n_levels - 1];
    const int lane = threadIdx.x & (WAVE - 1);
    const uint32_t n0 = S->fp_count[0], n_all = n0 + S->fp_count[1];
    for (uint32_t wi = blockIdx.x; wi < n_all; wi += hit_blocks) {
        const uint32_t fi = uni(wi < n0 ? S->fp_list[0][wi] : S->fp_list[1][wi - n0]);
        const uint32_t px = (fi & 0xFFFFu) * T + (lane % T), py = (fi >> 16) * T + (lane / T);
        const uint64_t zb = (px < P.width && py < P.height) ? S->zbuf[(size_t)py * P.width + px] : 0;
        uint32_t id = (uint32_t)zb;
        const uint32_t depth = (uint32_t)(zb >> 32);
        if (depth <= z_lo || depth > z_hi) id = 0;
        uint64_t todo = ballot(id != 0);
        uint32_t k = 0, mine = 0;      // lane k keeps the k-th distinct leaf
        while (todo) {
            const uint32_t cur = uni(__shfl(id, __builtin_ctzll(todo), WAVE));
            if (lane == (int)k) mine = cur;
            k++;
            todo &= ~ballot(id == cur);
        }
        if (k == 0) continue;
        const uint32_t b = wi % FH_HIT_BUCKETS;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&S->hit_list[b * FH_HIT_STRIDE], k);
        base = uni(__shfl(base, 0, WAVE));
        if ((uint32_t)lane < k && base + lane < bucket_cap) S->hit_list[FH_HIT_BUCKETS * FH_HIT_STRIDE + (size_t)b * bucket_cap + base + lane] = mine;
    }
}

// Per-slab reset of the work queues, leaf table and tape arena (frame-persistent tapes stay)
FH_DEV void reset_slab_body(FhRenderState* S, uint32_t i, uint32_t stride, uint32_t table_words, uint32_t slab, uint32_t n_root_groups, uint32_t reset_root_mind) {
    for (uint32_t k = i; k < table_words; k += stride) S->leaf_table[k] = FhLeafRef{0, 0, 0, 0};
    const uint32_t P0 = S->pre_levels;
    if (i == 0) {
        S->slab_z = slab * S->P.slab;
        for (uint32_t l = P0; l < FH_MAX_LEVELS; l++) {
            S->count[l] = 0; S->cursor[l] = 0; S->count_big[l] = 0; S->cursor_big[l] = 0;
            S->setup_cur[l] = 0; S->push_cur[l] = 0;
            S->n_slots[0][l] = 0; S->n_slots[1][l] = 0; S->eval_cur[0][l] = 0; S->eval_cur[1][l] = 0;
        }
        if (P0 == 0) {
            S->count_big[0] = n_root_groups;  // the root tape uses the large LDS layout
            S->arena_head = S->arena_root_end;
        } else {
            S->queue[P0] = S->squeue + (size_t)slab * S->squeue_cap;
            S->count[P0] = S->scount[slab];
            S->count_big[P0] = S->scount_big[slab];
            S->arena_head = S->arena_frame_end;
        }
        S->n_leaves = 0; S->n_leaves_lds = 0; S->leaf_cursor = 0; S->leaf_cursor_big = 0; S->normal_cursor = 0; S->normal_cursor_big = 0;
        for (int c = 0; c < 3; c++) { S->fp_count[c] = 0; S->fp_cursor[c] = 0; }
    }
    if (S->hit_list)
        for (uint32_t k = i; k < FH_HIT_BUCKETS; k += stride) S->hit_list[k * FH_HIT_STRIDE] = 0;
    if (P0 == 0 && i < n_root_groups) S->queue[0][S->qcap[0] - 1 - i].z = slab * S->P.slab;      // (no pre-pass: slab = one root-tile layer)
    // the coarse levels of the min-depth pyramid are rebuilt by k_minpyramid (not for the first slab: empty image)
    if (reset_root_mind) {
        const uint32_t T = S->P.tiles[0], n = ((S->P.width + T - 1) / T) * ((S->P.height + T - 1) / T);
        for (uint32_t k = i; k < n; k += stride) S->mind[0][k] = 0xFFFFFFFFu;
    }
}
__global__ void k_reset_slab(FhRenderState* S, uint32_t table_words, uint32_t slab, uint32_t n_root_groups, uint32_t reset_root_mind) {
    reset_slab_body(S, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x, table_words, slab, n_root_groups, reset_root_mind);
}
// End of the pre-pass: everything allocated so far lives for the whole frame
// The head of a frame in ONE launch (it was five asynchronous copies / fills, each a launch of the runtime's own with its gap in front -
// ~90 us of a 1.3 ms frame and 9 % of the default path's kernel time): the frame's state and root groups out of their pinned staging slot
// (host memory the device reads directly), and up to three buffers cleared - z-buffer, normals, the min-depth pyramid.
struct FhFrameBegin {
    uint32_t* state_dst; const uint32_t* state_src; uint32_t state_words;
    uint32_t* roots_dst; const uint32_t* roots_src; uint32_t roots_words;
    uint32_t* clear[3]; unsigned long long clear_words[3]; uint32_t fill[3];
};
// (workgroups of one wave - they find room beside another frame's leaf kernel where four waves together wait: 64 us against 11 -; the
// first four copy the state, the next four the root groups: grid >= 8)
__global__ void __launch_bounds__(WAVE) k_frame_begin(FhFrameBegin a) {
    if (blockIdx.x < 4) {
        for (uint32_t i = blockIdx.x * WAVE + threadIdx.x; i < a.state_words; i += 4 * WAVE) a.state_dst[i] = a.state_src[i];
    } else if (blockIdx.x < 8) {
        for (uint32_t i = (blockIdx.x - 4) * WAVE + threadIdx.x; i < a.roots_words; i += 4 * WAVE) a.roots_dst[i] = a.roots_src[i];
    }
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    for (int k = 0; k < 3; k++) {
        if (!a.clear[k]) continue;
        const uint4 f = make_uint4(a.fill[k], a.fill[k], a.fill[k], a.fill[k]);
        uint4* p4 = (uint4*)a.clear[k];                      // (device allocations: 256-byte aligned)
        const size_t n4 = a.clear_words[k] / 4;
        for (size_t i = tid; i < n4; i += nth) p4[i] = f;
        for (size_t i = n4 * 4 + tid; i < a.clear_words[k]; i += nth) a.clear[k][i] = a.fill[k];
    }
}
__global__ void k_mark_frame(FhRenderState* S) { S->arena_frame_end = min(S->arena_head, S->arena_cap); }

// Min-depth pyramid of the z-buffer, one workgroup per root tile: mind[l][tile] = smallest
// depth over the tile's in-image pixels.  Feeds the occlusion test of k_tiles.
__global__ void __launch_bounds__(256) k_minpyramid(FhRenderState* S) {
    const AS4 FhRender& P = *(const AS4 FhRender*)&S->P;
    const uint32_t T0 = P.tiles[0];
    const uint32_t x0 = (blockIdx.x / P.roots_y) * T0, y0 = (blockIdx.x % P.roots_y) * T0;
    for (int l = (int)P.n_levels - 1; l >= 0; l--) {
        const uint32_t T = P.tiles[l], n = T0 / T, ntx = (P.width + T - 1) / T;
        for (uint32_t t = threadIdx.x; t < n * n; t += blockDim.x) {
            const uint32_t tx = x0 / T + t % n, ty = y0 / T + t / n;
            if (tx * T >= P.width || ty * T >= P.height) continue;
            uint32_t mn = 0xFFFFFFFFu;
            if (l == (int)P.n_levels - 1) {
                for (uint32_t p = 0; p < T * T; p++) {
                    const uint32_t x = tx * T + p % T, y = ty * T + p / T;
                    if (x < P.width && y < P.height) mn = min(mn, (uint32_t)(S->zbuf[(size_t)y * P.width + x] >> 32));
                }
            } else {
                const uint32_t Tc = P.tiles[l + 1], c = T / Tc, ncx = (P.width + Tc - 1) / Tc;
                for (uint32_t q = 0; q < c * c; q++) {
                    const uint32_t sx = tx * c + q % c, sy = ty * c + q / c;
                    if (sx * Tc < P.width && sy * Tc < P.height) mn = min(mn, S->mind[l + 1][sy * ncx + sx]);
                }
            }
            S->mind[l][ty * ntx + tx] = mn;
        }
        __threadfence_block();
        __syncthreads();
    }
}

// The same for the usual shape of the pyramid (three levels, each 4 x 4 of the next, 8 x 8 leaf tiles): one
// block per tile of the middle level, one wave per leaf tile in it, lane = pixel (coalesced rows); the
// root level, reset to ~0 by k_reset_slab, takes 16 atomics per word.  (The kernel above, one thread
// walking each leaf tile, took 30 us of every slab's tile chain.)
template <bool ROOT>
FH_DEV void minpyramid3_body(FhRenderState* S, uint32_t block) {
    // one block of four waves per middle-level tile, four leaf tiles per wave (a block of 16 waves finds no room on a CU
    // while the leaf kernel of the slab in front keeps the machine full: 166 us instead of 10)
    __shared__ uint32_t part[4];
    const AS4 FhRender& P = *(const AS4 FhRender*)&S->P;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t T2 = P.tiles[2], T1 = P.tiles[1], T0 = P.tiles[0];
    const uint32_t n1x = (P.width + T1 - 1) / T1, n2x = (P.width + T2 - 1) / T2, n0x = (P.width + T0 - 1) / T0;
    const uint32_t bx = block % n1x, by = block / n1x;
    const uint32_t ty = by * 4 + w, y = ty * T2 + (lane >> 3);
    uint32_t v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t x = (bx * 4 + i) * T2 + (lane & 7);
        v[i] = (x < P.width && y < P.height) ? (uint32_t)(S->zbuf[(size_t)y * P.width + x] >> 32) : 0xFFFFFFFFu;
    }
    uint32_t row = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t mn = v[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, d, WAVE));
        const uint32_t tx = bx * 4 + i;
        if (lane == 0 && tx * T2 < P.width && ty * T2 < P.height) S->mind[2][ty * n2x + tx] = mn;
        row = min(row, mn);
    }
    if (lane == 0) part[w] = row;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t m = min(min(part[0], part[1]), min(part[2], part[3]));
        S->mind[1][by * n1x + bx] = m;
        if (ROOT) atomicMin(&S->mind[0][(by * T1 / T0) * n0x + bx * T1 / T0], m);
    }
}
__global__ void __launch_bounds__(256) k_minpyramid3(FhRenderState* S) { minpyramid3_body<true>(S, blockIdx.x); }
// Start of a slab's tile chain in one launch (every kernel boundary on that chain costs ~10 us beside a full machine): the
// first n1 blocks rebuild the two fine levels of the min-depth pyramid, the others reset the slab's queues and leaf table.
// For the 128 / 32 / 8 pyramid with two pre-pass levels: the per-slab level only reads pyramid levels 1 and 2, so the root
// level (which the reset would have to clear before the rebuild's atomic minima) is left alone.
__global__ void __launch_bounds__(256) k_slab_begin3(FhRenderState* S, uint32_t n1, uint32_t table_words, uint32_t slab, uint32_t n_root_groups) {
    if (blockIdx.x < n1) { minpyramid3_body<false>(S, blockIdx.x); return; }
    reset_slab_body(S, (blockIdx.x - n1) * blockDim.x + threadIdx.x, (gridDim.x - n1) * blockDim.x, table_words, slab, n_root_groups, 0);
}

// ... and for the two-level pyramid of a frame with root tiles of 32^3 (32 / 8, one pre-pass level: capi_render.hpp root32_max): one block
// of four waves per root tile, four leaf tiles per wave, lane = pixel; both levels are written outright (no level above them to fold into)
FH_DEV void minpyramid2_body(FhRenderState* S, uint32_t block) {
    __shared__ uint32_t part[4];
    const AS4 FhRender& P = *(const AS4 FhRender*)&S->P;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t T1 = P.tiles[1], T0 = P.tiles[0];
    const uint32_t n0x = (P.width + T0 - 1) / T0, n1x = (P.width + T1 - 1) / T1;
    const uint32_t bx = block % n0x, by = block / n0x;
    const uint32_t ty = by * 4 + w, y = ty * T1 + (lane >> 3);
    uint32_t row = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t tx = bx * 4 + i, x = tx * T1 + (lane & 7);
        uint32_t mn = (x < P.width && y < P.height) ? (uint32_t)(S->zbuf[(size_t)y * P.width + x] >> 32) : 0xFFFFFFFFu;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, d, WAVE));
        if (lane == 0 && tx * T1 < P.width && ty * T1 < P.height) S->mind[1][ty * n1x + tx] = mn;
        row = min(row, mn);
    }
    if (lane == 0) part[w] = row;
    __syncthreads();
    if (threadIdx.x == 0) S->mind[0][by * n0x + bx] = min(min(part[0], part[1]), min(part[2], part[3]));
}
__global__ void __launch_bounds__(256) k_slab_begin2(FhRenderState* S, uint32_t n0, uint32_t table_words, uint32_t slab, uint32_t n_root_groups) {
    if (blockIdx.x < n0) { minpyramid2_body(S, blockIdx.x); return; }
    reset_slab_body(S, (blockIdx.x - n0) * blockDim.x + threadIdx.x, (gridDim.x - n0) * blockDim.x, table_words, slab, n_root_groups, 0);
}

// Final image (voxel.rs:524-552): saturated columns become (D, [0,0,1])
// ... and the frame's queue-overflow flags (one per slab context) are latched into the context's sticky word: with frames
// pipelined over two buffer sets, a set is re-used by the frame after next before the host has looked at its flags.
// The frame's arena state goes to a pinned host word the same way (capi_core.hpp host_flags: [0] the arena ran out somewhere in the frame,
// [1] the most ops any slab context had in use): the next render call reads it without waiting and grows the arena.
FH_DEV void latch_arena(const FhRenderState* S, uint32_t n_ctx, volatile uint32_t* host_flags) {
    uint32_t a = 0, peak = 0;
    for (uint32_t k = 0; k < n_ctx; k++) { a |= S[k].arena_overflow; peak = max(peak, S[k].arena_head); }
    if (a) host_flags[0] = 1u;
    if (peak > host_flags[1]) host_flags[1] = peak;
}
__global__ void k_latch_arena(const FhRenderState* S, uint32_t n_ctx, volatile uint32_t* host_flags) { latch_arena(S, n_ctx, host_flags); }
__global__ void k_finish3d(FhRenderState* S, FhGeometryPixel* out, uint32_t n_ctx, uint32_t* sticky, volatile uint32_t* host_flags) {
    const AS4 FhRender& P = *(const AS4 FhRender*)&S->P;
    const size_t n = (size_t)P.width * P.height;
    if (blockIdx.x == 0 && threadIdx.x == 0 && sticky) {
        uint32_t q = 0;
        for (uint32_t k = 0; k < n_ctx; k++) q |= S[k].queue_overflow;
        if (q) atomicOr(sticky, 1u);
        if (host_flags) {
            latch_arena(S, n_ctx, host_flags);
            uint32_t r = 0;
            for (uint32_t k = 0; k < n_ctx; k++) r |= S[k].rare_seen;
            host_flags[2] = r;      // the frame met a large tape: the next frames launch the kernels for them on their own again (capi_render.hpp rare mode)
        }
    }
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t d = (uint32_t)(S->zbuf[i] >> 32);
        FhGeometryPixel o;
        if (d >= P.depth - 1) { o.normal[0] = 0.0f; o.normal[1] = 0.0f; o.normal[2] = 1.0f; o.depth = P.depth; }
        else { o.normal[0] = S->normals[i * 3]; o.normal[1] = S->normals[i * 3 + 1]; o.normal[2] = S->normals[i * 3 + 2]; o.depth = d; }
        out[i] = o;
    }
}

// Multi-GPU merge of partial images over the same pixels (z-split shards); see fhip_merge_depth
__global__ void k_merge_depth(FhGeometryPixel* front, const FhGeometryPixel* back, size_t n, uint32_t D) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        FhGeometryPixel a = front[i];
        const FhGeometryPixel b = back[i];
        if (b.depth > a.depth) a = b;     // ties: the front range keeps its pixel
        if (D && a.depth >= D - 1) { a.depth = D; a.normal[0] = 0.0f; a.normal[1] = 0.0f; a.normal[2] = 1.0f; }   // voxel.rs:533-539
        front[i] = a;
    }
}

// Sweep of the transcendental opcodes (diagnostics; tests/test_gpu_math.py): x_i = bits(first + i * stride), y = the device's f32
// result (dev_ops.hpp t_* = trans_libm.hpp), compared with ref[i] (the host libm's; op 8 = atan2 with the second argument the oracle's
// sweep uses): out[0] = max ulp distance, out[1] = results whose BITS differ (a NaN equals any NaN; +0 and -0 differ), out[2] = results
// more than 1 ulp apart, out[3] = an input of the worst case
__global__ void k_math_sweep(int op, uint32_t first, uint32_t stride, size_t n, const float* __restrict__ ref, unsigned long long* out) {
    unsigned long long worst = 0, ndiff = 0, nbad = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t xb = first + (uint32_t)i * stride;
        const float x = u2f(xb);
        float y;
        switch (op) {
            case 0: y = t_sin(x); break;
            case 1: y = t_cos(x); break;
            case 2: y = t_tan(x); break;
            case 3: y = t_asin(x); break;
            case 4: y = t_acos(x); break;
            case 5: y = t_atan(x); break;
            case 6: y = t_exp(x); break;
            case 8: y = t_atan2(x, u2f(xb * 2654435761u + 0x9E3779B9u)); break;
            default: y = t_ln(x); break;
        }
        const float r = ref[i];
        if (isnan_(y) && isnan_(r)) continue;
        // distance in units in the last place: map the bit patterns to a monotonic integer line
        const int32_t a = (int32_t)f2u(y), b = (int32_t)f2u(r);
        const long long ka = a < 0 ? -(long long)(f2u(y) & 0x7fffffffu) : (long long)a;
        const long long kb = b < 0 ? -(long long)(f2u(r) & 0x7fffffffu) : (long long)b;
        unsigned long long d = (unsigned long long)(ka > kb ? ka - kb : kb - ka);
        if (isnan_(y) != isnan_(r)) d = 0xFFFFFFFFull;
        if (f2u(y) != f2u(r)) ndiff++;
        if (!d && f2u(y) != f2u(r)) d = 1;   // +0 against -0
        if (d > 1) nbad++;
        if (d > (worst >> 32)) worst = (d << 32) | xb;
    }
    if (ndiff) atomicAdd(&out[1], ndiff);
    if (nbad) atomicAdd(&out[2], nbad);
    if (worst) atomicMax(&out[0], worst);
}

// The embedded copies of the compiled routines (fh_trans_probe, gen_interp.py) against the inlined ones above: got[i] = what copy
// computed for x = bits(first + i) (second argument as in k_math_sweep's op 8); fn indexes gen_trans.FUNCS + FUNCS4.  out[1] = results
// whose bits differ (a NaN equals any NaN), out[0] = (1 << 32 | an input where they do)
__global__ void k_trans_compare(int fn, uint32_t first, size_t n, const uint32_t* __restrict__ got, unsigned long long* out) {
    unsigned long long ndiff = 0, where = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t xb = first + (uint32_t)i;
        const float x = u2f(xb), b = u2f(xb * 2654435761u + 0x9E3779B9u);
        float y;
        switch (fn) {
            case 0: case 10: y = t_sin(x); break;
            case 1: case 11: y = t_cos(x); break;
            case 2: y = t_tan(x); break;
            case 3: y = t_asin(x); break;
            case 4: y = t_acos(x); break;
            case 5: y = t_atan(x); break;
            case 6: case 12: y = t_exp(x); break;
            case 8: y = t_atan2(x, b); break;
            case 9: y = rem_euclid(x, b); break;
            default: y = t_ln(x); break;
        }
        const float g = u2f(got[i]);
        if (isnan_(y) && isnan_(g)) continue;
        if (f2u(y) != got[i]) { ndiff++; where = (1ull << 32) | xb; }
    }
    if (ndiff) { atomicAdd(&out[1], ndiff); atomicMax(&out[0], where); }
}

// ---- micro-benchmark of the point interpreter (diagnostics only; fhip_debug_bench) -----------
template <int NR, int ZB>
__global__ void __launch_bounds__(WAVE) k_bench_points(FhRenderState* S, const uint64_t* tape_g, uint32_t len, uint32_t reps, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const AS4 FhRender& P = *(const AS4 FhRender*)&S->P;
    const ctape_t tape = (ctape_t)tape_g;
    float x[ZB], y[ZB], z[ZB], res[ZB], acc = 0.0f;
    FOR_Z { x[j] = (float)threadIdx.x * 0.01f + j; y[j] = (float)blockIdx.x * 0.001f; z[j] = 0.5f * j; res[j] = 0.0f; }
    for (uint32_t r = 0; r < reps; r++) {
        if (NR) run_points<(NR ? NR : 1), ZB, false>(tape, len, P, x, y, z, res);
        else res[0] = run_points_lds<false>(tape, len, P, (float*)smem, threadIdx.x, x[0], y[0], z[0]);
        FOR_Z { acc += res[j]; x[j] += 1e-3f; }
    }
    out[blockIdx.x * WAVE + threadIdx.x] = acc;
}
