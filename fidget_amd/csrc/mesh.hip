// fidget-hip: the evaluation side of fidget-mesh's octree construction on the device (fidget-mesh/src/octree.rs).
//
//   k_mesh_cells  one lane per octree cell of a level: cell bounds (cell.rs:184-194 midpoint splitting), interval
//                 evaluation of the shape's tape (octree.rs:521-544), classification Full / Empty / ambiguous; the ambiguous
//                 cells are appended to the next level's list (or, at the last level, to the leaf list);
//   k_mesh_corners / k_mesh_edges / k_mesh_grads   the ambiguous leaf cells sampled (octree.rs:590-862) in passes over a chunk of them,
//                 every lane a point of its own: the 8 corners (bulk f32) -> corner mask -> edges of the Manifold Dual Contouring
//                 table -> 4 rounds of 16-point search per edge (one wave per four edges) -> intersections (u16 cell
//                 coordinates) -> gradients there (one lane per edge);
//   k_mesh_leaf   the same with one wavefront per leaf cell (FHIP_MESH_LEAF_PASSES=0; what the passes are checked against);
//   k_mesh_leaf_qef  one lane per leaf record: one QEF per cell vertex (qef.rs).
//
// Tapes of 256 ops and more are simplified once down the octree (FhMeshParams::sub_tab: at the split level every ambiguous cell gets the root
// tape simplified under its own choices, and everything below it is evaluated with that - values do not depend on which ancestor's tape
// evaluates a cell); small tapes are evaluated as they are, the leaf samples through the assembly bulk interpreter.
//
//   k_oct_kind / k_oct_collapse / k_oct_place / k_oct_leaf_verts   the octree assembled from those results without leaving HBM
//                 (octree.rs:256-470 check_done / collapsible, 866-1035 merged Hermite data; mesh_collapse.hpp): level by level
//                 bottom-up what every ambiguous cell becomes, top-down where its vertices and its block of eight cells go.
//   k_walk_* / k_scan_*   Octree::walk_dual (dc.rs, builder.rs) on that octree: the recursion's calls as level arrays in call order, the
//                 mesh's vertices numbered by first use through atomic minima and prefix sums (mesh_walk.hpp).
#pragma once
#include <hip/hip_runtime.h>

#include "dev_ops.hpp"
#include "mesh_collapse.hpp"
#include "mesh_walk.hpp"
#include "mesh_edges.hpp"
#include "mesh_qef.hpp"
// (included by capi.hip after kernels.hip: Regs, step, ballot, uni, ctape_t)

struct FhMeshParams {
    const uint64_t* tape;
    uint32_t len, n_regs;
    float mat[16];
    uint32_t has_mat;
    uint32_t in_kind[FH_MAX_INPUTS];
    float in_value[FH_MAX_INPUTS];
    // Tape simplification down the octree (octree.rs:546-553, RenderHints::simplify_tree_during_meshing): once, at `split_level` - every
    // ambiguous cell there has a simplified tape of its own (VmData::simplify of the root tape under the cell's choices), which all the
    // cells and leaf samples below it use.  sub_tab[path of the ancestor at split_level - 8^split_level] = {offset into sub_ops, ops};
    // ops == 0 / sub_tab == null: the root tape.  (Values do not depend on which of an ancestor's tapes evaluates a cell: a min / max
    // decided over the ancestor's region is decided the same way everywhere inside it.)
    const uint64_t* sub_ops;
    const uint2* sub_tab;
    uint32_t split_level, pad_;
    // ... and a second time, further down (round 5: split_level2 = depth - 2, at most 7): the cells of that level get their ancestor's tape
    // simplified once more under their own choices - a 4^3-times smaller region, tapes a few times shorter again - for the levels and the leaf
    // samples below.  Same layout; an entry without a tape of its own (ops == 0) falls back to the first table.
    const uint64_t* sub_ops2;
    const uint2* sub_tab2;
    uint32_t split_level2, pad2_;
};

namespace fhm {
using namespace fhd;

using fhmesh::lerp_pos;

// the tape of the cell with this path (3 bits per level below a leading 1)
__device__ __forceinline__ void mesh_tape(const FhMeshParams& P, uint64_t path, const uint64_t*& ops, uint32_t& len) {
    ops = P.tape; len = P.len;
    if (!P.sub_tab) return;
    if (P.sub_tab2) {
        const int up2 = (63 - __clzll((long long)path)) - 3 * (int)P.split_level2;
        if (up2 > 0) {
            const uint2 e2 = P.sub_tab2[(uint32_t)((path >> up2) - (1ull << (3 * P.split_level2)))];
            if (e2.y) { ops = P.sub_ops2 + e2.x; len = e2.y; return; }
        }
    }
    const int up = (63 - __clzll((long long)path)) - 3 * (int)P.split_level;
    if (up <= 0) return;          // (at the split level itself a cell is still evaluated with the tape it inherited)
    const uint2 e = P.sub_tab[(uint32_t)((path >> up) - (1ull << (3 * P.split_level)))];
    if (e.y) { ops = P.sub_ops + e.x; len = e.y; }
}

// interval evaluation + classification of the cells of one level.  expand: cell i is child (i & 7) of in[i >> 3]
__global__ void __launch_bounds__(WAVE) k_mesh_cells(FhMeshParams P, const FhMeshCell* in, uint32_t n, int expand, FhMeshCell* out, uint32_t* counters /* amb, full, empty */,
                                                      uint32_t out_cap, uint8_t* cls /* per cell: 1 empty 2 full 3 ambiguous, 0 another part's */, uint32_t* slot /* of an ambiguous cell in out */,
                                                      uint32_t child_mask /* expand: the corners evaluated here (0xFF but below the root of a sharded build) */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const uint32_t i = blockIdx.x * WAVE + lane;
    const bool act = i < n && (!expand || ((child_mask >> (i & 7)) & 1u));
    FhMeshCell c;
    {
        const FhMeshCell p = in[expand ? min(i, n - 1) >> 3 : min(i, n - 1)];
        c = p;
        if (expand) {
            const int corner = i & 7;
            for (int k = 0; k < 3; k++) {
                const float mid = (p.b[2 * k] + p.b[2 * k + 1]) / 2.0f;
                if (corner & (1 << k)) { c.b[2 * k] = mid; c.b[2 * k + 1] = p.b[2 * k + 1]; } else { c.b[2 * k] = p.b[2 * k]; c.b[2 * k + 1] = mid; }
            }
            c.path = (p.path << 3) | (uint64_t)corner;
        }
    }
    IV X = iv(c.b[0], c.b[1]), Y = iv(c.b[2], c.b[3]), Z = iv(c.b[4], c.b[5]);
    if (P.has_mat) {
        Mat4 m;
#pragma unroll
        for (int k = 0; k < 16; k++) m.m[k] = P.mat[k];
        xf_interval(m, X, Y, Z, X, Y, Z);
    }
    Regs<IV, WAVE> R{(IV*)smem, lane};
    IV result = iv_nan();
    auto in_iv = [&](uint32_t slot) { const uint32_t kd = P.in_kind[slot]; return kd == 0 ? X : (kd == 1 ? Y : (kd == 2 ? Z : iv1(P.in_value[slot]))); };
    if (P.sub_tab) {       // every lane its cell's tape (a wave's cells mostly share an ancestor: their parents were neighbours)
        const uint64_t* ops; uint32_t len;
        mesh_tape(P, c.path, ops, len);
        for (uint32_t k = 0; k < len; k++) step<IVAL, WAVE, true>(ops[k], R, in_iv, [&](uint32_t, IV v) { result = v; }, [&](int) {});
    } else {
        const ctape_t tape = (ctape_t)P.tape;
        for (uint32_t k = 0; k < P.len; k++) step<IVAL, WAVE, true>(tape[k], R, in_iv, [&](uint32_t, IV v) { result = v; }, [&](int) {});
    }
    const bool full = act && result.hi < 0.0f, empty = act && !full && result.lo > 0.0f, amb = act && !full && !empty;
    const uint64_t am = ballot(amb);
    const uint32_t nf = (uint32_t)__popcll(ballot(full)), ne = (uint32_t)__popcll(ballot(empty));    // (every lane votes: outside the branch)
    uint32_t base = 0;
    if (lane == 0) {
        if (am) base = atomicAdd(&counters[0], (uint32_t)__popcll(am));
        if (nf) atomicAdd(&counters[1], nf);
        if (ne) atomicAdd(&counters[2], ne);
    }
    base = uni(base);
    uint32_t sl = 0xFFFFFFFFu;
    if (amb) {
        sl = base + (uint32_t)__popcll(am & ((1ull << lane) - 1));
        if (sl < out_cap) out[sl] = c;
    }
    if (i < n && cls) { cls[i] = !act ? 0 : (full ? 2 : (empty ? 1 : 3)); slot[i] = sl; }
}

// The choices of the root tape over each of n cells (the ambiguous cells of the split level): choices[cell * n_choices + ordinal] = what the
// interval evaluation decided at that min / max / and / or (vm/mod.rs:436-517), for VmData::simplify on the host (capi_mesh.hpp)
__global__ void __launch_bounds__(WAVE) k_mesh_choices(FhMeshParams P, const FhMeshCell* cells, uint32_t n, uint32_t n_choices, uint8_t* choices) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const uint32_t i = blockIdx.x * WAVE + lane;
    const FhMeshCell c = cells[min(i, n - 1)];
    IV X = iv(c.b[0], c.b[1]), Y = iv(c.b[2], c.b[3]), Z = iv(c.b[4], c.b[5]);
    if (P.has_mat) {
        Mat4 m;
#pragma unroll
        for (int k = 0; k < 16; k++) m.m[k] = P.mat[k];
        xf_interval(m, X, Y, Z, X, Y, Z);
    }
    Regs<IV, WAVE> R{(IV*)smem, lane};
    uint8_t* const mine = choices + (size_t)min(i, n - 1) * n_choices;
    uint32_t ci = 0;
    auto in_iv = [&](uint32_t slot) { const uint32_t kd = P.in_kind[slot]; return kd == 0 ? X : (kd == 1 ? Y : (kd == 2 ? Z : iv1(P.in_value[slot]))); };
    if (P.sub_tab) {       // (the second split: every lane the choices of the tape its cell inherited - the first split's; n_choices = the row length)
        const uint64_t* ops; uint32_t len;
        mesh_tape(P, c.path, ops, len);
        for (uint32_t k = 0; k < len; k++)
            step<IVAL, WAVE, true>(ops[k], R, in_iv, [&](uint32_t, IV) {}, [&](int ch) { if (i < n && ci < n_choices) mine[ci] = (uint8_t)ch; ci++; });
        return;
    }
    const ctape_t tape = (ctape_t)P.tape;
    for (uint32_t k = 0; k < P.len; k++) {
        step<IVAL, WAVE, true>(tape[k], R, in_iv, [&](uint32_t, IV) {}, [&](int ch) { if (i < n) mine[ci] = (uint8_t)ch; ci++; });
    }
}

// f32 value of the tape at this lane's point (lanes evaluate different points of the same leaf)
// (path: the leaf cell's, which says whose simplified tape to take when the octree was split, FhMeshParams)
__device__ __forceinline__ float eval_point(const FhMeshParams& P, const Regs<float, WAVE>& R, float x, float y, float z, uint64_t path) {
    if (P.has_mat) {
        Mat4 m;
#pragma unroll
        for (int k = 0; k < 16; k++) m.m[k] = P.mat[k];
        xf_point(m, x, y, z, x, y, z);
    }
    float result = qnan();
    auto in_f = [&](uint32_t slot) { const uint32_t kd = P.in_kind[slot]; return kd == 0 ? x : (kd == 1 ? y : (kd == 2 ? z : P.in_value[slot])); };
    if (P.sub_tab) {
        const uint64_t* ops; uint32_t len;
        mesh_tape(P, path, ops, len);
        for (uint32_t k = 0; k < len; k++) step<F32, WAVE, true>(ops[k], R, in_f, [&](uint32_t, float v) { result = v; }, [&](int) {});
    } else {
        const ctape_t tape = (ctape_t)P.tape;
        for (uint32_t k = 0; k < P.len; k++) step<F32, WAVE, true>(tape[k], R, in_f, [&](uint32_t, float v) { result = v; }, [&](int) {});
    }
    return result;
}
// ... and its gradient there
template <class GX>
__device__ __forceinline__ GR eval_grad(const FhMeshParams& P, const Regs<GR, WAVE>& G, const GX& gx, const GX& gy, const GX& gz, uint64_t path) {
    GR result = gr1(qnan());
    auto in_g = [&](uint32_t slot) { const uint32_t kd = P.in_kind[slot]; return kd == 0 ? gx : (kd == 1 ? gy : (kd == 2 ? gz : gr1(P.in_value[slot]))); };
    if (P.sub_tab) {
        const uint64_t* ops; uint32_t len;
        mesh_tape(P, path, ops, len);
        for (uint32_t k = 0; k < len; k++) step<GRAD, WAVE, true>(ops[k], G, in_g, [&](uint32_t, GR v) { result = v; }, [&](int) {});
    } else {
        const ctape_t tape = (ctape_t)P.tape;
        for (uint32_t k = 0; k < P.len; k++) step<GRAD, WAVE, true>(tape[k], G, in_g, [&](uint32_t, GR v) { result = v; }, [&](int) {});
    }
    return result;
}

__global__ void __launch_bounds__(WAVE) k_mesh_leaf(FhMeshParams P, const FhMeshCell* cells, uint32_t n, const FhMdcTable* T, FhMeshLeaf* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint16_t s_start[12][3], s_end[12][3];
    const int lane = threadIdx.x;
    const uint32_t li = blockIdx.x;
    if (li >= n) return;
    const FhMeshCell c = cells[li];
    Regs<float, WAVE> R{(float*)smem, lane};
    // corners (cell.rs:196-206): bit 0 x, 1 y, 2 z
    const int cr = lane & 7;
    const float v = eval_point(P, R, (cr & 1) ? c.b[1] : c.b[0], (cr & 2) ? c.b[3] : c.b[2], (cr & 4) ? c.b[5] : c.b[4], c.path);
    const uint32_t mask = (uint32_t)(ballot(v < 0.0f) & 0xFFull);
    FhMeshLeaf* o = &out[li];
    if (lane < 6) o->b[lane] = c.b[lane];
    if (lane == 0) { o->path = c.path; o->mask = mask; o->n_edges = 0; o->n_verts = 0; o->pad = 0; }
    if (mask == 0 || mask == 255) return;      // Cell::Empty / Cell::Full (octree.rs:633-637)
    const uint32_t ne = T->n_edges[mask], nv = T->n_verts[mask];
    if (lane < (int)ne) {
        const int st = T->edge[mask][lane][0], en = T->edge[mask][lane][1];
        const int axis = st ^ en, ai = axis == 1 ? 0 : (axis == 2 ? 1 : 2);
        const uint16_t a = (en & axis) ? 0 : 65535, b = (en & axis) ? 65535 : 0;
        uint16_t p[3] = {0, 0, 0};
        const int i = (ai + 1) % 3, j = (ai + 2) % 3;
        p[i] = (st & (1 << i)) ? 65535 : 0;
        p[j] = (st & (1 << j)) ? 65535 : 0;
        for (int k = 0; k < 3; k++) { s_start[lane][k] = p[k]; s_end[lane][k] = p[k]; }
        s_start[lane][ai] = a; s_end[lane][ai] = b;
    }
    __syncthreads();
    // N-ary search: 4 rounds of 16 points per edge, 4 edges per pass (octree.rs:697-768)
    for (int round = 0; round < 4; round++) {
        for (uint32_t e0 = 0; e0 < ne; e0 += 4) {
            const uint32_t e = e0 + (lane >> 4), j = lane & 15;
            const bool valid = e < ne;
            const uint32_t ee = valid ? e : 0;
            uint32_t p[3];
            for (int k = 0; k < 3; k++) p[k] = ((uint32_t)s_start[ee][k] * (15u - j) + (uint32_t)s_end[ee][k] * j) / 15u;
            const float r = eval_point(P, R, lerp_pos(c.b[0], c.b[1], p[0]), lerp_pos(c.b[2], c.b[3], p[1]), lerp_pos(c.b[4], c.b[5], p[2]), c.path);
            const uint64_t nonneg = ballot(r >= 0.0f);
            const uint32_t m16 = (uint32_t)(nonneg >> ((lane >> 4) * 16)) & 0xFFFFu;
            uint32_t frac = m16 ? (uint32_t)__builtin_ctz(m16) : 16u;
            if (frac == 0) frac = 1;
            if (frac > 15) frac = 15;
            __syncthreads();
            if (valid && j == 0) {
                uint16_t a[3], b[3];
                for (int k = 0; k < 3; k++) {
                    a[k] = (uint16_t)(((uint32_t)s_start[e][k] * (15u - (frac - 1)) + (uint32_t)s_end[e][k] * (frac - 1)) / 15u);
                    b[k] = (uint16_t)(((uint32_t)s_start[e][k] * (15u - frac) + (uint32_t)s_end[e][k] * frac) / 15u);
                }
                for (int k = 0; k < 3; k++) { s_start[e][k] = a[k]; s_end[e][k] = b[k]; }
            }
            __syncthreads();
        }
    }
    // intersections, gradients (octree.rs:771-803)
    {
        const uint32_t e = lane < (int)ne ? lane : 0;
        uint16_t q[3];
        for (int k = 0; k < 3; k++) q[k] = (uint16_t)(((uint32_t)s_start[e][k] + (uint32_t)s_end[e][k]) / 2u);
        const float px = lerp_pos(c.b[0], c.b[1], q[0]), py = lerp_pos(c.b[2], c.b[3], q[1]), pz = lerp_pos(c.b[4], c.b[5], q[2]);
        GR gx = gr(px, 1.0f, 0.0f, 0.0f), gy = gr(py, 0.0f, 1.0f, 0.0f), gz = gr(pz, 0.0f, 0.0f, 1.0f);
        if (P.has_mat) {
            Mat4 m;
#pragma unroll
            for (int k = 0; k < 16; k++) m.m[k] = P.mat[k];
            xf_grad(m, gx, gy, gz, gx, gy, gz);
        }
        __syncthreads();
        Regs<GR, WAVE> G{(GR*)smem, lane};
        const GR result = eval_grad(P, G, gx, gy, gz, c.path);
        if (lane < (int)ne) {
            for (int k = 0; k < 3; k++) o->inter[lane][k] = q[k];
            o->pos[lane][0] = px; o->pos[lane][1] = py; o->pos[lane][2] = pz;
            o->grad[lane][0] = result.dx; o->grad[lane][1] = result.dy; o->grad[lane][2] = result.dz; o->grad[lane][3] = result.v;
        }
    }
    // (the QEFs of the cell vertices: k_mesh_leaf_qef, one LANE per record - here they kept a whole wavefront waiting on lane 0)
    if (lane == 0) { o->n_edges = ne; o->n_verts = nv; }
}

// ---- the same leaf sampling as passes over all leaf cells of a chunk (what fhip_mesh_* run; FHIP_MESH_LEAF_PASSES=0: k_mesh_leaf) --------------------------------------
// k_mesh_leaf gives a leaf a wavefront and leaves most of its lanes idle most of the time: 8 of 64 at the corners, 16 x (edges mod 4)
// in the last pass of a search round, one per edge in the gradient pass - about half of its lane-evaluations are wasted, and every one
// is ~90 f64 operations for a sin / cos.  Here every lane of every pass has a point of its own: corners 8 cells per wave; the edge
// search one wave per FOUR EDGES, of whatever cells (the cells' edges are appended to a list as the corners find them; the 16 lanes of an
// edge keep its bracket in registers through the four rounds: no LDS, no barriers); gradients one lane per edge.  Per lane the arithmetic
// is k_mesh_leaf's, operation for operation: the records are the same, bit for bit.
__global__ void __launch_bounds__(WAVE) k_mesh_corners(FhMeshParams P, const FhMeshCell* cells, uint32_t n, const FhMdcTable* T, FhMeshLeaf* out,
                                                        uint32_t* edge_count, uint32_t* edge_list /* (cell << 4) | edge */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x, cr = lane & 7;
    const uint32_t li = blockIdx.x * 8 + (lane >> 3);
    const bool act = li < n;
    const FhMeshCell c = cells[act ? li : n - 1];
    Regs<float, WAVE> R{(float*)smem, lane};
    const float v = eval_point(P, R, (cr & 1) ? c.b[1] : c.b[0], (cr & 2) ? c.b[3] : c.b[2], (cr & 4) ? c.b[5] : c.b[4], c.path);
    const uint32_t mask = (uint32_t)((ballot(v < 0.0f) >> (lane & ~7)) & 0xFFull);
    const uint32_t ne = (mask == 0 || mask == 255) ? 0u : T->n_edges[mask];
    // this wave's edges: one reservation in the list, the cells' shares in cell order
    uint32_t before = 0, total = 0;
    for (int g = 0; g < 8; g++) {
        const uint32_t ng = (uint32_t)__shfl((int)(act ? ne : 0u), g * 8);
        if (g < (lane >> 3)) before += ng;
        total += ng;
    }
    uint32_t base = 0;
    if (lane == 0 && total) base = atomicAdd(edge_count, total);
    base = uni(base);
    if (!act) return;
    FhMeshLeaf* o = &out[li];
    if (cr < 6) o->b[cr] = c.b[cr];
    if (cr == 0) { o->path = c.path; o->mask = mask; o->n_edges = ne; o->n_verts = ne ? T->n_verts[mask] : 0u; o->pad = 0; }
    for (uint32_t e = cr; e < ne; e += 8) edge_list[base + before + e] = (li << 4) | e;
}

// The corners as passes around the assembly bulk interpreter, like the edge search (capi_mesh.hpp sample_chunk): k_mesh_corner_points writes the
// chunk's 8 n corner points as the interpreter's [slot][8 n] arrays (after xf_point when there is a matrix: what eval_point does), the
// interpreter evaluates them, k_mesh_corner_masks does the rest of k_mesh_corners from the values - same points, same f32 evaluator
// semantics (tests/test_render_random.py drives every opcode through both), so the same masks and records.
__global__ void __launch_bounds__(256) k_mesh_corner_points(FhMeshParams P, const FhMeshCell* cells, uint32_t n, float* vars) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 8u) return;
    const FhMeshCell c = cells[i >> 3];
    const uint32_t cr = i & 7u;
    float x = (cr & 1) ? c.b[1] : c.b[0], y = (cr & 2) ? c.b[3] : c.b[2], z = (cr & 4) ? c.b[5] : c.b[4];
    if (P.has_mat) {
        Mat4 m;
#pragma unroll
        for (int q = 0; q < 16; q++) m.m[q] = P.mat[q];
        xf_point(m, x, y, z, x, y, z);
    }
    const size_t np = (size_t)n * 8;
    for (uint32_t s = 0; s < FH_MAX_INPUTS; s++) {
        const uint32_t kd = P.in_kind[s];
        if (kd < 3) vars[(size_t)s * np + i] = kd == 0 ? x : (kd == 1 ? y : z);
    }
}
__global__ void __launch_bounds__(WAVE) k_mesh_corner_masks(const FhMeshCell* cells, uint32_t n, const float* values, const FhMdcTable* T, FhMeshLeaf* out,
                                                             uint32_t* edge_count, uint32_t* edge_list /* (cell << 4) | edge */) {
    const int lane = threadIdx.x, cr = lane & 7;
    const uint32_t li = blockIdx.x * 8 + (lane >> 3);
    const bool act = li < n;
    const FhMeshCell c = cells[act ? li : n - 1];
    const float v = values[(size_t)(act ? li : n - 1) * 8 + cr];
    const uint32_t mask = (uint32_t)((ballot(v < 0.0f) >> (lane & ~7)) & 0xFFull);
    const uint32_t ne = (mask == 0 || mask == 255) ? 0u : T->n_edges[mask];
    uint32_t before = 0, total = 0;
    for (int g = 0; g < 8; g++) {
        const uint32_t ng = (uint32_t)__shfl((int)(act ? ne : 0u), g * 8);
        if (g < (lane >> 3)) before += ng;
        total += ng;
    }
    uint32_t base = 0;
    if (lane == 0 && total) base = atomicAdd(edge_count, total);
    base = uni(base);
    if (!act) return;
    FhMeshLeaf* o = &out[li];
    if (cr < 6) o->b[cr] = c.b[cr];
    if (cr == 0) { o->path = c.path; o->mask = mask; o->n_edges = ne; o->n_verts = ne ? T->n_verts[mask] : 0u; o->pad = 0; }
    for (uint32_t e = cr; e < ne; e += 8) edge_list[base + before + e] = (li << 4) | e;
}

__global__ void __launch_bounds__(WAVE) k_mesh_edges(FhMeshParams P, const FhMdcTable* T, FhMeshLeaf* out, const uint32_t* edge_list, uint32_t n_edges) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x, grp = lane >> 4;
    const uint32_t j = lane & 15;
    const uint32_t k = blockIdx.x * 4 + grp;
    const bool valid = k < n_edges;
    const uint32_t ent = edge_list[valid ? k : 0];
    FhMeshLeaf* o = &out[ent >> 4];
    const uint32_t e = ent & 15u, mask = o->mask;
    float b[6];
    for (int q = 0; q < 6; q++) b[q] = o->b[q];
    // the edge's end points in u16 cell coordinates (octree.rs:662-695), as k_mesh_leaf sets them up
    uint32_t s[3] = {0, 0, 0}, t[3];
    {
        const int st = T->edge[mask][e][0], en = T->edge[mask][e][1];
        const int axis = st ^ en, ai = axis == 1 ? 0 : (axis == 2 ? 1 : 2);
        const int i1 = (ai + 1) % 3, i2 = (ai + 2) % 3;
        s[i1] = (st & (1 << i1)) ? 65535u : 0u;
        s[i2] = (st & (1 << i2)) ? 65535u : 0u;
        for (int q = 0; q < 3; q++) t[q] = s[q];
        s[ai] = (en & axis) ? 0u : 65535u;
        t[ai] = (en & axis) ? 65535u : 0u;
    }
    Regs<float, WAVE> R{(float*)smem, lane};
    for (int round = 0; round < 4; round++) {       // N-ary search: 16 points per round (octree.rs:697-768)
        uint32_t p[3];
        for (int q = 0; q < 3; q++) p[q] = (s[q] * (15u - j) + t[q] * j) / 15u;
        const float r = eval_point(P, R, lerp_pos(b[0], b[1], p[0]), lerp_pos(b[2], b[3], p[1]), lerp_pos(b[4], b[5], p[2]), o->path);
        const uint32_t m16 = (uint32_t)(ballot(r >= 0.0f) >> (grp * 16)) & 0xFFFFu;
        uint32_t frac = m16 ? (uint32_t)__builtin_ctz(m16) : 16u;
        if (frac == 0) frac = 1;
        if (frac > 15) frac = 15;
        for (int q = 0; q < 3; q++) {
            const uint32_t lo = (uint32_t)(uint16_t)((s[q] * (15u - (frac - 1)) + t[q] * (frac - 1)) / 15u);
            const uint32_t hi = (uint32_t)(uint16_t)((s[q] * (15u - frac) + t[q] * frac) / 15u);
            s[q] = lo; t[q] = hi;
        }
    }
    if (valid && j == 0) {
        uint16_t qq[3];
        for (int q = 0; q < 3; q++) qq[q] = (uint16_t)((s[q] + t[q]) / 2u);
        for (int q = 0; q < 3; q++) o->inter[e][q] = qq[q];
        o->pos[e][0] = lerp_pos(b[0], b[1], qq[0]); o->pos[e][1] = lerp_pos(b[2], b[3], qq[1]); o->pos[e][2] = lerp_pos(b[4], b[5], qq[2]);
    }
}

// ---- ... and with the values from the assembly bulk interpreter (fh_float_eval_*[_t]: 256 or 128 samples per wave at ~22 instructions per op,
// where eval_point's generic interpreter spends ~40 per op on 64): the four rounds as passes over the chunk's edges - this round's 16 sample
// points per edge written as the interpreter's [slot][n] input arrays, the interpreter, the brackets narrowed by the signs (mesh_edges.hpp).
__global__ void __launch_bounds__(256) k_mesh_edge_begin(const FhMdcTable* T, const FhMeshLeaf* recs, const uint32_t* edge_list, uint32_t n_edges, fhmesh::EdgeBracket* br) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_edges) return;
    const uint32_t ent = edge_list[k], mask = recs[ent >> 4].mask, e = ent & 15u;
    br[k] = fhmesh::edge_ends(T->edge[mask][e][0], T->edge[mask][e][1]);
}
// one lane per sample: sample i = 16 * edge + j; vars[slot * n + i] for the tape's x / y / z slots (constant slots are filled once per chunk)
__global__ void __launch_bounds__(256) k_mesh_edge_points(FhMeshParams P, const FhMeshLeaf* recs, const uint32_t* edge_list, const fhmesh::EdgeBracket* br, uint32_t n_edges,
                                                           float* vars, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_edges * 16u) return;
    const uint32_t k = i >> 4, j = i & 15u;
    const FhMeshLeaf* o = &recs[edge_list[k] >> 4];
    uint32_t p[3];
    fhmesh::edge_sample(br[k], j, p);
    float x = lerp_pos(o->b[0], o->b[1], p[0]), y = lerp_pos(o->b[2], o->b[3], p[1]), z = lerp_pos(o->b[4], o->b[5], p[2]);
    if (P.has_mat) {
        Mat4 m;
#pragma unroll
        for (int q = 0; q < 16; q++) m.m[q] = P.mat[q];
        xf_point(m, x, y, z, x, y, z);
    }
    for (uint32_t s = 0; s < FH_MAX_INPUTS; s++) {
        const uint32_t kd = P.in_kind[s];
        if (kd < 3) vars[(size_t)s * n + i] = kd == 0 ? x : (kd == 1 ? y : z);
    }
}
__global__ void __launch_bounds__(256) k_mesh_fill(float* p, float v, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void __launch_bounds__(256) k_mesh_edge_narrow(fhmesh::EdgeBracket* br, const float* values, uint32_t n_edges) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_edges) return;
    uint32_t m16 = 0;
    for (uint32_t j = 0; j < 16; j++) m16 |= (values[(size_t)k * 16 + j] >= 0.0f ? 1u : 0u) << j;
    br[k] = fhmesh::edge_narrow(br[k], m16);
}
__global__ void __launch_bounds__(256) k_mesh_edge_end(FhMeshLeaf* recs, const uint32_t* edge_list, const fhmesh::EdgeBracket* br, uint32_t n_edges) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_edges) return;
    const uint32_t ent = edge_list[k], e = ent & 15u;
    FhMeshLeaf* o = &recs[ent >> 4];
    uint16_t q[3];
    fhmesh::edge_mid(br[k], q);
    for (int a = 0; a < 3; a++) o->inter[e][a] = q[a];
    o->pos[e][0] = lerp_pos(o->b[0], o->b[1], q[0]); o->pos[e][1] = lerp_pos(o->b[2], o->b[3], q[1]); o->pos[e][2] = lerp_pos(o->b[4], o->b[5], q[2]);
}

// gradients at the intersections (octree.rs:771-803): one lane per edge
__global__ void __launch_bounds__(WAVE) k_mesh_grads(FhMeshParams P, FhMeshLeaf* out, const uint32_t* edge_list, uint32_t n_edges) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const uint32_t k = blockIdx.x * WAVE + lane;
    const bool valid = k < n_edges;
    const uint32_t ent = edge_list[valid ? k : 0];
    FhMeshLeaf* o = &out[ent >> 4];
    const uint32_t e = ent & 15u;
    const float px = o->pos[e][0], py = o->pos[e][1], pz = o->pos[e][2];
    GR gx = gr(px, 1.0f, 0.0f, 0.0f), gy = gr(py, 0.0f, 1.0f, 0.0f), gz = gr(pz, 0.0f, 0.0f, 1.0f);
    if (P.has_mat) {
        Mat4 m;
#pragma unroll
        for (int q = 0; q < 16; q++) m.m[q] = P.mat[q];
        xf_grad(m, gx, gy, gz, gx, gy, gz);
    }
    Regs<GR, WAVE> G{(GR*)smem, lane};
    const GR result = eval_grad(P, G, gx, gy, gz, o->path);
    if (valid) { o->grad[e][0] = result.dx; o->grad[e][1] = result.dy; o->grad[e][2] = result.dz; o->grad[e][3] = result.v; }
}

// one QEF per cell vertex (octree.rs:805-848), vertices in order: a NaN gradient snaps the vertex to that intersection and
// stops its loop WITHOUT consuming the edge (the reference's `break` comes before `i += 1`), so the next vertex starts there.
// One lane per leaf record: the Jacobi sweeps of fhq::Qef::solve are a few thousand dependent f64 operations, which at the end of
// k_mesh_leaf occupied a wavefront for the sake of one lane.
__global__ void __launch_bounds__(WAVE) k_mesh_leaf_qef(const FhMdcTable* T, FhMeshLeaf* recs, uint32_t n) {
    const uint32_t li = blockIdx.x * WAVE + threadIdx.x;
    if (li >= n) return;
    FhMeshLeaf* o = &recs[li];
    const uint32_t mask = o->mask;
    if (mask == 0 || mask == 255) return;
    const uint32_t nv = T->n_verts[mask];
    uint32_t i = 0;
    for (uint32_t vtx = 0; vtx < nv; vtx++) {
        fhq::Qef q;
        q.init();
        bool forced = false;
        float pos[3] = {0, 0, 0}, err = -1.0f;
        for (uint32_t k = 0; k < T->per_vert[mask][vtx]; k++) {
            const uint32_t ii = i < 12 ? i : 11;
            float g[4], p[3];
            for (int a = 0; a < 4; a++) g[a] = o->grad[ii][a];
            for (int a = 0; a < 3; a++) p[a] = o->pos[ii][a];
            if (g[0] != g[0] || g[1] != g[1] || g[2] != g[2] || g[3] != g[3]) {
                forced = true;
                for (int a = 0; a < 3; a++) pos[a] = p[a];
                err = -2.0f;
                break;
            }
            q.add(p, g);
            i++;
        }
        if (!forced) q.solve(pos, &err);
        for (int a = 0; a < 3; a++) o->vert[vtx][a] = pos[a];
        o->qef_err[vtx] = err;
    }
}

// ---- the octree assembled on the device (mesh_collapse.hpp: one thread per ambiguous cell / candidate / leaf record and pass) ----------
__global__ void __launch_bounds__(256) k_oct_kind(fhmesh::OctLevel D, fhmesh::OctLevel C, fhmesh::OctLeaves L, const FhMdcTable* T, uint32_t* counter, uint32_t n) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) fhmesh::oct_kind(D, C, L, T, s, counter);
}
__global__ void __launch_bounds__(64) k_oct_collapse(fhmesh::OctLevel D, fhmesh::OctLevel C, fhmesh::OctLeaves L, const FhMdcTable* T, uint32_t n) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) fhmesh::oct_collapse(D, C, L, T, k);
}
__global__ void __launch_bounds__(256) k_oct_place(fhmesh::OctLevel D, fhmesh::OctLevel C, fhmesh::OctLeaves L, const FhMdcTable* T, fhmesh::Cell* cells, fhmesh::V3* verts,
                                                    const float* mat, uint32_t n) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) fhmesh::oct_place(D, C, L, T, s, cells, verts, mat);
}
__global__ void __launch_bounds__(256) k_oct_leaf_verts(fhmesh::OctLeaves L, fhmesh::V3* verts, const float* mat, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fhmesh::oct_leaf_verts(L, i, verts, mat);
}

// the mesh's vertices out of the octree's (for the dual walk on the HOST, option mesh_device_walk 0: it says which - first uses, in its order)
__global__ void __launch_bounds__(256) k_oct_gather(const fhmesh::V3* verts, const uint32_t* idx, fhmesh::V3* out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = verts[idx[i]];
}


// ---- Octree::walk_dual on the device (mesh_walk.hpp: the recursion as level arrays, MeshBuilder's numbering by atomic minima) ----------
__global__ void __launch_bounds__(256) k_walk_count(fhmesh::WalkTree o, const fhmesh::WalkItem* items, uint32_t n, uint32_t* cnt, uint32_t* live) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fhmesh::WalkItem k = items[i];
    cnt[i] = fhmesh::wk_count(o, k);
    if ((k.hdr & 3u) != fhmesh::WK_REC) *live = 1;      // (every writer writes the same value)
}
__global__ void __launch_bounds__(256) k_walk_expand(fhmesh::WalkTree o, const fhmesh::WalkItem* items, uint32_t n, const uint32_t* off, fhmesh::WalkItem* next) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t at = off[i];
    if (off[i + 1] == at) return;
    fhmesh::wk_expand(o, items[i], next + at);
}
__global__ void __launch_bounds__(256) k_walk_first(const fhmesh::WalkItem* recs, uint32_t n, uint32_t* first) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;       // one thread per reference: 5 per record
    if (i >= n * 5u) return;
    atomicMin(&first[recs[i / 5u].a[i % 5u]], i);
}
__global__ void __launch_bounds__(256) k_walk_rec_counts(const fhmesh::WalkItem* recs, uint32_t n, const uint32_t* first, uint32_t* nn, uint32_t* nt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fhmesh::wk_rec_counts(recs[i], i, first, &nn[i], &nt[i]);
}
__global__ void __launch_bounds__(256) k_walk_rec_number(const fhmesh::WalkItem* recs, uint32_t n, uint32_t* first, const uint32_t* vb, const fhmesh::V3* octree_verts, fhmesh::V3* verts) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fhmesh::wk_rec_number(recs[i], i, first, vb[i], octree_verts, verts);
}
__global__ void __launch_bounds__(256) k_walk_rec_triangles(const fhmesh::WalkItem* recs, uint32_t n, const uint32_t* first, const uint32_t* tb, uint64_t* tris) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fhmesh::wk_rec_triangles(recs[i], first, tb[i], tris);
}
// exclusive prefix sums: out[i] = in[0] + .. + in[i - 1] for i in [0, n] (in[n] is not read), a block of 256 threads over 2048 elements;
// sums[block] = the block's total; k_scan_add adds the (already scanned) block totals back
constexpr uint32_t FH_SCAN_PER_BLOCK = 2048;
__global__ void __launch_bounds__(256) k_scan_block(const uint32_t* in, uint32_t n, uint32_t* out, uint32_t* sums) {
    __shared__ uint32_t part[256];
    const uint32_t base = blockIdx.x * FH_SCAN_PER_BLOCK + threadIdx.x * 8u;
    uint32_t v[8], s = 0;
    for (uint32_t k = 0; k < 8; k++) { v[k] = base + k < n ? in[base + k] : 0u; s += v[k]; }
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        const uint32_t add = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;
    for (uint32_t k = 0; k < 8; k++) { if (base + k <= n) out[base + k] = run; run += v[k]; }
    if (threadIdx.x == 255 && sums) sums[blockIdx.x] = part[255];
}
__global__ void __launch_bounds__(256) k_scan_add(uint32_t* out, uint32_t n, const uint32_t* block_off) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n) out[i] += block_off[i / FH_SCAN_PER_BLOCK];
}
}  // namespace fhm
