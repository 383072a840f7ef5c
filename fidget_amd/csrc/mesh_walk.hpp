// Octree::walk_dual (fidget-mesh/src/dc.rs:10-60 with builder.rs MeshBuilder) as data-parallel passes: the recursion
//   cell -> 8 cells, 12 faces, 6 edges;  face -> 4 faces, 4 edges;  edge -> 2 edges, or a quad at four leaves
// unrolled breadth first, level by level, every call replaced by its sub-calls IN THE ORDER THE RECURSION MAKES THEM.  A level is an array
// of items in that order: per item the number of items it leaves at the next level (0: a call that does nothing; 1: an edge call on four
// leaves that emits - it becomes a RECORD - and a record, which is carried along unchanged; 2 / 8 / 26: the sub-calls), an exclusive
// prefix sum of these numbers, and the items written at their places.  Because dead calls vanish and records keep their place among
// their neighbours, the array that holds records only is the sequence of quads the sequential walk emits, in its order.  MeshBuilder's
// numbering of the vertices by first use is then the scheme of host_mesh.hpp ParallelWalker: reference number p = 5 * record + slot (iv,
// vs[0..3]), first[v] = the smallest p that names octree vertex v (atomic minimum), first uses and triangles counted per record, two more
// prefix sums, vertices copied and triangles written at their final places.
//
// Every function here is compiled for the device (mesh.hip: one kernel per pass) and for the host (fhip_debug_walk_dual mode 3 runs the
// passes as plain loops: tests/test_mesh.py compares them with the sequential walk without a GPU).  X says where arrays live and how a
// pass runs, as for oct_assemble (mesh_collapse.hpp).
#pragma once
#include <stdint.h>

#include "mesh_collapse.hpp"

namespace fhmesh {

// CELL_TO_EDGE_TO_VERT (build.rs) as signed bytes: per mask and edge (vertex, intersection) offsets or -1; `any`: the first edge that has one
struct WalkTable {
    int8_t e2v[256][12][2];
    int8_t any[256][2];
};
enum { WK_CELL = 0, WK_FACE = 1, WK_EDGE = 2, WK_REC = 3 };
// hdr: kind | f << 2; a record: kind | winding << 4 | push << 8, a = {iv, vs[0..3]}; a call: a[0..3] = cell references
struct WalkItem {
    uint32_t hdr, a[5];
};
// a cell reference: (block * 8 + child) << 5 | depth (5 bits: fhip_mesh_build accepts depths up to 20; 27 bits of cell index for
// < 2^24 blocks; all ones would need depth 31, so it is free for the root); the root is not stored in a block
constexpr uint32_t WREF_ROOT = 0xFFFFFFFFu;
constexpr uint32_t WALK_MAX_BLOCKS = 1u << 24, WALK_MAX_DEPTH = 30, WREF_DEPTH_BITS = 5;
struct WalkTree {
    const Cell* cells;      // blocks of eight
    Cell root;
    const WalkTable* T;
};
FHQ_HD static inline Cell wk_at(const WalkTree& o, uint32_t r) { return r == WREF_ROOT ? o.root : o.cells[r >> WREF_DEPTH_BITS]; }
FHQ_HD static inline uint32_t wk_depth(uint32_t r) { return r == WREF_ROOT ? 0u : (r & ((1u << WREF_DEPTH_BITS) - 1u)); }
FHQ_HD static inline bool wk_is_leaf(const WalkTree& o, uint32_t r) { const uint8_t k = wk_at(o, r).kind; return k == C_LEAF || k == C_FULL || k == C_EMPTY; }
FHQ_HD static inline uint32_t wk_child(const WalkTree& o, uint32_t r, int i) {
    const Cell x = wk_at(o, r);
    if (x.kind != C_BRANCH) return r;
    return ((x.index * 8u + (uint32_t)i) << WREF_DEPTH_BITS) | (wk_depth(r) + 1u);
}
FHQ_HD static inline void wk_frame(int f, int* t, int* u, int* v) {
    *t = f == 0 ? AX : (f == 1 ? AY : AZ);
    *u = axis_next(*t);
    *v = axis_next(*u);
}
FHQ_HD static inline void wk_edge_corners(int e, int* start, int* end) {   // types.rs Edge::corners
    int t, u, v;
    wk_frame(e / 4, &t, &u, &v);
    const int uu = ((e % 4) % 2 != 0) ? u : 0, vv = ((e % 4) / 2 != 0) ? v : 0;
    *start = uu | vv; *end = t | uu | vv;
}
// dc.rs edge on four leaves (host_mesh.hpp Walker::edge): false = nothing is emitted
FHQ_HD static inline bool wk_emit(const WalkTree& o, const WalkItem& k, WalkItem* out) {
    Cell leafs[4];
    for (int i = 0; i < 4; i++) { leafs[i] = wk_at(o, k.a[i]); if (leafs[i].kind != C_LEAF) return false; }
    int deepest = 0;     // Iterator::max_by_key: the last maximum
    for (int i = 0; i < 4; i++) if (wk_depth(k.a[i]) >= wk_depth(k.a[deepest])) deepest = i;
    int t, u, v;
    wk_frame((int)((k.hdr >> 2) & 3u), &t, &u, &v);
    const int ti = axis_index(t);
    const int edges[4] = {ti * 4 + 3, ti * 4 + 2, ti * 4 + 0, ti * 4 + 1};
    int s0, e0;
    wk_edge_corners(edges[deepest], &s0, &e0);
    const bool st = !((leafs[deepest].mask >> s0) & 1), en = !((leafs[deepest].mask >> e0) & 1);
    if (st == en) return false;
    int vv[4][2];
    for (int i = 0; i < 4; i++) {
        const int8_t* e = wk_depth(k.a[i]) == wk_depth(k.a[deepest]) ? o.T->e2v[leafs[i].mask][edges[i]] : o.T->any[leafs[i].mask];
        vv[i][0] = e[0]; vv[i][1] = e[1];
        if (vv[i][0] < 0) return false;
    }
    const uint32_t winding = st ? 3u : 1u;
    uint32_t push = 0;
    for (uint32_t j = 0; j < 4; j++)
        if (k.a[j] != k.a[(j + winding) % 4]) push |= 1u << j;      // (two references are equal exactly when they name the same cell)
    if (out) {
        out->hdr = WK_REC | winding << 4 | push << 8;
        out->a[0] = leafs[deepest].index + (uint32_t)vv[deepest][1];
        for (int i = 0; i < 4; i++) out->a[1 + i] = leafs[i].index + (uint32_t)vv[i][0];
    }
    return true;
}
// items this item leaves at the next level
FHQ_HD static inline uint32_t wk_count(const WalkTree& o, const WalkItem& k) {
    const uint32_t kind = k.hdr & 3u;
    if (kind == WK_REC) return 1;
    if (kind == WK_CELL) return wk_at(o, k.a[0]).kind == C_BRANCH ? 26u : 0u;
    if (kind == WK_FACE) return (wk_is_leaf(o, k.a[0]) && wk_is_leaf(o, k.a[1])) ? 0u : 8u;
    bool all_leaf = true;
    for (int i = 0; i < 4; i++) all_leaf = all_leaf && wk_is_leaf(o, k.a[i]);
    if (!all_leaf) return 2;
    return wk_emit(o, k, nullptr) ? 1u : 0u;
}
// ... written to out[0 .. wk_count): the sub-calls in the recursion's order (host_mesh.hpp ParallelWalker::expand)
FHQ_HD static inline void wk_expand(const WalkTree& o, const WalkItem& k, WalkItem* out) {
    const uint32_t kind = k.hdr & 3u;
    int n = 0;
    auto cell = [&](uint32_t a) { WalkItem& c = out[n++]; c.hdr = WK_CELL; c.a[0] = a; c.a[1] = c.a[2] = c.a[3] = c.a[4] = 0; };
    auto face = [&](int f, uint32_t lo, uint32_t hi) { WalkItem& c = out[n++]; c.hdr = WK_FACE | (uint32_t)f << 2; c.a[0] = lo; c.a[1] = hi; c.a[2] = c.a[3] = c.a[4] = 0; };
    auto edge = [&](int f, uint32_t a, uint32_t b, uint32_t c2, uint32_t d) {
        WalkItem& c = out[n++]; c.hdr = WK_EDGE | (uint32_t)f << 2; c.a[0] = a; c.a[1] = b; c.a[2] = c2; c.a[3] = d; c.a[4] = 0;
    };
    if (kind == WK_REC) { out[0] = k; return; }
    if (kind == WK_CELL) {
        const uint32_t c = k.a[0];
        if (wk_at(o, c).kind != C_BRANCH) return;
        for (int i = 0; i < 8; i++) cell(wk_child(o, c, i));
        for (int f = 0; f < 3; f++) {
            int t, u, v;
            wk_frame(f, &t, &u, &v);
            const int qs[4] = {0, u, v, u | v};
            for (int q : qs) face(f, wk_child(o, c, q), wk_child(o, c, q | t));
        }
        for (int i = 0; i < 2; i++) {
            const int x = i ? AX : 0, y = i ? AY : 0, z = i ? AZ : 0;
            edge(0, wk_child(o, c, x), wk_child(o, c, x | AY), wk_child(o, c, x | AY | AZ), wk_child(o, c, x | AZ));
            edge(1, wk_child(o, c, y), wk_child(o, c, y | AZ), wk_child(o, c, y | AX | AZ), wk_child(o, c, y | AX));
            edge(2, wk_child(o, c, z), wk_child(o, c, z | AX), wk_child(o, c, z | AX | AY), wk_child(o, c, z | AY));
        }
        return;
    }
    const int f = (int)((k.hdr >> 2) & 3u);
    int t, u, v;
    wk_frame(f, &t, &u, &v);
    if (kind == WK_FACE) {
        const uint32_t lo = k.a[0], hi = k.a[1];
        if (wk_is_leaf(o, lo) && wk_is_leaf(o, hi)) return;
        face(f, wk_child(o, lo, t), wk_child(o, hi, 0));
        face(f, wk_child(o, lo, t | u), wk_child(o, hi, u));
        face(f, wk_child(o, lo, t | v), wk_child(o, hi, v));
        face(f, wk_child(o, lo, t | u | v), wk_child(o, hi, u | v));
        for (int i = 0; i < 2; i++) {
            const int ui = i ? u : 0, vi = i ? v : 0;
            edge((f + 1) % 3, wk_child(o, lo, ui | t), wk_child(o, lo, ui | v | t), wk_child(o, hi, ui | v), wk_child(o, hi, ui));
            edge((f + 2) % 3, wk_child(o, lo, vi | t), wk_child(o, hi, vi), wk_child(o, hi, vi | u), wk_child(o, lo, vi | u | t));
        }
        return;
    }
    bool all_leaf = true;
    for (int i = 0; i < 4; i++) all_leaf = all_leaf && wk_is_leaf(o, k.a[i]);
    if (all_leaf) { (void)wk_emit(o, k, out); return; }
    for (int i = 0; i < 2; i++) {
        const int ti = i ? t : 0;
        edge(f, wk_child(o, k.a[0], ti | u | v), wk_child(o, k.a[1], ti | v), wk_child(o, k.a[2], ti), wk_child(o, k.a[3], ti | u));
    }
}

// ---- the numbering of MeshBuilder (builder.rs): per-record functions -----------------------------------------------------
constexpr uint32_t WALK_TAG = 0x80000000u;
// first uses and triangles of record r (after the atomic minima)
FHQ_HD static inline void wk_rec_counts(const WalkItem& k, uint32_t r, const uint32_t* first, uint32_t* n_new, uint32_t* n_tri) {
    uint32_t nn = 0;
    for (uint32_t s = 0; s < 5; s++) nn += first[k.a[s]] == 5u * r + s;
    uint32_t p = (k.hdr >> 8) & 15u, nt = 0;
    for (; p; p &= p - 1) nt++;
    *n_new = nn; *n_tri = nt;
}
// record r numbers its new vertices vbase .. and copies them out; their entries of `first` become WALK_TAG | number
FHQ_HD static inline void wk_rec_number(const WalkItem& k, uint32_t r, uint32_t* first, uint32_t vbase, const V3* octree_verts, V3* verts) {
    for (uint32_t s = 0; s < 5; s++) {
        const uint32_t v = k.a[s];
        if (first[v] == 5u * r + s) {
            verts[vbase] = octree_verts[v];
            first[v] = WALK_TAG | vbase++;
        }
    }
}
// ... and its triangles, once every vertex has its number (dc.rs:158-170)
FHQ_HD static inline void wk_rec_triangles(const WalkItem& k, const uint32_t* first, uint32_t tbase, uint64_t* tris) {
    const uint32_t winding = (k.hdr >> 4) & 15u, push = (k.hdr >> 8) & 15u;
    const uint64_t iv = first[k.a[0]] & ~WALK_TAG;
    for (uint32_t j = 0; j < 4; j++) {
        if (!((push >> j) & 1u)) continue;
        uint64_t* t = tris + 3 * (size_t)tbase++;
        t[0] = first[k.a[1 + j]] & ~WALK_TAG;
        t[1] = first[k.a[1 + (j + winding) % 4]] & ~WALK_TAG;
        t[2] = iv;
    }
}

// ---- the passes in order ---------------------------------------------------------------------------------------------------
// X: alloc(bytes) -> pointer or null, free(p), zero(p, bytes), fill_ff(p, bytes), read(dst, src, bytes) (a synchronising copy to the
// host), write(dst, src, bytes) (host -> where the arrays live), scan(in, n, out) -> false on failure (out[0 .. n] = exclusive prefix sums of
// in[0 .. n), out[n] = the total), and the passes count / expand / first_min / rec_counts / rec_number / rec_triangles over n items.
struct WalkOut {
    V3* verts = nullptr;           // the mesh's vertices and triangles, where X keeps arrays (the caller frees them through X)
    uint64_t* tris = nullptr;
    uint32_t n_verts = 0, n_tris = 0, n_records = 0, levels = 0;
    uint64_t items = 0;            // calls + carried records over all levels
};
enum { WALK_OK = 0, WALK_NO_MEMORY = 1, WALK_TOO_BIG = 2 };
template <class X>
static inline int walk_dual_passes(X& x, const Cell* cells, uint32_t n_blocks, Cell root, const V3* octree_verts, uint32_t n_octree_verts, const WalkTable* table,
                                   WalkOut* out) {
    *out = WalkOut();
    if (n_blocks >= WALK_MAX_BLOCKS) return WALK_TOO_BIG;
    WalkTree o{cells, root, table};
    WalkItem* cur = (WalkItem*)x.alloc(sizeof(WalkItem));
    uint32_t* live = (uint32_t*)x.alloc(4);
    if (!cur || !live) return WALK_NO_MEMORY;
    WalkItem first_call{WK_CELL, {WREF_ROOT, 0, 0, 0, 0}};
    x.write(cur, &first_call, sizeof(first_call));
    uint32_t n = 1;
    for (uint32_t level = 0;; level++) {
        if (level > 2 * WALK_MAX_DEPTH + 4) return WALK_TOO_BIG;      // (cannot be: every level descends the octree)
        uint32_t* cnt = (uint32_t*)x.alloc((size_t)(n + 1) * 4);
        uint32_t* off = (uint32_t*)x.alloc((size_t)(n + 1) * 4);
        if (!cnt || !off) return WALK_NO_MEMORY;
        x.zero(live, 4);
        x.count(o, cur, n, cnt, live);
        uint32_t n_live = 0;
        x.read(&n_live, live, 4);
        out->items += n;
        out->levels = level + 1;
        if (!n_live) { x.free(cnt); x.free(off); break; }       // records only: the walk's quads in its order
        if (!x.scan(cnt, n, off)) return WALK_NO_MEMORY;
        uint32_t total = 0;
        x.read(&total, off + n, 4);
        WalkItem* next = (WalkItem*)x.alloc((size_t)(total ? total : 1) * sizeof(WalkItem));
        if (!next) return WALK_NO_MEMORY;
        x.expand(o, cur, n, off, next);
        x.free(cnt); x.free(off); x.free(cur);
        cur = next; n = total;
        if ((uint64_t)n * 5 >= WALK_TAG) return WALK_TOO_BIG;     // (reference numbers are 31 bits wide)
        if (!n) break;
    }
    x.free(live);
    out->n_records = n;
    if (!n) { x.free(cur); return WALK_OK; }
    uint32_t* first = (uint32_t*)x.alloc((size_t)(n_octree_verts ? n_octree_verts : 1) * 4);
    uint32_t* nn = (uint32_t*)x.alloc((size_t)(n + 1) * 4);
    uint32_t* nt = (uint32_t*)x.alloc((size_t)(n + 1) * 4);
    uint32_t* vb = (uint32_t*)x.alloc((size_t)(n + 1) * 4);
    uint32_t* tb = (uint32_t*)x.alloc((size_t)(n + 1) * 4);
    if (!first || !nn || !nt || !vb || !tb) return WALK_NO_MEMORY;
    x.fill_ff(first, (size_t)n_octree_verts * 4);
    x.first_min(cur, n, first);
    x.rec_counts(cur, n, first, nn, nt);
    if (!x.scan(nn, n, vb) || !x.scan(nt, n, tb)) return WALK_NO_MEMORY;
    uint32_t tv = 0, tt = 0;
    x.read(&tv, vb + n, 4);
    x.read(&tt, tb + n, 4);
    out->n_verts = tv; out->n_tris = tt;
    out->verts = (V3*)x.alloc((size_t)(tv ? tv : 1) * sizeof(V3));
    out->tris = (uint64_t*)x.alloc((size_t)(tt ? tt : 1) * 24);
    if (!out->verts || !out->tris) return WALK_NO_MEMORY;
    x.rec_number(cur, n, first, vb, octree_verts, out->verts);
    x.rec_triangles(cur, n, first, tb, out->tris);
    x.free(first); x.free(nn); x.free(nt); x.free(vb); x.free(tb); x.free(cur);
    return WALK_OK;
}

}  // namespace fhmesh
