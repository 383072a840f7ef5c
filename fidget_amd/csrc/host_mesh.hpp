// Host side of the meshing path: what fidget-mesh does with the evaluation results and has no evaluation in it.
//
//   * the Manifold Dual Contouring connectivity tables (fidget-mesh/build.rs:26-160);
//   * assembly of the octree from the device's per-level cell classes and leaf records, bottom-up as Octree::recurse unwinds
//     (octree.rs:556-583): eight children -> check_done / try_collapse / collapsible (octree.rs:256-470) with the merged Hermite
//     data of LeafHermiteData (octree.rs:866-1035);
//   * the dual walk that emits the triangles (dc.rs, builder.rs).
//
// Integer / topological code, plus the QEF of a collapsed cell (mesh_qef.hpp, the definition the device kernel uses too).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <functional>
#include <memory>
#include <chrono>
#include <thread>
#include <utility>
#include <vector>

#include "mesh_collapse.hpp"
#include "mesh_qef.hpp"

namespace fhmesh {

// (AX / AY / AZ, axis_next, axis_index, to_undirected, Cell, V3, LeafIntersection, Hermite, collapsible_children: mesh_collapse.hpp)

struct Tables {
    std::vector<std::vector<std::pair<uint8_t, uint8_t>>> v2e[256];   // CELL_TO_VERT_TO_EDGES
    int e2v[256][12][2];                                              // CELL_TO_EDGE_TO_VERT: (vertex, intersection) offsets or -1
};
// build.rs:26-160
static inline Tables* build_tables() {
    Tables* t = new Tables();
    for (int i = 0; i < 256; i++) {
        int region_of[2][8];
        for (int pass = 0; pass < 2; pass++) {
            int* r = region_of[pass];
            for (int j = 0; j < 8; j++) r[j] = 1 << j;
            for (bool changed = true; changed;) {
                changed = false;
                for (int f = 0; f < 8; f++) {
                    if ((((i >> f) & 1) != 0) != (pass == 0)) continue;
                    for (int axis : {AX, AY, AZ}) {
                        const int g = f ^ axis;
                        if ((((i >> g) & 1) != 0) != (pass == 0)) continue;
                        const int v = r[f] | r[g];
                        if (r[f] != v || r[g] != v) { r[f] = v; r[g] = v; changed = true; }
                    }
                }
            }
        }
        std::vector<int> fr, er;
        for (int j = 0; j < 8; j++) ((i >> j) & 1 ? fr : er).push_back(region_of[(i >> j) & 1 ? 0 : 1][j]);
        for (auto* v : {&fr, &er}) { std::sort(v->begin(), v->end()); v->erase(std::unique(v->begin(), v->end()), v->end()); }
        int regions[8], ri = 0;
        for (auto* rs : {&fr, &er})
            for (int r : *rs) { for (int j = 0; j < 8; j++) if (r & (1 << j)) regions[j] = ri; ri++; }
        std::vector<std::pair<int, std::vector<std::pair<uint8_t, uint8_t>>>> verts;
        for (int rev = 0; rev < 2; rev++)
            for (int tt : {AX, AY, AZ}) {
                const int u = axis_next(tt), v = axis_next(u);
                for (int b = 0; b < 2; b++)
                    for (int a = 0; a < 2; a++) {
                        int start = (a * u) | (b * v), end = start | tt;
                        if (rev) std::swap(start, end);
                        if (!(((i >> start) & 1) && !((i >> end) & 1))) continue;
                        auto it = std::find_if(verts.begin(), verts.end(), [&](auto& kv) { return kv.first == regions[start]; });
                        if (it == verts.end()) { verts.push_back({regions[start], {}}); it = verts.end() - 1; }
                        it->second.push_back({(uint8_t)start, (uint8_t)end});
                    }
            }
        std::sort(verts.begin(), verts.end(), [](auto& a, auto& b) { return a.first < b.first; });
        for (int e = 0; e < 12; e++) t->e2v[i][e][0] = t->e2v[i][e][1] = -1;
        const int vert_count = (int)verts.size();
        int n = 0;
        for (int vi = 0; vi < vert_count; vi++) {
            t->v2e[i].push_back(verts[vi].second);
            for (auto& se : verts[vi].second) {
                const int tt = se.first ^ se.second, u = axis_next(tt), v = axis_next(u);
                const int edge = axis_index(tt) * 4 + ((se.first & u) ? 1 : 0) + ((se.first & v) ? 2 : 0);
                t->e2v[i][edge][0] = vi;
                t->e2v[i][edge][1] = vert_count + n++;
            }
        }
    }
    return t;
}
static inline const Tables& tables() {
    static const Tables* const T = build_tables();     // (initialised once, thread-safe: contexts on several host threads build meshes)
    return *T;
}
static inline void edge_corners(int e, int* start, int* end) {   // types.rs Edge::corners
    static const int FR[3][3] = {{AX, AY, AZ}, {AY, AZ, AX}, {AZ, AX, AY}};
    const int t = FR[e / 4][0], u = ((e % 4) % 2 != 0) ? FR[e / 4][1] : 0, v = ((e % 4) / 2 != 0) ? FR[e / 4][2] : 0;
    *start = u | v; *end = t | u | v;
}

struct CellRef {       // CellIndex (cell.rs:87-110) without the bounds: where the cell is stored, how deep it is
    int64_t ci = -1;
    uint8_t cj = 0;
    uint32_t depth = 0;
};

// (vector growth without value-initialisation: the parallel assembly makes room for a subtree's vertices first and fills it in
// from another thread later - zeroing two gigabytes in between, page by page, was a third of its time)
template <class T>
struct default_init_allocator : std::allocator<T> {
    template <class U> struct rebind { using other = default_init_allocator<U>; };
    default_init_allocator() = default;
    template <class U> default_init_allocator(const default_init_allocator<U>&) {}
    template <class U> void construct(U* p) { ::new ((void*)p) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};
using VertVec = std::vector<V3, default_init_allocator<V3>>;
using TriVec = std::vector<std::array<uint64_t, 3>, default_init_allocator<std::array<uint64_t, 3>>>;
struct Octree {
    Cell root;
    std::vector<std::array<Cell, 8>> cells;
    std::vector<V3, default_init_allocator<V3>> verts;
    // fhip_mesh_build: the arrays as the device assembled them, in the context's pinned landing area, instead of the vectors
    const std::array<Cell, 8>* cells_view = nullptr;
    const V3* verts_view = nullptr;
    size_t n_cells_view = 0, n_verts_view = 0;
    Cell& at(const CellRef& c) { return c.ci < 0 ? root : cells[(size_t)c.ci][c.cj]; }
    const Cell& at(const CellRef& c) const { return c.ci < 0 ? root : (cells_view ? cells_view[(size_t)c.ci][c.cj] : cells[(size_t)c.ci][c.cj]); }
    const V3& vert(size_t v) const { return cells_view ? verts_view[v] : verts[v]; }      // (verts_view may be null: then the walk gathers, ParallelWalker::gather)
    size_t n_verts() const { return cells_view ? n_verts_view : verts.size(); }
    size_t n_cells() const { return cells_view ? n_cells_view : cells.size(); }
    bool is_leaf(const CellRef& c) const { const uint8_t k = at(c).kind; return k == C_LEAF || k == C_FULL || k == C_EMPTY; }
    CellRef child(const CellRef& c, int i) const {
        const Cell& x = at(c);
        if (x.kind != C_BRANCH) return c;
        CellRef r; r.ci = x.index; r.cj = (uint8_t)i; r.depth = c.depth + 1;
        return r;
    }
    // octree.rs:389-470
    bool collapsible(size_t rootc, uint8_t* out_mask) const {
        const auto& cs = cells[rootc];
        static const struct NV { uint8_t n[256]; NV() { for (int m = 0; m < 256; m++) n[m] = (uint8_t)tables().v2e[m].size(); } } nv;
        uint8_t kind[8], mask[8];
        for (int i = 0; i < 8; i++) { kind[i] = cs[i].kind; mask[i] = cs[i].mask; }
        return collapsible_children(kind, mask, nv.n, out_mask);
    }
    // octree.rs:256-340; bounds = this cell's x.lo x.hi y.lo y.hi z.lo z.hi
    Cell check_done(const float* bounds, size_t index, const Hermite* hd, Hermite* hermite) {
        int full = 0, empty = 0;
        for (int i = 0; i < 8; i++) {
            const uint8_t k = cells[index][i].kind;
            if (k == C_FULL) full++;
            else if (k == C_EMPTY) empty++;
            else if (k == C_BRANCH) { Cell c; c.kind = C_BRANCH; c.index = (uint32_t)index; return c; }
        }
        Cell out;
        if (full == 8) out.kind = C_FULL;
        else if (empty == 8) out.kind = C_EMPTY;
        else {
            uint8_t mask;
            bool ok = collapsible(index, &mask) && Hermite::merge(hd, hermite);
            float pos[3], err = 0;
            if (ok) {
                hermite->solve(pos, &err);
                bool inside = true;
                for (int k = 0; k < 3; k++) inside &= pos[k] >= bounds[2 * k] && pos[k] <= bounds[2 * k + 1];
                if (err >= hermite->qef_err * 2.0f || !inside) ok = false;
            }
            if (ok) {
                hermite->qef_err = err;
                const size_t vi = verts.size();
                verts.push_back(V3{pos[0], pos[1], pos[2]});
                for (auto& e : tables().v2e[mask][0]) {
                    const LeafIntersection& li = hermite->inter[to_undirected(e.first, e.second)];
                    verts.push_back(V3{li.pos[0], li.pos[1], li.pos[2]});
                }
                out.kind = C_LEAF; out.mask = mask; out.index = (uint32_t)vi;
            } else { out.kind = C_BRANCH; out.index = (uint32_t)index; }
        }
        if (out.kind != C_BRANCH) {
            if (index == cells.size() - 1) cells.resize(index);
            else cells[index] = std::array<Cell, 8>();
        }
        return out;
    }
};

// dc.rs / builder.rs
struct Walker {
    const Octree& o;
    TriVec triangles;
    VertVec vertices;
    std::vector<size_t> map;
    explicit Walker(const Octree& oc) : o(oc) {}
    static void frame(int f, int* t, int* u, int* v) { static const int FR[3][3] = {{AX, AY, AZ}, {AY, AZ, AX}, {AZ, AX, AY}}; *t = FR[f][0]; *u = FR[f][1]; *v = FR[f][2]; }
    size_t vertex(size_t v) {
        if (v >= map.size()) map.resize(v + 1, (size_t)-1);
        if (map[v] == (size_t)-1) { map[v] = vertices.size(); vertices.push_back(o.vert(v)); }
        return map[v];
    }
    void cell(const CellRef& c) {
        if (o.at(c).kind != C_BRANCH) return;
        for (int i = 0; i < 8; i++) cell(o.child(c, i));
        for (int f = 0; f < 3; f++) {
            int t, u, v; frame(f, &t, &u, &v);
            for (int k : {0, u, v, u | v}) face(f, o.child(c, k), o.child(c, k | t));
        }
        for (int i = 0; i < 2; i++) {
            const int x = i ? AX : 0, y = i ? AY : 0, z = i ? AZ : 0;
            edge(0, o.child(c, x), o.child(c, x | AY), o.child(c, x | AY | AZ), o.child(c, x | AZ));
            edge(1, o.child(c, y), o.child(c, y | AZ), o.child(c, y | AX | AZ), o.child(c, y | AX));
            edge(2, o.child(c, z), o.child(c, z | AX), o.child(c, z | AX | AY), o.child(c, z | AY));
        }
    }
    void face(int f, const CellRef& lo, const CellRef& hi) {
        if (o.is_leaf(lo) && o.is_leaf(hi)) return;
        int t, u, v; frame(f, &t, &u, &v);
        face(f, o.child(lo, t), o.child(hi, 0));
        face(f, o.child(lo, t | u), o.child(hi, u));
        face(f, o.child(lo, t | v), o.child(hi, v));
        face(f, o.child(lo, t | u | v), o.child(hi, u | v));
        for (int i = 0; i < 2; i++) {
            const int ui = i ? u : 0, vi = i ? v : 0;
            edge((f + 1) % 3, o.child(lo, ui | t), o.child(lo, ui | v | t), o.child(hi, ui | v), o.child(hi, ui));
            edge((f + 2) % 3, o.child(lo, vi | t), o.child(hi, vi), o.child(hi, vi | u), o.child(lo, vi | u | t));
        }
    }
    void edge(int f, const CellRef& a, const CellRef& b, const CellRef& c, const CellRef& d) {
        const CellRef cs[4] = {a, b, c, d};
        bool all_leaf = true;
        for (auto& x : cs) all_leaf &= o.is_leaf(x);
        int t, u, v; frame(f, &t, &u, &v);
        if (!all_leaf) {
            for (int i = 0; i < 2; i++) {
                const int ti = i ? t : 0;
                edge(f, o.child(a, ti | u | v), o.child(b, ti | v), o.child(c, ti), o.child(d, ti | u));
            }
            return;
        }
        Cell leafs[4];
        for (int i = 0; i < 4; i++) { leafs[i] = o.at(cs[i]); if (leafs[i].kind != C_LEAF) return; }
        int deepest = 0;     // Iterator::max_by_key: the last maximum
        for (int i = 0; i < 4; i++) if (cs[i].depth >= cs[deepest].depth) deepest = i;
        const int ti = axis_index(t);
        const int edges[4] = {ti * 4 + 3, ti * 4 + 2, ti * 4 + 0, ti * 4 + 1};
        int s0, e0;
        edge_corners(edges[deepest], &s0, &e0);
        const bool st = !((leafs[deepest].mask >> s0) & 1), en = !((leafs[deepest].mask >> e0) & 1);
        if (st == en) return;
        const Tables& T = tables();
        int vv[4][2];
        for (int i = 0; i < 4; i++) {
            vv[i][0] = vv[i][1] = -1;
            if (cs[i].depth == cs[deepest].depth) { vv[i][0] = T.e2v[leafs[i].mask][edges[i]][0]; vv[i][1] = T.e2v[leafs[i].mask][edges[i]][1]; }
            else for (int j = 0; j < 12; j++) if (T.e2v[leafs[i].mask][j][0] >= 0) { vv[i][0] = T.e2v[leafs[i].mask][j][0]; vv[i][1] = T.e2v[leafs[i].mask][j][1]; break; }
            if (vv[i][0] < 0) return;
        }
        const size_t iv = vertex(leafs[deepest].index + (size_t)vv[deepest][1]);
        size_t vs[4];
        for (int i = 0; i < 4; i++) vs[i] = vertex(leafs[i].index + (size_t)vv[i][0]);
        const int winding = st ? 3 : 1;
        for (int j = 0; j < 4; j++) {
            const CellRef &p = cs[j], &q = cs[(j + winding) % 4];
            if (p.ci != q.ci || p.cj != q.cj) triangles.push_back({vs[j], vs[(j + winding) % 4], iv});
        }
    }
};

// ---- host threads ------------------------------------------------------------------------------------------------
// (FHIP_MESH_THREADS overrides; the work items below are independent subtrees / sub-walks, taken from a shared counter)
static inline unsigned mesh_threads() {
    if (const char* e = getenv("FHIP_MESH_THREADS")) return (unsigned)std::max(1, atoi(e));
    const unsigned hc = std::thread::hardware_concurrency();
    return std::max(1u, std::min(hc ? hc : 1u, 32u));      // (measured on 2 x 64 cores: 32 threads beat 64, 128 and 256 - the work is bound by memory, not by arithmetic)
}
static inline void parallel_for(size_t n, const std::function<void(size_t)>& f) {
    const unsigned nt = (unsigned)std::min<size_t>(mesh_threads(), n);
    if (nt <= 1) { for (size_t i = 0; i < n; i++) f(i); return; }
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < n;) f(i); });
    for (auto& t : th) t.join();
}

// The dual walk split into independent pieces with the sequential walk's output, triangle for triangle.  The recursion
// of Walker (cell -> cells, faces, edges; face -> faces, edges; edge -> edges) is unrolled breadth first, every call replaced
// by its sub-calls IN THE ORDER THE RECURSION MAKES THEM, until there are enough calls to keep the threads busy; each
// remaining call is then walked on its own and records what it would emit - per edge the five octree vertices in the order
// Walker::edge asks MeshBuilder for them, and which of the four triangles exist; a final sequential pass over the records in
// call order numbers the vertices by first use (builder.rs) and writes the triangles.
struct ParallelWalker {
    const Octree& o;
    TriVec triangles;
    VertVec vertices;
    explicit ParallelWalker(const Octree& oc) : o(oc) {}
    // The octree's vertices may not be on the host at all (fhip_mesh_build: 191 M of them at depth 10, of which the mesh uses 7.5 M):
    // then `gather` fetches the mesh's - out[i] = octree vertex idx[i] - once the walk knows which they are.  false: it failed.
    std::function<bool(const uint32_t* idx, size_t n, V3* out)> gather;
    bool gather_failed = false;
    uint32_t** scratch = nullptr;      // the caller's first-use table and its capacity in entries, kept between runs (or none)
    size_t* scratch_cap = nullptr;
    struct Call { uint8_t kind, f; CellRef c[4]; };        // kind 0 cell(c0), 1 face(f, c0, c1), 2 edge(f, c0..c3)
    struct Rec { uint64_t iv, vs[4]; uint8_t winding, push; };
    static void frame(int f, int* t, int* u, int* v) { Walker::frame(f, t, u, v); }
    // the sub-calls of one call, in order; false: the call emits (or does nothing) itself
    struct CallBuf { Call c[26]; int n = 0; void push_back(const Call& k) { c[n++] = k; } };
    bool expand(const Call& k, CallBuf& out) const {
        auto cell = [&](const CellRef& a) { Call c{}; c.kind = 0; c.c[0] = a; out.push_back(c); };
        auto face = [&](int f, const CellRef& lo, const CellRef& hi) { Call c{}; c.kind = 1; c.f = (uint8_t)f; c.c[0] = lo; c.c[1] = hi; out.push_back(c); };
        auto edge = [&](int f, const CellRef& a, const CellRef& b, const CellRef& c2, const CellRef& d) {
            Call c{}; c.kind = 2; c.f = (uint8_t)f; c.c[0] = a; c.c[1] = b; c.c[2] = c2; c.c[3] = d; out.push_back(c);
        };
        if (k.kind == 0) {
            const CellRef& c = k.c[0];
            if (o.at(c).kind != C_BRANCH) return true;     // nothing to do: expands to no calls
            for (int i = 0; i < 8; i++) cell(o.child(c, i));
            for (int f = 0; f < 3; f++) {
                int t, u, v; frame(f, &t, &u, &v);
                for (int q : {0, u, v, u | v}) face(f, o.child(c, q), o.child(c, q | t));
            }
            for (int i = 0; i < 2; i++) {
                const int x = i ? AX : 0, y = i ? AY : 0, z = i ? AZ : 0;
                edge(0, o.child(c, x), o.child(c, x | AY), o.child(c, x | AY | AZ), o.child(c, x | AZ));
                edge(1, o.child(c, y), o.child(c, y | AZ), o.child(c, y | AX | AZ), o.child(c, y | AX));
                edge(2, o.child(c, z), o.child(c, z | AX), o.child(c, z | AX | AY), o.child(c, z | AY));
            }
            return true;
        }
        if (k.kind == 1) {
            const CellRef &lo = k.c[0], &hi = k.c[1];
            if (o.is_leaf(lo) && o.is_leaf(hi)) return true;
            const int f = k.f;
            int t, u, v; frame(f, &t, &u, &v);
            face(f, o.child(lo, t), o.child(hi, 0));
            face(f, o.child(lo, t | u), o.child(hi, u));
            face(f, o.child(lo, t | v), o.child(hi, v));
            face(f, o.child(lo, t | u | v), o.child(hi, u | v));
            for (int i = 0; i < 2; i++) {
                const int ui = i ? u : 0, vi = i ? v : 0;
                edge((f + 1) % 3, o.child(lo, ui | t), o.child(lo, ui | v | t), o.child(hi, ui | v), o.child(hi, ui));
                edge((f + 2) % 3, o.child(lo, vi | t), o.child(hi, vi), o.child(hi, vi | u), o.child(lo, vi | u | t));
            }
            return true;
        }
        bool all_leaf = true;
        for (int i = 0; i < 4; i++) all_leaf &= o.is_leaf(k.c[i]);
        if (all_leaf) return false;
        int t, u, v; frame(k.f, &t, &u, &v);
        for (int i = 0; i < 2; i++) {
            const int ti = i ? t : 0;
            edge(k.f, o.child(k.c[0], ti | u | v), o.child(k.c[1], ti | v), o.child(k.c[2], ti), o.child(k.c[3], ti | u));
        }
        return true;
    }
    // Walker::edge on four leaves, recording instead of numbering
    void emit(const Call& k, std::vector<Rec>& out) const {
        const CellRef* cs = k.c;
        Cell leafs[4];
        for (int i = 0; i < 4; i++) { leafs[i] = o.at(cs[i]); if (leafs[i].kind != C_LEAF) return; }
        int deepest = 0;
        for (int i = 0; i < 4; i++) if (cs[i].depth >= cs[deepest].depth) deepest = i;
        int t, u, v; frame(k.f, &t, &u, &v);
        const int ti = axis_index(t);
        const int edges[4] = {ti * 4 + 3, ti * 4 + 2, ti * 4 + 0, ti * 4 + 1};
        int s0, e0;
        edge_corners(edges[deepest], &s0, &e0);
        const bool st = !((leafs[deepest].mask >> s0) & 1), en = !((leafs[deepest].mask >> e0) & 1);
        if (st == en) return;
        const Tables& T = tables();
        int vv[4][2];
        for (int i = 0; i < 4; i++) {
            vv[i][0] = vv[i][1] = -1;
            if (cs[i].depth == cs[deepest].depth) { vv[i][0] = T.e2v[leafs[i].mask][edges[i]][0]; vv[i][1] = T.e2v[leafs[i].mask][edges[i]][1]; }
            else for (int j = 0; j < 12; j++) if (T.e2v[leafs[i].mask][j][0] >= 0) { vv[i][0] = T.e2v[leafs[i].mask][j][0]; vv[i][1] = T.e2v[leafs[i].mask][j][1]; break; }
            if (vv[i][0] < 0) return;
        }
        Rec r;
        r.iv = leafs[deepest].index + (uint64_t)vv[deepest][1];
        for (int i = 0; i < 4; i++) r.vs[i] = leafs[i].index + (uint64_t)vv[i][0];
        r.winding = st ? 3 : 1;
        r.push = 0;
        for (int j = 0; j < 4; j++) {
            const CellRef &p = cs[j], &q = cs[(j + r.winding) % 4];
            if (p.ci != q.ci || p.cj != q.cj) r.push |= (uint8_t)(1 << j);
        }
        out.push_back(r);
    }
    void walk(const Call& k, std::vector<Rec>& out) const {     // depth first from one call
        CallBuf sub;
        if (!expand(k, sub)) { emit(k, out); return; }
        for (int i = 0; i < sub.n; i++) walk(sub.c[i], out);
    }
    void run() {
        tables();
        const bool times = getenv("FHIP_MESH_TIMES") != nullptr;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t0 = now();
        std::vector<Call> calls(1);
        calls[0] = Call{};
        const size_t want = (size_t)mesh_threads() * 64;
        for (int round = 0; round < 8 && calls.size() < want && mesh_threads() > 1; round++) {
            std::vector<Call> next;
            next.reserve(calls.size() * 8);
            for (const Call& k : calls) {
                CallBuf sub;
                if (expand(k, sub)) next.insert(next.end(), sub.c, sub.c + sub.n);
                else next.push_back(k);
            }
            if (next.size() == calls.size()) break;
            calls.swap(next);
        }
        const double t1 = now();
        std::vector<std::vector<Rec>> recs(calls.size());
        parallel_for(calls.size(), [&](size_t i) { walk(calls[i], recs[i]); });
        const double t2 = now();
        size_t nrec = 0;
        std::vector<size_t> rec_base(recs.size() + 1, 0);
        for (size_t c = 0; c < recs.size(); c++) { rec_base[c] = nrec; nrec += recs[c].size(); }
        rec_base[recs.size()] = nrec;
        const size_t nv = std::max<size_t>(o.n_verts(), 1);
        uint32_t* first = nullptr;
        if (mesh_threads() > 1 && nrec * 5 < 0x7FFFFFF0ull) {
            if (scratch) {
                if (*scratch_cap < nv) { free(*scratch); *scratch = (uint32_t*)malloc((nv + nv / 8) * sizeof(uint32_t)); *scratch_cap = *scratch ? nv + nv / 8 : 0; }
                first = *scratch;
            } else
                first = (uint32_t*)malloc(nv * sizeof(uint32_t));
        }
        if (first) {
            // MeshBuilder numbers the vertices by first use (builder.rs), records in call order, within a record iv, vs[0..3].
            // In parallel, with the same result: reference number p = 5 * record + slot; first[v] = the smallest p that names v
            // (atomic min, chunks in parallel); the references with first[v] == p are the first uses, counted per chunk, and a
            // prefix sum over the chunks gives every chunk the number of its first new vertex and of its first triangle.
            // (one array of the octree's vertex count for both: once a chunk has numbered its new vertices it overwrites their
            // entries with TAG | number - a reference number never has the top bit - which is what the triangles then read)
            constexpr uint32_t TAG = 0x80000000u;
            {
                const size_t CH = 1u << 20, nch = (nv + CH - 1) / CH;
                parallel_for(nch, [&](size_t i) { memset(first + i * CH, 0xFF, std::min(CH, nv - i * CH) * sizeof(uint32_t)); });
            }
            auto ref = [](const Rec& r, int k) { return k == 0 ? r.iv : r.vs[k - 1]; };
            const double n0 = now();
            parallel_for(recs.size(), [&](size_t c) {
                uint32_t p = (uint32_t)(rec_base[c] * 5);
                for (const Rec& r : recs[c])
                    for (int k = 0; k < 5; k++, p++) {
                        uint32_t* f = &first[ref(r, k)];
                        uint32_t cur = __atomic_load_n(f, __ATOMIC_RELAXED);
                        while (p < cur && !__atomic_compare_exchange_n(f, &cur, p, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
                    }
            });
            const double n1 = now();
            std::vector<size_t> vbase(recs.size() + 1, 0), tbase(recs.size() + 1, 0);
            parallel_for(recs.size(), [&](size_t c) {
                uint32_t p = (uint32_t)(rec_base[c] * 5);
                size_t nvert = 0, ntri = 0;
                for (const Rec& r : recs[c]) {
                    for (int k = 0; k < 5; k++, p++) nvert += first[ref(r, k)] == p;
                    ntri += (size_t)__builtin_popcount(r.push & 15u);
                }
                vbase[c + 1] = nvert; tbase[c + 1] = ntri;
            });
            for (size_t c = 0; c < recs.size(); c++) { vbase[c + 1] += vbase[c]; tbase[c + 1] += tbase[c]; }
            const double n2 = now();
            vertices.resize(vbase[recs.size()]);
            std::vector<uint32_t, default_init_allocator<uint32_t>> gidx(gather ? vertices.size() : 0);
            triangles.resize(tbase[recs.size()]);
            const double n3 = now();
            parallel_for(recs.size(), [&](size_t c) {
                uint32_t p = (uint32_t)(rec_base[c] * 5);
                size_t id = vbase[c];
                for (const Rec& r : recs[c])
                    for (int k = 0; k < 5; k++, p++) {
                        const uint64_t v = ref(r, k);
                        // (another chunk may be looking at the same entry: it sees the first use's number or the tagged value, neither is its own p)
                        if (__atomic_load_n(&first[v], __ATOMIC_RELAXED) == p) {
                            if (gather) gidx[id] = (uint32_t)v; else vertices[id] = o.vert(v);
                            __atomic_store_n(&first[v], TAG | (uint32_t)id, __ATOMIC_RELAXED);
                            id++;
                        }
                    }
            });
            parallel_for(recs.size(), [&](size_t c) {
                size_t t = tbase[c];
                for (const Rec& r : recs[c]) {
                    const uint64_t iv = first[r.iv] & ~TAG;
                    uint64_t vs[4];
                    for (int i = 0; i < 4; i++) vs[i] = first[r.vs[i]] & ~TAG;
                    for (int j = 0; j < 4; j++)
                        if (r.push & (1 << j)) triangles[t++] = {vs[j], vs[(j + r.winding) % 4], iv};
                }
            });
            if (gather && !gather(gidx.data(), gidx.size(), vertices.data())) gather_failed = true;
            const double n4 = now();
            if (!scratch) free(first);
            if (times) fprintf(stderr, "fhip dual walk numbering: %zu octree vertices; clear %.4f s, first uses %.4f s, counts %.4f s, room %.4f s, vertices + triangles %.4f s, free %.4f s\n",
                               nv, n0 - t2, n1 - n0, n2 - n1, n3 - n2, n4 - n3, now() - n4);
        } else {
            // octree vertex -> mesh vertex + 1 (0: not seen yet); calloc: only the pages that are touched cost anything
            uint32_t* map = (uint32_t*)calloc(nv, sizeof(uint32_t));
            triangles.reserve(nrec * 4);
            vertices.reserve(nrec * 2);
            std::vector<uint32_t> gidx;
            auto vertex = [&](uint64_t v) {
                if (map[v] == 0) {
                    if (gather) { gidx.push_back((uint32_t)v); map[v] = (uint32_t)gidx.size(); }
                    else { vertices.push_back(o.vert(v)); map[v] = (uint32_t)vertices.size(); }
                }
                return (uint64_t)(map[v] - 1);
            };
            for (auto& rs : recs)
                for (const Rec& r : rs) {
                    const uint64_t iv = vertex(r.iv);
                    uint64_t vs[4];
                    for (int i = 0; i < 4; i++) vs[i] = vertex(r.vs[i]);
                    for (int j = 0; j < 4; j++)
                        if (r.push & (1 << j)) triangles.push_back({vs[j], vs[(j + r.winding) % 4], iv});
                }
            free(map);
            if (gather) { vertices.resize(gidx.size()); if (!gather(gidx.data(), gidx.size(), vertices.data())) gather_failed = true; }
        }
        if (times) fprintf(stderr, "fhip dual walk: unroll %.4f s (%zu calls), sub-walks %.4f s (%zu records), numbering %.4f s\n", t1 - t0, calls.size(), t2 - t1, nrec, now() - t2);
    }
};

}  // namespace fhmesh
