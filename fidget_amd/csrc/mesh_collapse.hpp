// fidget-hip: the bottom-up half of fidget_mesh::Octree::build - check_done / collapsible (fidget-mesh/src/octree.rs:256-470) and the
// merged Hermite data of a collapsed cell, LeafHermiteData (octree.rs:866-1035) - as per-cell functions.  ONE definition, compiled
// for the device (mesh.hip: the assembly kernels of fhip_mesh_build) and for the host (host_mesh.hpp: the assembly fhip_mesh_merge
// runs on the parts' records), so that a collapsed cell comes out of the same arithmetic on either side.
//
// Also here: the records the device side of the mesh path writes (a level's cells, the leaf samples, the Manifold-DC table).
#pragma once
#include <math.h>
#include <stdint.h>

#include "mesh_qef.hpp"

struct FhMeshCell {
    float b[6];        // x.lo x.hi y.lo y.hi z.lo z.hi
    uint64_t path;     // 3 bits per level below the root (corner index), leading 1
};
struct FhMeshLeaf {
    float b[6];
    uint64_t path;
    uint32_t mask, n_edges, n_verts, pad;
    uint16_t inter[12][3];
    uint16_t pad2[4];
    float pos[12][3];
    float grad[12][4];   // dx dy dz v
    float vert[4][3];
    float qef_err[4];
};
// CELL_TO_VERT_TO_EDGES (fidget-mesh/build.rs), flattened: per mask the edges in vertex order as (start, end), edges per vertex
struct FhMdcTable {
    uint8_t n_edges[256], n_verts[256];
    uint8_t per_vert[256][4];
    uint8_t edge[256][12][2];
};

namespace fhmesh {

enum { AX = 1, AY = 2, AZ = 4 };
FHQ_HD static inline int axis_next(int a) { return (a << 1) > AZ ? AX : (a << 1); }   // types.rs Axis::next
FHQ_HD static inline int axis_index(int a) { return a == 1 ? 0 : (a == 2 ? 1 : 2); }
FHQ_HD static inline int to_undirected(int start, int end) {     // types.rs DirectedEdge::to_undirected
    const int t = start ^ end, u = axis_next(t), v = axis_next(u);
    return axis_index(t) * 4 + ((start & v) ? 2 : 0) + ((start & u) ? 1 : 0);
}

enum CellKind : uint8_t { C_INVALID = 0, C_EMPTY, C_FULL, C_BRANCH, C_LEAF };
struct Cell {
    uint8_t kind = C_INVALID, mask = 0;
    uint32_t index = 0;
    FHQ_HD bool corner(int c) const { return kind == C_LEAF ? ((mask >> c) & 1) : kind == C_FULL; }
};
static_assert(sizeof(Cell) == 8, "a block of the octree is 8 cells of 8 bytes");
struct V3 { float x, y, z; };

constexpr float QEF_ERR_EMPTY = -1.0f, QEF_ERR_INVALID = -2.0f;
struct LeafIntersection { float pos[4] = {0, 0, 0, 0}, grad[4] = {0, 0, 0, 0}; };
FHQ_HD static inline fhq::Qef qef_zero() { fhq::Qef q; q.init(); return q; }
FHQ_HD static inline fhq::Qef qef_of(const LeafIntersection& i) { fhq::Qef q = qef_zero(); if (i.pos[3] != 0.0f) q.add(i.pos, i.grad); return q; }

// LeafHermiteData (octree.rs:866-1035)
struct Hermite {
    LeafIntersection inter[12];
    fhq::Qef face[6], center;
    float qef_err = QEF_ERR_EMPTY;
    FHQ_HD Hermite() { for (int i = 0; i < 6; i++) face[i].init(); center.init(); }
    // merge (octree.rs:895-1003) of the eight children `s` describes: s.err(i), s.inter(i, edge), s.face(i, f), s.center(i)
    template <class Src>
    FHQ_HD static bool merge_from(const Src& s, Hermite* out) {
        *out = Hermite();
        for (int i = 0; i < 8; i++) if (s.err(i) == QEF_ERR_INVALID) return false;
        for (int ti = 0; ti < 3; ti++) {
            const int t = 1 << ti, u = axis_next(t), v = axis_next(u);
            for (int edge = 0; edge < 4; edge++) {
                int start = 0;
                if (edge & 1) start |= u;
                if (edge & 2) start |= v;
                const int end = start | t, e = axis_index(t) * 4 + edge;
                const LeafIntersection a = s.inter(start, e), b = s.inter(end, e);
                if (a.pos[3] > 0.0f && !(b.pos[3] > 0.0f)) out->inter[e] = a;
                else if (!(a.pos[3] > 0.0f) && b.pos[3] > 0.0f) out->inter[e] = b;
            }
        }
        for (int ti = 0; ti < 3; ti++) {
            const int t = 1 << ti, u = axis_next(t), v = axis_next(t);   // (octree.rs:946-947: both are t.next())
            for (int fc = 0; fc < 2; fc++) {
                const int a = fc == 1 ? t : 0, b = a | u, c = a | v, d = a | u | v, f = axis_index(t) * 2 + fc;
                const int four[4] = {a, b, c, d};
                for (int q = 0; q < 4; q++) out->face[f].merge(s.face(four[q], f));
                const int ev = axis_index(v) * 4 + fc * 2 + 1;
                out->face[f].merge(qef_of(s.inter(a, ev)));
                out->face[f].merge(qef_of(s.inter(b, ev)));
                out->face[f].merge(qef_of(s.inter(a, ev)));
                out->face[f].merge(qef_of(s.inter(c, ev)));
            }
        }
        for (int ti = 0; ti < 3; ti++) {
            const int t = 1 << ti, u = axis_next(t), v = axis_next(t);
            const int a = 0, b = a | u, c = a | v, d = a | u | v;
            const int four[4] = {a, b, c, d};
            for (int q = 0; q < 4; q++) out->center.merge(s.face(four[q], axis_index(t) * 2 + 1));
            out->center.merge(qef_of(s.inter(a, axis_index(u) * 4 + 3)));
            out->center.merge(qef_of(s.inter(b, axis_index(u) * 4 + 3)));
        }
        for (int i = 0; i < 8; i++) out->center.merge(s.center(i));
        out->qef_err = INFINITY;
        for (int i = 0; i < 8; i++) if (s.err(i) >= 0.0f) out->qef_err = fminf(out->qef_err, s.err(i));
        return true;
    }
    struct ArraySrc {
        const Hermite* h;
        FHQ_HD float err(int i) const { return h[i].qef_err; }
        FHQ_HD const LeafIntersection& inter(int i, int e) const { return h[i].inter[e]; }
        FHQ_HD const fhq::Qef& face(int i, int f) const { return h[i].face[f]; }
        FHQ_HD const fhq::Qef& center(int i) const { return h[i].center; }
    };
    static bool merge(const Hermite* leafs, Hermite* out) { return merge_from(ArraySrc{leafs}, out); }
    FHQ_HD void solve(float* pos, float* err) const {
        fhq::Qef q = center;
        for (int i = 0; i < 12; i++) q.merge(qef_of(inter[i]));
        for (int f = 0; f < 6; f++) q.merge(face[f]);
        q.solve(pos, err);
    }
};

// collapsible (octree.rs:389-470) on the eight children's kinds and corner masks; n_verts[mask] = cell vertices of a corner mask
FHQ_HD static inline bool collapsible_children(const uint8_t* kind, const uint8_t* cmask, const uint8_t* n_verts, uint8_t* out_mask) {
    int mask = 0;
    for (int i = 0; i < 8; i++) {
        int b;
        if (kind[i] == C_LEAF) { if (n_verts[cmask[i]] > 1) return false; b = (cmask[i] >> i) & 1; }
        else if (kind[i] == C_EMPTY) b = 0;
        else if (kind[i] == C_FULL) b = 1;
        else return false;
        mask |= b << i;
    }
    auto corner = [&](int i, int c) -> bool { return kind[i] == C_LEAF ? (((cmask[i] >> c) & 1) != 0) : kind[i] == C_FULL; };
    auto bit = [&](int q) -> bool { return ((mask >> q) & 1) != 0; };
    for (int fi = 0; fi < 3; fi++) {
        const int t = 1 << fi, u = axis_next(t), v = axis_next(u);
        for (int i = 0; i < 4; i++) {
            const int a = ((i & 1) ? u : 0) | ((i & 2) ? v : 0), b = a | t;
            const bool center = corner(a, b);
            if (bit(a) != center && bit(b) != center) return false;
        }
        for (int i = 0; i < 2; i++) {
            const int a = ((i & 1) == 0) ? t : 0, b = a | u, c = a | v, d = a | u | v;
            const bool center = corner(a, d);
            if (bit(a) != center && bit(b) != center && bit(c) != center && bit(d) != center) return false;
        }
        const bool center = corner(0, t | u | v);
        bool all = true;
        for (int q = 0; q < 8; q++) all &= (bit(q) != center);
        if (all) return false;
    }
    if (n_verts[mask] == 1) { *out_mask = (uint8_t)mask; return true; }
    return false;
}

// octree.rs:58-65: a vertex back to model space (nalgebra transform_point)
FHQ_HD static inline V3 vertex_to_model(const float* mat, V3 p) {
    const float x = p.x, y = p.y, z = p.z;
    const float n = ((mat[12] * x + mat[13] * y) + mat[14] * z) + mat[15];
    float a = ((mat[0] * x + mat[1] * y) + mat[2] * z) + mat[3];
    float b = ((mat[4] * x + mat[5] * y) + mat[6] * z) + mat[7];
    float c = ((mat[8] * x + mat[9] * y) + mat[10] * z) + mat[11];
    if (n != 0.0f) { a = a / n; b = b / n; c = c / n; }
    return V3{a, b, c};
}

// ---- the assembly level by level (the device's form of Octree::recurse unwinding, octree.rs:556-583) --------------------------------
// Level d holds the cells the recursion evaluates at depth d: class (1 empty 2 full 3 ambiguous) and, for an ambiguous cell, its
// slot among the level's ambiguous cells; the children of slot s are cells 8 s .. 8 s + 7 of level d + 1; the ambiguous cells of the
// last level are the leaf records.  Bottom-up every ambiguous cell gets a result (what check_done returns for it, and how many
// vertices / blocks of eight cells its subtree leaves in the octree's arrays); top-down every cell learns where its vertices
// and its block go - the positions the single-threaded recursion gives them (vertices in the order the leaves are reached, a
// collapsed cell's after its children's; blocks in pre-order of the branches that stay).
struct OctRes {
    uint8_t kind, mask;
    uint16_t own;          // vertices this cell itself appends (a collapsed cell: its vertex + its edges' intersections)
    uint32_t tv, tb;       // vertices / blocks its subtree leaves in the arrays
    uint32_t herm;         // a collapsed cell: its entry in the level's pool of merged Hermite data
};
struct OctPlace { uint32_t vo, bo; };
struct OctCollapsed { Hermite h; float pos[3]; uint32_t pad; };
struct OctLevel {
    const uint8_t* cls = nullptr;
    const uint32_t* slot = nullptr;
    const FhMeshCell* amb = nullptr;       // the level's ambiguous cells by slot (bounds)
    OctRes* res = nullptr;                 // ... their results
    OctPlace* place = nullptr;             // ... and places
    uint32_t* cand = nullptr;              // slots of the cells collapsible() lets through
    OctCollapsed* pool = nullptr;          // one entry per candidate
    uint32_t n_amb = 0;
};
struct OctLeaves {
    const FhMeshLeaf* rec = nullptr;       // non-null: the level below is the leaf level, its ambiguous cells are these records
    uint32_t* vo = nullptr;                // where a record's vertices go
    const FhMdcTable* T = nullptr;
};

struct OctChild { uint8_t kind, mask; uint32_t tv, tb, own, slot; };
FHQ_HD static inline OctChild oct_child(const OctLevel& C, const OctLeaves& L, uint32_t j) {
    OctChild c{C_INVALID, 0, 0, 0, 0, 0xFFFFFFFFu};
    const uint8_t cl = C.cls[j];
    if (cl == 2) c.kind = C_FULL;
    else if (cl == 1) c.kind = C_EMPTY;
    else if (cl == 3) {
        c.slot = C.slot[j];
        if (L.rec) {
            const FhMeshLeaf& lf = L.rec[c.slot];
            if (lf.mask == 0) c.kind = C_EMPTY;
            else if (lf.mask == 255) c.kind = C_FULL;
            else { c.kind = C_LEAF; c.mask = (uint8_t)lf.mask; c.tv = c.own = lf.n_verts + lf.n_edges; }
        } else {
            const OctRes r = C.res[c.slot];
            c.kind = r.kind; c.mask = r.mask; c.tv = r.tv; c.tb = r.tb; c.own = r.own;
        }
    }
    return c;
}

// what check_done needs of a child for the merge: a leaf record's intersections (by the record's slot of each undirected edge, as
// leaf() fills LeafHermiteData, octree.rs:805-848: a NaN gradient invalidates the data and ends its vertex' loop without consuming
// the edge), a collapsed cell's merged data, or nothing (empty / full: the default data)
struct OctChildData {
    const FhMeshLeaf* lf = nullptr;
    const Hermite* h = nullptr;
    float err = QEF_ERR_EMPTY;
    int8_t rec_slot[12];
};
FHQ_HD static inline void oct_child_data(const OctLevel& C, const OctLeaves& L, const OctChild& c, OctChildData* d) {
    d->lf = nullptr; d->h = nullptr; d->err = QEF_ERR_EMPTY;
    for (int e = 0; e < 12; e++) d->rec_slot[e] = -1;
    if (c.kind != C_LEAF) return;
    if (!L.rec) { d->h = &C.pool[C.res[c.slot].herm].h; d->err = d->h->qef_err; return; }
    const FhMeshLeaf* lf = &L.rec[c.slot];
    d->lf = lf;
    const FhMdcTable& T = *L.T;
    uint32_t ii = 0, n = 0;
    for (uint32_t vi = 0; vi < T.n_verts[c.mask]; vi++) {
        bool forced = false;
        for (uint32_t k = 0; k < T.per_vert[c.mask][vi]; k++) {
            const uint32_t kk = ii < 11 ? ii : 11;
            const float* g = lf->grad[kk];
            if (g[0] != g[0] || g[1] != g[1] || g[2] != g[2] || g[3] != g[3]) { forced = true; d->err = QEF_ERR_INVALID; break; }
            d->rec_slot[to_undirected(T.edge[c.mask][n + k][0], T.edge[c.mask][n + k][1])] = (int8_t)kk;
            ii++;
        }
        if (!forced) d->err = lf->qef_err[vi];
        n += T.per_vert[c.mask][vi];
    }
}
struct OctChildSrc {
    const OctChildData* d;
    FHQ_HD float err(int i) const { return d[i].err; }
    FHQ_HD LeafIntersection inter(int i, int e) const {
        if (d[i].h) return d[i].h->inter[e];
        LeafIntersection li;
        if (d[i].lf && d[i].rec_slot[e] >= 0) {
            const int k = d[i].rec_slot[e];
            li.pos[0] = d[i].lf->pos[k][0]; li.pos[1] = d[i].lf->pos[k][1]; li.pos[2] = d[i].lf->pos[k][2]; li.pos[3] = 1.0f;
            for (int q = 0; q < 4; q++) li.grad[q] = d[i].lf->grad[k][q];
        }
        return li;
    }
    FHQ_HD fhq::Qef face(int i, int f) const { return d[i].h ? d[i].h->face[f] : qef_zero(); }
    FHQ_HD fhq::Qef center(int i) const { return d[i].h ? d[i].h->center : qef_zero(); }
};

#if defined(__HIP_DEVICE_COMPILE__)
#define FH_OCT_COUNT(p) atomicAdd((p), 1u)
#else
#define FH_OCT_COUNT(p) ((*(p))++)
#endif

// pass 1, one ambiguous cell (slot s) of level d: everything of check_done (octree.rs:256-340) but the merge; cells collapsible()
// lets through are listed for pass 2
FHQ_HD static inline void oct_kind(const OctLevel& D, const OctLevel& C, const OctLeaves& L, const FhMdcTable* T, uint32_t s, uint32_t* n_cand) {
    uint8_t kind[8], mask[8];
    uint64_t tv = 0;       // (32-bit vertex indices, as the host's octree has them: the count saturates, oct_assemble refuses)
    uint32_t tb = 0;
    int full = 0, empty = 0;
    bool branch = false;
    for (int c = 0; c < 8; c++) {
        const OctChild ch = oct_child(C, L, s * 8 + c);
        kind[c] = ch.kind; mask[c] = ch.mask;
        tv += ch.tv; tb += ch.tb;
        if (ch.kind == C_FULL) full++;
        else if (ch.kind == C_EMPTY) empty++;
        else if (ch.kind != C_LEAF) branch = true;       // (a branch; or a cell of another part: nothing collapses over it)
    }
    OctRes r{C_BRANCH, 0, 0, tv > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)tv, tb + 1, 0xFFFFFFFFu};
    if (!branch) {
        if (full == 8) { r.kind = C_FULL; r.tb = 0; }
        else if (empty == 8) { r.kind = C_EMPTY; r.tb = 0; }
        else {
            uint8_t m;
            if (collapsible_children(kind, mask, T->n_verts, &m)) { r.mask = m; D.cand[FH_OCT_COUNT(n_cand)] = s; }
        }
    }
    D.res[s] = r;
}
// pass 2, candidate k of level d: merge, solve, the error and bounds tests (octree.rs:296-330)
FHQ_HD static inline void oct_collapse(const OctLevel& D, const OctLevel& C, const OctLeaves& L, const FhMdcTable* T, uint32_t k) {
    const uint32_t s = D.cand[k];
    OctChildData cd[8];
    for (int c = 0; c < 8; c++) oct_child_data(C, L, oct_child(C, L, s * 8 + c), &cd[c]);
    OctCollapsed* out = &D.pool[k];
    if (!Hermite::merge_from(OctChildSrc{cd}, &out->h)) return;
    float pos[3], err = 0;
    out->h.solve(pos, &err);
    const float* b = D.amb[s].b;
    bool inside = true;
    for (int q = 0; q < 3; q++) inside &= pos[q] >= b[2 * q] && pos[q] <= b[2 * q + 1];
    if (err >= out->h.qef_err * 2.0f || !inside) return;
    out->h.qef_err = err;
    for (int q = 0; q < 3; q++) out->pos[q] = pos[q];
    OctRes r = D.res[s];
    r.kind = C_LEAF;
    r.own = (uint16_t)(1 + T->per_vert[r.mask][0]);
    r.tv = r.tv > 0xFFFFFFFFu - r.own ? 0xFFFFFFFFu : r.tv + r.own;
    r.tb = 0; r.herm = k;
    D.res[s] = r;
}
// what the parent's block (or Octree::root) says of a cell whose vertices start at v and whose first block is b
FHQ_HD static inline Cell oct_cell(const OctChild& c, uint32_t v, uint32_t b) {
    Cell x;
    x.kind = c.kind; x.mask = c.kind == C_LEAF ? c.mask : 0;
    x.index = c.kind == C_BRANCH ? b : (c.kind == C_LEAF ? v + c.tv - c.own : 0);
    return x;
}
// top-down, slot s of level d, placed at D.place[s]: its children's places, its block if it stays a branch, its own vertices if it collapsed
FHQ_HD static inline void oct_place(const OctLevel& D, const OctLevel& C, const OctLeaves& L, const FhMdcTable* T, uint32_t s, Cell* cells, V3* verts,
                                    const float* mat /* or null */) {
    const OctRes r = D.res[s];
    const OctPlace p = D.place[s];
    uint32_t v = p.vo, b = p.bo + 1;
    for (int c = 0; c < 8; c++) {
        const OctChild ch = oct_child(C, L, s * 8 + c);
        if (ch.slot != 0xFFFFFFFFu) {
            if (L.rec) L.vo[ch.slot] = v;
            else C.place[ch.slot] = OctPlace{v, b};
        }
        if (r.kind == C_BRANCH) cells[(size_t)p.bo * 8 + c] = oct_cell(ch, v, b);
        v += ch.tv; b += ch.tb;
    }
    if (r.kind == C_LEAF) {
        const OctCollapsed& oc = D.pool[r.herm];
        V3 q{oc.pos[0], oc.pos[1], oc.pos[2]};
        verts[v++] = mat ? vertex_to_model(mat, q) : q;
        for (uint32_t k = 0; k < T->per_vert[r.mask][0]; k++) {
            const LeafIntersection& li = oc.h.inter[to_undirected(T->edge[r.mask][k][0], T->edge[r.mask][k][1])];
            q = V3{li.pos[0], li.pos[1], li.pos[2]};
            verts[v++] = mat ? vertex_to_model(mat, q) : q;
        }
    }
}
// ... and leaf record i: the cell's vertices, then its edges' intersections (octree.rs:850-861)
FHQ_HD static inline void oct_leaf_verts(const OctLeaves& L, uint32_t i, V3* verts, const float* mat) {
    const FhMeshLeaf& lf = L.rec[i];
    if (lf.mask == 0 || lf.mask == 255) return;
    uint32_t v = L.vo[i];
    for (uint32_t k = 0; k < lf.n_verts; k++) { const V3 q{lf.vert[k][0], lf.vert[k][1], lf.vert[k][2]}; verts[v++] = mat ? vertex_to_model(mat, q) : q; }
    for (uint32_t k = 0; k < lf.n_edges; k++) { const V3 q{lf.pos[k][0], lf.pos[k][1], lf.pos[k][2]}; verts[v++] = mat ? vertex_to_model(mat, q) : q; }
}

// The passes in order.  X says where the arrays live and how a pass over n items runs (the device: hipMalloc and one kernel launch per
// pass, capi.hip; a plain loop in the tests' host build of these functions): alloc(bytes) -> pointer or null, zero(p, bytes),
// read(dst, src, bytes) (a synchronising copy to the host), kind / collapse / place / leaf_verts(.., n).  lv[0 .. n_levels - 1]: the levels
// the recursion evaluated, with cls / slot / amb / n_amb filled in (n_amb of the leaf level = n_rec); the leaf level is level `depth`, if
// the recursion got there.  Returns OCT_OK, OCT_NO_MEMORY (X keeps what alloc handed out) or OCT_TOO_MANY_VERTICES (the octree's vertex
// indices are 32 bits wide, here as in the host's assembly).
struct OctOut { Cell root; Cell* cells = nullptr; V3* verts = nullptr; uint32_t n_blocks = 0, n_verts = 0; };
enum { OCT_OK = 0, OCT_NO_MEMORY = 1, OCT_TOO_MANY_VERTICES = 2 };
template <class X>
static inline int oct_assemble(X& x, uint32_t depth, OctLevel* lv, uint32_t n_levels, const FhMeshLeaf* rec, uint32_t n_rec, const FhMdcTable* T, const float* mat,
                                OctOut* out) {
    *out = OctOut();
    uint8_t root_cls = 0;
    x.read(&root_cls, lv[0].cls, 1);
    if (root_cls != 3) { out->root.kind = root_cls == 2 ? C_FULL : C_EMPTY; return OCT_OK; }
    const bool leaf_level = n_levels == depth + 1 && n_rec > 0;
    OctLeaves leaves;
    leaves.T = T;
    if (leaf_level) {
        leaves.rec = rec;
        leaves.vo = (uint32_t*)x.alloc((size_t)n_rec * 4);
        if (!leaves.vo) return OCT_NO_MEMORY;
    }
    const OctLeaves none{nullptr, nullptr, T};
    OctChild top{C_INVALID, 0, 0, 0, 0, 0};
    if (depth == 0) {       // the root is the one leaf cell
        FhMeshLeaf lf;
        x.read(&lf, rec, sizeof(lf));
        x.zero(leaves.vo, 4);
        top.kind = lf.mask == 0 ? C_EMPTY : (lf.mask == 255 ? C_FULL : C_LEAF);
        if (top.kind == C_LEAF) { top.mask = (uint8_t)lf.mask; top.tv = top.own = lf.n_verts + lf.n_edges; }
    } else {
        if (n_levels < 2) return OCT_OK;       // (cannot be: an ambiguous root above the leaf level has children)
        uint32_t* counter = (uint32_t*)x.alloc(4);
        if (!counter) return OCT_NO_MEMORY;
        for (uint32_t d = n_levels - 1; d-- > 0;) {
            OctLevel& D = lv[d];
            D.res = (OctRes*)x.alloc((size_t)D.n_amb * sizeof(OctRes));
            D.cand = (uint32_t*)x.alloc((size_t)D.n_amb * 4);
            if (!D.res || !D.cand) return OCT_NO_MEMORY;
            x.zero(counter, 4);
            const OctLeaves& L = (d + 1 == depth) ? leaves : none;
            x.kind(D, lv[d + 1], L, T, counter, D.n_amb);
            uint32_t nc = 0;
            x.read(&nc, counter, 4);
            if (nc) {
                D.pool = (OctCollapsed*)x.alloc((size_t)nc * sizeof(OctCollapsed));
                if (!D.pool) return OCT_NO_MEMORY;
                x.collapse(D, lv[d + 1], L, T, nc);
            }
        }
        OctRes r;
        x.read(&r, lv[0].res, sizeof(r));
        top.kind = r.kind; top.mask = r.mask; top.tv = r.tv; top.tb = r.tb; top.own = r.own;
    }
    if (top.tv >= 0xFFFFFF00u) return OCT_TOO_MANY_VERTICES;
    out->root = oct_cell(top, 0, 0);
    out->n_verts = top.tv; out->n_blocks = top.tb;
    if (top.tv) { out->verts = (V3*)x.alloc((size_t)top.tv * sizeof(V3)); if (!out->verts) return OCT_NO_MEMORY; }
    if (top.tb) { out->cells = (Cell*)x.alloc((size_t)top.tb * 8 * sizeof(Cell)); if (!out->cells) return OCT_NO_MEMORY; }
    if (depth > 0) {
        lv[0].place = (OctPlace*)x.alloc(sizeof(OctPlace));
        if (!lv[0].place) return OCT_NO_MEMORY;
        x.zero(lv[0].place, sizeof(OctPlace));
        for (uint32_t d = 0; d + 1 < n_levels; d++) {
            OctLevel& C = lv[d + 1];
            const bool to_leaves = d + 1 == depth;
            if (!to_leaves && C.n_amb) { C.place = (OctPlace*)x.alloc((size_t)C.n_amb * sizeof(OctPlace)); if (!C.place) return OCT_NO_MEMORY; }
            x.place(lv[d], C, to_leaves ? leaves : none, T, out->cells, out->verts, mat, lv[d].n_amb);
        }
    }
    if (leaf_level) x.leaf_verts(leaves, out->verts, mat, n_rec);
    return OCT_OK;
}

}  // namespace fhmesh
