// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of fidget-mesh (Manifold Dual Contouring): the connectivity tables of build.rs,
// cell geometry (cell.rs), the octree builder (octree.rs:521-862: recurse, leaf sampling, QEF
// vertices; 256-470: check_done / try_collapse / collapsible; 866-1035: LeafHermiteData) and the
// dual walk (dc.rs, builder.rs).  Each function cites the lines it follows.
//
// PARITY: cell classification, corner masks, edge-search intersections (u16 positions and their
// f32 images) and gradients are exact restatements (integer / op-by-op f32 arithmetic).  The QEF
// solve uses nalgebra's SVD in the reference (qef.rs:67-126); here the 3x3 symmetric A^T A is
// diagonalised by cyclic Jacobi rotations: vertex positions and QEF errors agree with the reference
// only to rounding (the reference's own tests use 1e-3 .. 2/65535 tolerances), and cell collapse
// decisions that compare such errors (try_collapse) can differ in ties.
#pragma once
#include <deque>
#include <array>
#include <cmath>
#include <cstdint>
#include <functional>
#include <vector>

#include "render.hpp"

namespace orc {
namespace mesh {

enum { AX = 1, AY = 2, AZ = 4 };
static inline int axis_next(int a) { return (a << 1) > AZ ? AX : (a << 1); }   // types.rs Axis::next / build.rs next()
static inline int axis_index(int a) { return a == 1 ? 0 : (a == 2 ? 1 : 2); }

// ---- build.rs: tables ------------------------------------------------------------------------------
struct Tables {
    // vert_to_edges[mask] = vertices, each a list of (start = inside corner, end = outside corner)
    std::vector<std::vector<std::pair<uint8_t, uint8_t>>> v2e[256];
    // edge_to_vert[mask][edge] = (vertex offset, intersection offset) or (-1, -1)
    int e2v[256][12][2];
};
static inline const Tables& tables() {
    static Tables* T = nullptr;
    if (T) return *T;
    Tables* t = new Tables();
    for (int i = 0; i < 256; i++) {
        // connected regions of filled / empty corners (build.rs:39-74)
        int filled[8], empty[8];
        bool is_f[8];
        for (int j = 0; j < 8; j++) { is_f[j] = (i >> j) & 1; filled[j] = empty[j] = 1 << j; }
        for (int pass = 0; pass < 2; pass++) {
            int* r = pass == 0 ? filled : empty;
            bool changed = true;
            while (changed) {
                changed = false;
                int next[8];
                for (int j = 0; j < 8; j++) next[j] = r[j];
                for (int f = 0; f < 8; f++) {
                    if (is_f[f] != (pass == 0)) continue;
                    for (int axis : {AX, AY, AZ}) {
                        const int g = f ^ axis;
                        if (is_f[g] != (pass == 0)) continue;
                        const int v = next[f] | next[g];
                        changed |= (next[f] != v) | (next[g] != v);
                        next[f] = v; next[g] = v;
                    }
                }
                for (int j = 0; j < 8; j++) r[j] = next[j];
            }
        }
        // distinct region masks in ascending order (BTreeSet), filled first, then empty (build.rs:80-97)
        std::vector<int> fr, er;
        for (int j = 0; j < 8; j++) {
            if (is_f[j]) fr.push_back(filled[j]); else er.push_back(empty[j]);
        }
        auto uniq = [](std::vector<int>& v) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); };
        uniq(fr); uniq(er);
        int regions[8];
        for (int j = 0; j < 8; j++) regions[j] = 255;
        int ri = 0;
        for (auto* rs : {&fr, &er})
            for (int r : *rs) {
                for (int j = 0; j < 8; j++) if (r & (1 << j)) regions[j] = ri;
                ri++;
            }
        // transition edges grouped by the region of their inside corner (build.rs:103-133): BTreeMap order = region number
        std::vector<std::pair<int, std::vector<std::pair<uint8_t, uint8_t>>>> verts;
        auto entry = [&](int region) -> std::vector<std::pair<uint8_t, uint8_t>>& {
            for (auto& kv : verts) if (kv.first == region) return kv.second;
            verts.push_back({region, {}});
            return verts.back().second;
        };
        for (int rev = 0; rev < 2; rev++)
            for (int tt : {AX, AY, AZ}) {
                const int u = axis_next(tt), v = axis_next(u);
                for (int b = 0; b < 2; b++)
                    for (int a = 0; a < 2; a++) {
                        int start = (a * u) | (b * v), end = start | tt;
                        if (rev) std::swap(start, end);
                        if (((i >> start) & 1) && !((i >> end) & 1)) entry(regions[start]).push_back({(uint8_t)start, (uint8_t)end});
                    }
            }
        std::sort(verts.begin(), verts.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        for (int e = 0; e < 12; e++) t->e2v[i][e][0] = t->e2v[i][e][1] = -1;
        const int vert_count = (int)verts.size();
        int intersection_count = 0;
        for (int vi = 0; vi < vert_count; vi++) {
            t->v2e[i].push_back(verts[vi].second);
            for (auto& se : verts[vi].second) {
                const int start = se.first, end = se.second;
                const int tt = start ^ end, u = axis_next(tt), v = axis_next(u);
                const int edge = axis_index(tt) * 4 + ((start & u) ? 1 : 0) + ((start & v) ? 2 : 0);
                t->e2v[i][edge][0] = vi;
                t->e2v[i][edge][1] = vert_count + intersection_count;
                intersection_count++;
            }
        }
    }
    T = t;
    return *T;
}
// types.rs DirectedEdge::to_undirected
static inline int to_undirected(int start, int end) {
    const int t = start ^ end, u = axis_next(t), v = axis_next(u);
    return axis_index(t) * 4 + ((start & v) ? 2 : 0) + ((start & u) ? 1 : 0);
}
// types.rs Edge::corners
static inline void edge_corners(int e, int* start, int* end) {
    static const int FR[3][3] = {{AX, AY, AZ}, {AY, AZ, AX}, {AZ, AX, AY}};
    const int t = FR[e / 4][0], u = ((e % 4) % 2 != 0) ? FR[e / 4][1] : 0, v = ((e % 4) / 2 != 0) ? FR[e / 4][2] : 0;
    *start = u | v; *end = t | u | v;
}

// ---- cell.rs ---------------------------------------------------------------------------------------
struct Bounds {
    Interval b[3];
    Bounds() { for (auto& i : b) i = Interval(-1.0f, 1.0f); }
    Bounds child(int corner) const {      // cell.rs:184-194
        Bounds o;
        for (int i = 0; i < 3; i++) {
            const float mid = (b[i].lo + b[i].hi) / 2.0f;   // Interval::midpoint (interval.rs:421-424)
            o.b[i] = (corner & (1 << i)) ? Interval(mid, b[i].hi) : Interval(b[i].lo, mid);
        }
        return o;
    }
    void corner(int c, float* out) const { for (int i = 0; i < 3; i++) out[i] = (c & (1 << i)) ? b[i].hi : b[i].lo; }
    void pos(const uint16_t* p, float* out) const {   // cell.rs:208-217, Interval::lerp (interval.rs:454-456)
        for (int i = 0; i < 3; i++) {
            const float f = (float)p[i] / 65535.0f;
            out[i] = b[i].lo * (1.0f - f) + b[i].hi * f;
        }
    }
    bool contains(const float* p) const {
        for (int i = 0; i < 3; i++) if (!(p[i] >= b[i].lo && p[i] <= b[i].hi)) return false;
        return true;
    }
};
enum CellKind : uint8_t { C_INVALID = 0, C_EMPTY, C_FULL, C_BRANCH, C_LEAF };
struct Cell {
    uint8_t kind = C_INVALID, mask = 0;
    uint32_t index = 0;     // Branch: index in cells; Leaf: first vertex
    bool corner(int c) const { return kind == C_LEAF ? ((mask >> c) & 1) : kind == C_FULL; }   // cell.rs:27-34
};
struct CellIndex {
    int64_t ci = -1;        // index in cells (-1: root)
    uint8_t cj = 0;
    uint32_t depth = 0;
    Bounds bounds;
    CellIndex child(size_t index, int i) const { CellIndex c; c.ci = (int64_t)index; c.cj = (uint8_t)i; c.depth = depth + 1; c.bounds = bounds.child(i); return c; }
};

// ---- qef.rs ------------------------------------------------------------------------------------------
struct V3f { float x, y, z; };
struct Qef {
    float ata[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, atb[3] = {0, 0, 0}, btb = 0, mass[4] = {0, 0, 0, 0};
    void add(const Qef& o) {
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) ata[i][j] += o.ata[i][j]; atb[i] += o.atb[i]; }
        btb += o.btb;
        for (int i = 0; i < 4; i++) mass[i] += o.mass[i];
    }
    // qef.rs:45-59
    void add_intersection(const float* pos, const float* grad) {
        mass[0] += pos[0]; mass[1] += pos[1]; mass[2] += pos[2]; mass[3] += 1.0f;
        const float nn = std::sqrt(0.0f + ((grad[0] * grad[0] + grad[1] * grad[1]) + grad[2] * grad[2]));
        const float n[3] = {grad[0] / nn, grad[1] / nn, grad[2] / nn};
        const float d = (n[0] * pos[0] + n[1] * pos[1]) + n[2] * pos[2];
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) ata[i][j] += n[i] * n[j];
            atb[i] += n[i] * d;
        }
        btb += d * d;
    }
    // qef.rs:67-126 with a Jacobi eigen-decomposition of the symmetric A^T A in place of nalgebra's SVD
    void solve(float* pos, float* err) const {
        const float center[3] = {mass[0] / mass[3], mass[1] / mass[3], mass[2] / mass[3]};
        float b[3];
        for (int i = 0; i < 3; i++) b[i] = atb[i] - ((ata[i][0] * center[0] + ata[i][1] * center[1]) + ata[i][2] * center[2]);
        double a[3][3], v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a[i][j] = ata[i][j];
        for (int sweep = 0; sweep < 32; sweep++) {
            const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
            if (off < 1e-30) break;
            for (int p = 0; p < 2; p++)
                for (int q = p + 1; q < 3; q++) {
                    if (std::fabs(a[p][q]) < 1e-300) continue;
                    const double th = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                    const double tt = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
                    const double c = 1.0 / std::sqrt(tt * tt + 1.0), s = tt * c;
                    for (int k = 0; k < 3; k++) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq; }
                    for (int k = 0; k < 3; k++) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk; }
                    for (int k = 0; k < 3; k++) { const double vkp = v[k][p], vkq = v[k][q]; v[k][p] = c * vkp - s * vkq; v[k][q] = s * vkp + c * vkq; }
                }
        }
        // singular values of a symmetric PSD matrix = |eigenvalues|, sorted descending
        int order[3] = {0, 1, 2};
        std::sort(order, order + 3, [&](int i, int j) { return std::fabs(a[i][i]) > std::fabs(a[j][j]); });
        float sv[3];
        for (int i = 0; i < 3; i++) sv[i] = (float)std::fabs(a[order[i]][order[i]]);
        const float cutoff = std::fabs(sv[0]) * 1e-3f;
        int rank = 3;
        for (int i = 0; i < 3; i++) if (std::fabs(sv[i]) < cutoff) { rank = i; break; }
        const float eps = rank < 3 ? sv[rank] : 0.0f;
        // svd.solve(b, eps): pseudo-inverse keeping singular values > eps
        double sol[3] = {0, 0, 0};
        for (int k = 0; k < 3; k++) {
            const int e = order[k];
            if (!((float)std::fabs(a[e][e]) > eps)) continue;
            const double proj = (v[0][e] * b[0] + v[1][e] * b[1] + v[2][e] * b[2]) / a[e][e];
            for (int i = 0; i < 3; i++) sol[i] += v[i][e] * proj;
        }
        for (int i = 0; i < 3; i++) pos[i] = (float)sol[i] + center[i];
        float ap[3];
        for (int i = 0; i < 3; i++) ap[i] = (ata[i][0] * pos[0] + ata[i][1] * pos[1]) + ata[i][2] * pos[2];
        float e = ((pos[0] * ap[0] + pos[1] * ap[1]) + pos[2] * ap[2]) - 2.0f * ((pos[0] * atb[0] + pos[1] * atb[1]) + pos[2] * atb[2]);
        e += btb;
        *err = e > 1e-6f ? e : 1e-6f;   // .max(1e-6): NaN -> 1e-6 as f32::max ignores NaN
    }
};

// ---- octree.rs:866-1035 --------------------------------------------------------------------------------
static const float QEF_ERR_EMPTY = -1.0f, QEF_ERR_INVALID = -2.0f;
struct LeafIntersection { float pos[4] = {0, 0, 0, 0}, grad[4] = {0, 0, 0, 0}; };
static inline Qef qef_of(const LeafIntersection& i) { Qef q; if (i.pos[3] != 0.0f) q.add_intersection(i.pos, i.grad); return q; }
struct Hermite {
    LeafIntersection inter[12];
    Qef face[6], center;
    float qef_err = QEF_ERR_EMPTY;
    // octree.rs:904-1021
    static bool merge(const Hermite* leafs, Hermite* out) {
        *out = Hermite();
        for (int i = 0; i < 8; i++) if (leafs[i].qef_err == QEF_ERR_INVALID) return false;
        for (int t : {AX, AY, AZ}) {
            const int u = axis_next(t), v = axis_next(u);
            for (int edge = 0; edge < 4; edge++) {
                int start = 0;
                if (edge & 1) start |= u;
                if (edge & 2) start |= v;
                const int end = start | t, e = axis_index(t) * 4 + edge;
                const LeafIntersection &a = leafs[start].inter[e], &b = leafs[end].inter[e];
                if (a.pos[3] > 0.0f && !(b.pos[3] > 0.0f)) out->inter[e] = a;
                else if (!(a.pos[3] > 0.0f) && b.pos[3] > 0.0f) out->inter[e] = b;
            }
        }
        for (int t : {AX, AY, AZ}) {
            const int u = axis_next(t), v = axis_next(t);   // (sic: octree.rs:946-947 takes t.next() twice)
            for (int fc = 0; fc < 2; fc++) {
                const int a = fc == 1 ? t : 0, b = a | u, c = a | v, d = a | u | v, f = axis_index(t) * 2 + fc;
                for (int q : {a, b, c, d}) out->face[f].add(leafs[q].face[f]);
                const int ev = axis_index(v) * 4 + fc * 2 + 1;
                out->face[f].add(qef_of(leafs[a].inter[ev]));
                out->face[f].add(qef_of(leafs[b].inter[ev]));
                const int eu = axis_index(v) * 4 + fc * 2 + 1;
                out->face[f].add(qef_of(leafs[a].inter[eu]));
                out->face[f].add(qef_of(leafs[c].inter[eu]));
            }
        }
        for (int t : {AX, AY, AZ}) {
            const int u = axis_next(t), v = axis_next(t);
            const int a = 0, b = a | u, c = a | v, d = a | u | v;
            for (int q : {a, b, c, d}) out->center.add(leafs[q].face[axis_index(t) * 2 + 1]);
            out->center.add(qef_of(leafs[a].inter[axis_index(u) * 4 + 3]));
            out->center.add(qef_of(leafs[b].inter[axis_index(u) * 4 + 3]));
        }
        for (int i = 0; i < 8; i++) out->center.add(leafs[i].center);
        out->qef_err = INFINITY;
        for (int i = 0; i < 8; i++) if (leafs[i].qef_err >= 0.0f) out->qef_err = rmin(out->qef_err, leafs[i].qef_err);
        return true;
    }
    void solve(float* pos, float* err) const {      // octree.rs:1024-1034
        Qef q = center;
        for (auto& i : inter) q.add(qef_of(i));
        for (auto& f : face) q.add(f);
        q.solve(pos, err);
    }
};

struct MeshOut {
    std::vector<std::array<uint64_t, 3>> triangles;
    std::vector<V3f> vertices;
};

// One sampled leaf, as the device pipeline reproduces it (not part of the reference's data structures)
struct LeafSample {
    Bounds bounds;
    uint8_t mask = 0, n_edges = 0, n_verts = 0;
    uint16_t inter[12][3];
    float pos[12][3], grad[12][4], vert[4][3];
};

struct Octree {
    Cell root;
    std::vector<std::array<Cell, 8>> cells;
    std::vector<V3f> verts;
    std::vector<LeafSample> samples;   // every Leaf produced by leaf(), in evaluation order (collapsed parents are not in here)
    uint64_t interval_evals = 0;
    Cell& at(const CellIndex& c) { return c.ci < 0 ? root : cells[(size_t)c.ci][c.cj]; }
    const Cell& at(const CellIndex& c) const { return c.ci < 0 ? root : cells[(size_t)c.ci][c.cj]; }
    bool is_leaf(const CellIndex& c) const { const uint8_t k = at(c).kind; return k == C_LEAF || k == C_FULL || k == C_EMPTY; }
    CellIndex child(const CellIndex& c, int i) const { const Cell& x = at(c); return x.kind == C_BRANCH ? c.child(x.index, i) : c; }

    // octree.rs:389-470
    bool collapsible(size_t rootc, uint8_t* out_mask) const {
        const auto& cs = cells[rootc];
        const Tables& T = tables();
        int mask = 0;
        for (int i = 0; i < 8; i++) {
            int b;
            if (cs[i].kind == C_LEAF) { if (T.v2e[cs[i].mask].size() > 1) return false; b = (cs[i].mask >> i) & 1; }
            else if (cs[i].kind == C_EMPTY) b = 0;
            else if (cs[i].kind == C_FULL) b = 1;
            else return false;
            mask |= b << i;
        }
        static const int FR[3][3] = {{AX, AY, AZ}, {AY, AZ, AX}, {AZ, AX, AY}};
        for (auto& f : FR) {
            const int t = f[0], u = f[1], v = f[2];
            for (int i = 0; i < 4; i++) {
                const int a = ((i & 1) ? u : 0) | ((i & 2) ? v : 0), b = a | t;
                const bool center = cs[a].corner(b);
                if ((((mask >> a) & 1) != 0) != center && (((mask >> b) & 1) != 0) != center) return false;
            }
            for (int i = 0; i < 2; i++) {
                const int a = ((i & 1) == 0) ? t : 0, b = a | u, c = a | v, d = a | u | v;
                const bool center = cs[a].corner(d);
                bool all = true;
                for (int q : {a, b, c, d}) all &= ((((mask >> q) & 1) != 0) != center);
                if (all) return false;
            }
            const bool center = cs[0].corner(t | u | v);
            bool all = true;
            for (int q = 0; q < 8; q++) all &= ((((mask >> q) & 1) != 0) != center);
            if (all) return false;
        }
        if (T.v2e[mask].size() == 1) { *out_mask = (uint8_t)mask; return true; }
        return false;
    }
    // octree.rs:256-340
    Cell check_done(const CellIndex& cell, size_t index, const Hermite* hd, Hermite* hermite) {
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
                if (err >= hermite->qef_err * 2.0f || !cell.bounds.contains(pos)) ok = false;
            }
            if (ok) {
                hermite->qef_err = err;
                const size_t vi = verts.size();
                verts.push_back(V3f{pos[0], pos[1], pos[2]});
                for (auto& e : tables().v2e[mask][0]) {
                    const LeafIntersection& li = hermite->inter[to_undirected(e.first, e.second)];
                    verts.push_back(V3f{li.pos[0], li.pos[1], li.pos[2]});
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

struct Builder {
    Octree o;
    uint32_t max_depth;
    bool has_mat;
    Mat4 mat;
    Axes axes;
    int mode;
    bool keep_samples = true;          // (the per-leaf sampling records are for the device tests; a depth-10 build would hold 16 GB of them)
    TracingEval<Interval> eval_interval;
    BulkEval<float> eval_float;
    BulkEval<Grad> eval_grad;
    std::vector<Interval> ivars;
    RenderStats st;
    Builder(uint32_t depth, const Mat4* m, const Axes& ax, int mode_) : max_depth(depth), has_mat(m != nullptr), axes(ax), mode(mode_) {
        if (m) mat = *m;
        ivars.resize(std::max(axes.n, 1));
    }

    // ShapeBulkEval::eval_raw over points (shape/mod.rs:719-802): transform, then the tape
    const std::vector<float>& eval_points(const VmData& tape, std::vector<float>& xs, std::vector<float>& ys, std::vector<float>& zs) {
        const size_t n = xs.size();
        if (has_mat) for (size_t i = 0; i < n; i++) transform_f32(mat, xs[i], ys[i], zs[i], &xs[i], &ys[i], &zs[i]);
        std::vector<const float*> vars((size_t)std::max(axes.n, 1), nullptr);
        std::vector<float> zeros(n, 0.0f);
        std::vector<std::vector<float>> bound;
        for (auto& v : vars) v = zeros.data();
        if (axes.ix >= 0) vars[axes.ix] = xs.data();
        if (axes.iy >= 0) vars[axes.iy] = ys.data();
        if (axes.iz >= 0) vars[axes.iz] = zs.data();
        for (auto& b : axes.bound) { bound.push_back(std::vector<float>(n, b.second)); vars[b.first] = bound.back().data(); }
        eval_float.eval(tape, vars.data(), vars.size(), n);
        return eval_float.out[0];
    }

    // octree.rs:521-583
    void recurse(RenderHandle* eval, const CellIndex& cell, Hermite* hermite) {
        Interval tr[3] = {cell.bounds.b[0], cell.bounds.b[1], cell.bounds.b[2]};
        if (has_mat) transform_interval(mat, cell.bounds.b[0], cell.bounds.b[1], cell.bounds.b[2], tr);
        for (auto& v : ivars) v = Interval(0.0f);
        if (axes.ix >= 0) ivars[axes.ix] = tr[0];
        if (axes.iy >= 0) ivars[axes.iy] = tr[1];
        if (axes.iz >= 0) ivars[axes.iz] = tr[2];
        for (auto& b : axes.bound) ivars[b.first] = Interval(b.second);
        const int simplify = eval_interval.eval(*eval->shape, ivars.data(), ivars.size());
        const Interval i = eval_interval.out[0];
        o.interval_evals++;
        Cell res;
        if (i.hi < 0.0f) res.kind = C_FULL;
        else if (i.lo > 0.0f) res.kind = C_EMPTY;
        else {
            RenderHandle* sub = eval;
            if (simplify == 1) sub = eval->simplify(eval_interval.choices, mode, st);   // simplify_tree_during_meshing: always (render/mod.rs:271)
            if (cell.depth == max_depth) res = leaf(sub, cell, hermite);
            else {
                const size_t index = o.cells.size();
                o.cells.push_back(std::array<Cell, 8>());
                Hermite hc[8];
                for (int c = 0; c < 8; c++) recurse(sub, cell.child(index, c), &hc[c]);
                res = o.check_done(cell, index, hc, hermite);
            }
        }
        o.at(cell) = res;
    }

    // octree.rs:590-862
    Cell leaf(RenderHandle* eval, const CellIndex& cell, Hermite* hc) {
        const Tables& T = tables();
        std::vector<float> xs(8), ys(8), zs(8);
        for (int c = 0; c < 8; c++) { float p[3]; cell.bounds.corner(c, p); xs[c] = p[0]; ys[c] = p[1]; zs[c] = p[2]; }
        const std::vector<float>& out = eval_points(*eval->shape, xs, ys, zs);
        int mask = 0;
        for (int c = 0; c < 8; c++) if (out[c] < 0.0f) mask |= 1 << c;
        Cell res;
        if (mask == 0) { res.kind = C_EMPTY; return res; }
        if (mask == 255) { res.kind = C_FULL; return res; }
        uint16_t start[12][3], end[12][3];
        int ne = 0;
        for (auto& vs : T.v2e[mask])
            for (auto& e : vs) {
                const int axis = e.first ^ e.second, ai = axis_index(axis);
                const uint16_t a = (e.second & axis) ? 0 : 65535, b = (e.second & axis) ? 65535 : 0;
                uint16_t v[3] = {0, 0, 0};
                const int i = (ai + 1) % 3, j = (ai + 2) % 3;
                v[i] = (e.first & (1 << i)) ? 65535 : 0;
                v[j] = (e.first & (1 << j)) ? 65535 : 0;
                v[ai] = a; for (int k = 0; k < 3; k++) start[ne][k] = v[k];
                v[ai] = b; for (int k = 0; k < 3; k++) end[ne][k] = v[k];
                ne++;
            }
        const int SEARCH = 16, DEPTH = 4;
        for (int round = 0; round < DEPTH; round++) {
            xs.assign((size_t)ne * SEARCH, 0); ys = xs; zs = xs;
            for (int e = 0, i = 0; e < ne; e++)
                for (int j = 0; j < SEARCH; j++, i++) {
                    uint16_t p[3];
                    for (int k = 0; k < 3; k++) p[k] = (uint16_t)(((uint32_t)start[e][k] * (uint32_t)(SEARCH - j - 1) + (uint32_t)end[e][k] * (uint32_t)j) / (uint32_t)(SEARCH - 1));
                    float f[3];
                    cell.bounds.pos(p, f);
                    xs[i] = f[0]; ys[i] = f[1]; zs[i] = f[2];
                }
            const std::vector<float>& r = eval_points(*eval->shape, xs, ys, zs);
            for (int e = 0; e < ne; e++) {
                int frac = 0;
                while (frac < SEARCH && !(r[(size_t)e * SEARCH + frac] >= 0.0f)) frac++;
                // (the reference unwraps: a search that never turns non-negative panics; inside-to-outside holds for sane fields)
                if (frac == 0) frac = 1;
                if (frac >= SEARCH) frac = SEARCH - 1;
                uint16_t a[3], b[3];
                for (int k = 0; k < 3; k++) {
                    a[k] = (uint16_t)(((uint32_t)start[e][k] * (uint32_t)(SEARCH - (frac - 1) - 1) + (uint32_t)end[e][k] * (uint32_t)(frac - 1)) / (uint32_t)(SEARCH - 1));
                    b[k] = (uint16_t)(((uint32_t)start[e][k] * (uint32_t)(SEARCH - frac - 1) + (uint32_t)end[e][k] * (uint32_t)frac) / (uint32_t)(SEARCH - 1));
                }
                for (int k = 0; k < 3; k++) { start[e][k] = a[k]; end[e][k] = b[k]; }
            }
        }
        LeafSample ls;
        ls.bounds = cell.bounds; ls.mask = (uint8_t)mask; ls.n_edges = (uint8_t)ne;
        std::vector<Grad> gx(ne), gy(ne), gz(ne);
        for (int e = 0; e < ne; e++) {
            for (int k = 0; k < 3; k++) ls.inter[e][k] = (uint16_t)(((uint32_t)start[e][k] + (uint32_t)end[e][k]) / 2);
            cell.bounds.pos(ls.inter[e], ls.pos[e]);
            gx[e] = Grad(ls.pos[e][0], 1, 0, 0); gy[e] = Grad(ls.pos[e][1], 0, 1, 0); gz[e] = Grad(ls.pos[e][2], 0, 0, 1);
        }
        if (has_mat) for (int e = 0; e < ne; e++) { Grad t3[3]; transform_grad(mat, gx[e], gy[e], gz[e], t3); gx[e] = t3[0]; gy[e] = t3[1]; gz[e] = t3[2]; }
        {
            std::vector<const Grad*> vars((size_t)std::max(axes.n, 1), nullptr);
            std::vector<Grad> zeros(ne, Grad(0.0f));
            std::vector<std::vector<Grad>> bound;
            for (auto& v : vars) v = zeros.data();
            if (axes.ix >= 0) vars[axes.ix] = gx.data();
            if (axes.iy >= 0) vars[axes.iy] = gy.data();
            if (axes.iz >= 0) vars[axes.iz] = gz.data();
            for (auto& b : axes.bound) { bound.push_back(std::vector<Grad>(ne, Grad(b.second))); vars[b.first] = bound.back().data(); }
            eval_grad.eval(*eval->shape, vars.data(), vars.size(), (size_t)ne);
        }
        const std::vector<Grad>& grads = eval_grad.out[0];
        for (int e = 0; e < ne; e++) { ls.grad[e][0] = grads[e].dx; ls.grad[e][1] = grads[e].dy; ls.grad[e][2] = grads[e].dz; ls.grad[e][3] = grads[e].v; }
        std::vector<V3f> cv;
        int i = 0;
        for (auto& vs : T.v2e[mask]) {
            bool forced = false;
            float fpos[3] = {0, 0, 0};
            Qef qef;
            for (auto& e : vs) {
                const float* pos = ls.pos[i];
                const float g[4] = {grads[i].dx, grads[i].dy, grads[i].dz, grads[i].v};   // Grad -> Vector4: (dx, dy, dz, v)
                if (g[0] != g[0] || g[1] != g[1] || g[2] != g[2] || g[3] != g[3]) {
                    forced = true; fpos[0] = pos[0]; fpos[1] = pos[1]; fpos[2] = pos[2];
                    hc->qef_err = QEF_ERR_INVALID;
                    break;      // (octree.rs:810: `i` is not advanced for the remaining edges of this vertex)
                }
                qef.add_intersection(pos, g);
                LeafIntersection& li = hc->inter[to_undirected(e.first, e.second)];
                li.pos[0] = pos[0]; li.pos[1] = pos[1]; li.pos[2] = pos[2]; li.pos[3] = 1.0f;
                for (int k = 0; k < 4; k++) li.grad[k] = g[k];
                i++;
            }
            if (forced) cv.push_back(V3f{fpos[0], fpos[1], fpos[2]});
            else {
                float p[3], err;
                qef.solve(p, &err);
                cv.push_back(V3f{p[0], p[1], p[2]});
                hc->qef_err = err;
            }
        }
        ls.n_verts = (uint8_t)cv.size();
        for (size_t k = 0; k < cv.size() && k < 4; k++) { ls.vert[k][0] = cv[k].x; ls.vert[k][1] = cv[k].y; ls.vert[k][2] = cv[k].z; }
        if (keep_samples) o.samples.push_back(ls);
        const size_t index = o.verts.size();
        for (auto& v : cv) o.verts.push_back(v);
        for (int e = 0; e < ne; e++) o.verts.push_back(V3f{ls.pos[e][0], ls.pos[e][1], ls.pos[e][2]});
        res.kind = C_LEAF; res.mask = (uint8_t)mask; res.index = (uint32_t)index;
        return res;
    }
};

// ---- dc.rs / builder.rs: the dual walk --------------------------------------------------------------------
// Octree::build_inner_mt (octree.rs:94-210): cells split off breadth-first until there are >= 10 x threads of them (never evaluated:
// they are taken to be ambiguous), every one built as an octree of its own by a worker with a fresh RenderHandle on the ROOT tape,
// the sub-octrees appended in task order with their cell and vertex indices shifted (176-195), then check_done over the split cells
// in reverse (197-208).  The cell / vertex LAYOUT differs from the single-threaded build's (and depends on `threads` through the task
// count, as in the reference); the tree - and so walk_dual's mesh - does not.
static inline Octree build_mt(VmDataP shape, uint32_t depth, const Mat4* m, const Axes& axes, int mode, int threads, bool keep_samples, uint64_t* interval_evals) {
    Octree root;
    std::deque<CellIndex> todo;
    todo.push_back(CellIndex());
    std::vector<std::pair<CellIndex, size_t>> fixup;
    std::vector<std::array<Hermite, 8>> hermites;
    size_t pow8 = 1;
    for (uint32_t i = 0; i < depth && pow8 < ((size_t)1 << 40); i++) pow8 *= 8;
    const size_t target = std::min(pow8, (size_t)std::max(threads, 1) * 10);
    while (todo.size() < target) {
        const CellIndex next = todo.front();
        todo.pop_front();
        const size_t index = root.cells.size();
        root.cells.push_back(std::array<Cell, 8>());
        hermites.push_back(std::array<Hermite, 8>());
        for (int i = 0; i < 8; i++) todo.push_back(next.child(index, i));
        fixup.push_back({next, index});
    }
    struct Output { Octree octree; Hermite hermite; };
    std::vector<CellIndex> tasks(todo.begin(), todo.end());
    std::vector<Output> out(tasks.size());
    uint64_t evals = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(std::max(threads, 1)) reduction(+ : evals)
    for (int64_t t = 0; t < (int64_t)tasks.size(); t++) {
        Builder b(depth, m, axes, mode);
        b.keep_samples = keep_samples;
        RenderHandle rh(shape);
        CellIndex local = tasks[(size_t)t];
        local.ci = -1; local.cj = 0;            // "patch our cell so that it builds at index 0" (octree.rs:139-143): depth and bounds stay
        b.recurse(&rh, local, &out[(size_t)t].hermite);
        evals += b.o.interval_evals;
        out[(size_t)t].octree = std::move(b.o);
    }
    std::vector<size_t> cell_off{root.cells.size()}, vert_off{0};
    for (size_t i = 0; i < out.size(); i++) {
        hermites[(size_t)tasks[i].ci][tasks[i].cj] = out[i].hermite;
        cell_off.push_back(cell_off.back() + out[i].octree.cells.size());
        vert_off.push_back(vert_off.back() + out[i].octree.verts.size());
    }
    root.cells.reserve(cell_off.back());
    root.verts.reserve(vert_off.back());
    for (size_t i = 0; i < out.size(); i++) {
        auto remap = [&](Cell c) {
            if (c.kind == C_LEAF) c.index += (uint32_t)vert_off[i];
            else if (c.kind == C_BRANCH) c.index += (uint32_t)cell_off[i];
            return c;
        };
        for (auto& cs : out[i].octree.cells) { std::array<Cell, 8> r; for (int k = 0; k < 8; k++) r[k] = remap(cs[k]); root.cells.push_back(r); }
        root.verts.insert(root.verts.end(), out[i].octree.verts.begin(), out[i].octree.verts.end());
        if (keep_samples) root.samples.insert(root.samples.end(), out[i].octree.samples.begin(), out[i].octree.samples.end());
        root.at(tasks[i]) = remap(out[i].octree.root);
        out[i].octree = Octree();
    }
    for (size_t k = fixup.size(); k-- > 0;) {
        const CellIndex& cell = fixup[k].first;
        const size_t index = fixup[k].second;
        const std::array<Hermite, 8> h = hermites[index];
        Hermite scratch;
        Hermite* dst = cell.ci >= 0 ? &hermites[(size_t)cell.ci][cell.cj] : &scratch;
        const Cell r = root.check_done(cell, index, h.data(), dst);
        root.at(cell) = r;
    }
    root.interval_evals = evals;
    if (interval_evals) *interval_evals = evals;
    return root;
}

struct Walker {
    const Octree& o;
    MeshOut out;
    std::vector<size_t> map;
    explicit Walker(const Octree& oc) : o(oc) {}
    static void frame(int f, int* t, int* u, int* v) { static const int FR[3][3] = {{AX, AY, AZ}, {AY, AZ, AX}, {AZ, AX, AY}}; *t = FR[f][0]; *u = FR[f][1]; *v = FR[f][2]; }
    size_t vertex(size_t v) {
        if (v >= map.size()) map.resize(v + 1, (size_t)-1);
        if (map[v] == (size_t)-1) { map[v] = out.vertices.size(); out.vertices.push_back(o.verts[v]); }
        return map[v];
    }
    void cell(const CellIndex& c) {      // dc.rs dc_cell
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
    void face(int f, const CellIndex& lo, const CellIndex& hi) {     // dc.rs dc_face
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
    void edge(int f, const CellIndex& a, const CellIndex& b, const CellIndex& c, const CellIndex& d) {    // dc.rs dc_edge
        const CellIndex cs[4] = {a, b, c, d};
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
        const bool starting_sign = st;
        const Tables& T = tables();
        int vv[4][2];
        for (int i = 0; i < 4; i++) {
            if (cs[i].depth == cs[deepest].depth) { vv[i][0] = T.e2v[leafs[i].mask][edges[i]][0]; vv[i][1] = T.e2v[leafs[i].mask][edges[i]][1]; }
            else {
                vv[i][0] = vv[i][1] = -1;
                for (int j = 0; j < 12; j++) if (T.e2v[leafs[i].mask][j][0] >= 0) { vv[i][0] = T.e2v[leafs[i].mask][j][0]; vv[i][1] = T.e2v[leafs[i].mask][j][1]; break; }
            }
            if (vv[i][0] < 0) return;   // (the reference unwraps)
        }
        const size_t iv = vertex(leafs[deepest].index + (size_t)vv[deepest][1]);
        size_t vs[4];
        for (int i = 0; i < 4; i++) vs[i] = vertex(leafs[i].index + (size_t)vv[i][0]);
        const int winding = starting_sign ? 3 : 1;
        for (int j = 0; j < 4; j++) {
            const CellIndex &p = cs[j], &q = cs[(j + winding) % 4];
            if (p.ci != q.ci || p.cj != q.cj) out.triangles.push_back({vs[j], vs[(j + winding) % 4], iv});
        }
    }
};

}  // namespace mesh
}  // namespace orc
