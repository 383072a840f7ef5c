// fidget-hip: the edge search of fidget-mesh's leaf sampling (fidget-mesh/src/octree.rs:662-803) as per-edge steps - the end points of
// a Manifold-DC edge in u16 cell coordinates, a round's 16 sample points, the bracket narrowed by the samples' signs, the intersection -
// so that the rounds can run as passes over ALL edges of a chunk of cells with the values coming from the assembly bulk interpreter
// (mesh.hip: k_mesh_edge_*), and be checked on the host against the oracle's samples (tests/test_mesh_edges.py).  The arithmetic is
// k_mesh_leaf's, operation for operation.
#pragma once
#include <stdint.h>

#include "mesh_qef.hpp"      // FHQ_HD

namespace fhmesh {

struct EdgeBracket { uint16_t s[3], t[3]; };      // the two ends of the bracket, u16 cell coordinates

FHQ_HD static inline float lerp_pos(float lo, float hi, uint32_t p) {   // cell.rs:208-217, Interval::lerp
    const float f = (float)p / 65535.0f;
    return lo * (1.0f - f) + hi * f;
}
// the edge from corner `st` (inside) to corner `en` (outside) of the cell (octree.rs:662-695)
FHQ_HD static inline EdgeBracket edge_ends(int st, int en) {
    EdgeBracket b;
    const int axis = st ^ en, ai = axis == 1 ? 0 : (axis == 2 ? 1 : 2);
    const int i1 = (ai + 1) % 3, i2 = (ai + 2) % 3;
    uint16_t p[3] = {0, 0, 0};
    p[i1] = (st & (1 << i1)) ? 65535 : 0;
    p[i2] = (st & (1 << i2)) ? 65535 : 0;
    for (int q = 0; q < 3; q++) { b.s[q] = p[q]; b.t[q] = p[q]; }
    b.s[ai] = (en & axis) ? 0 : 65535;
    b.t[ai] = (en & axis) ? 65535 : 0;
    return b;
}
// sample j (0..15) of a round: between the ends, in u16 cell coordinates (octree.rs:715-730)
FHQ_HD static inline void edge_sample(const EdgeBracket& b, uint32_t j, uint32_t* p) {
    for (int q = 0; q < 3; q++) p[q] = ((uint32_t)b.s[q] * (15u - j) + (uint32_t)b.t[q] * j) / 15u;
}
// the bracket after a round: m16 bit j = sample j is >= 0 (outside); the first such sample and the one before it (octree.rs:732-768)
FHQ_HD static inline EdgeBracket edge_narrow(const EdgeBracket& b, uint32_t m16) {
    uint32_t frac = 16;
    for (uint32_t j = 0; j < 16; j++) if ((m16 >> j) & 1u) { frac = j; break; }
    if (frac == 0) frac = 1;
    if (frac > 15) frac = 15;
    EdgeBracket r;
    for (int q = 0; q < 3; q++) {
        r.s[q] = (uint16_t)(((uint32_t)b.s[q] * (15u - (frac - 1)) + (uint32_t)b.t[q] * (frac - 1)) / 15u);
        r.t[q] = (uint16_t)(((uint32_t)b.s[q] * (15u - frac) + (uint32_t)b.t[q] * frac) / 15u);
    }
    return r;
}
// the intersection: the middle of the last bracket (octree.rs:771-780)
FHQ_HD static inline void edge_mid(const EdgeBracket& b, uint16_t* q) {
    for (int k = 0; k < 3; k++) q[k] = (uint16_t)(((uint32_t)b.s[k] + (uint32_t)b.t[k]) / 2u);
}

}  // namespace fhmesh
