// TEST INFRASTRUCTURE: the per-edge steps of the mesh path's edge search (fidget_amd/csrc/mesh_edges.hpp) built for the host, so that
// tests/test_mesh_edges.py can drive the four rounds with the oracle's f32 evaluator and compare with the oracle's own leaf samples.
#include <string.h>

#include "mesh_edges.hpp"

using namespace fhmesh;
extern "C" {
void fh_edge_begin(int st, int en, uint16_t* br /* s[3], t[3] */) { const EdgeBracket b = edge_ends(st, en); memcpy(br, &b, sizeof(b)); }
void fh_edge_points(const uint16_t* br, const float* bounds, float* xyz /* [16][3] */) {
    EdgeBracket b; memcpy(&b, br, sizeof(b));
    for (uint32_t j = 0; j < 16; j++) {
        uint32_t p[3];
        edge_sample(b, j, p);
        for (int q = 0; q < 3; q++) xyz[3 * j + q] = lerp_pos(bounds[2 * q], bounds[2 * q + 1], p[q]);
    }
}
void fh_edge_narrow(uint16_t* br, uint32_t m16) { EdgeBracket b; memcpy(&b, br, sizeof(b)); b = edge_narrow(b, m16); memcpy(br, &b, sizeof(b)); }
void fh_edge_end(const uint16_t* br, const float* bounds, uint16_t* q, float* pos) {
    EdgeBracket b; memcpy(&b, br, sizeof(b));
    edge_mid(b, q);
    for (int k = 0; k < 3; k++) pos[k] = lerp_pos(bounds[2 * k], bounds[2 * k + 1], q[k]);
}
uint32_t fh_edge_bracket_bytes() { return sizeof(EdgeBracket); }
}
