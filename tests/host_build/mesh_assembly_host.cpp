// TEST INFRASTRUCTURE: the per-cell functions and the pass order of the device's octree assembly (fidget_amd/csrc/mesh_collapse.hpp)
// compiled for the host, every pass a plain loop, so that tests/test_mesh_assembly.py can check them against the oracle's octree
// without a GPU.  Nothing in the product links or loads this.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "mesh_collapse.hpp"

using namespace fhmesh;
namespace {
struct HostX {
    std::vector<void*> owned;
    ~HostX() { for (void* p : owned) free(p); }
    void* alloc(size_t b) { void* p = malloc(b ? b : 1); owned.push_back(p); return p; }
    void zero(void* p, size_t b) { memset(p, 0, b); }
    void read(void* d, const void* s, size_t b) { memcpy(d, s, b); }
    void kind(const OctLevel& D, const OctLevel& C, const OctLeaves& L, const FhMdcTable* T, uint32_t* counter, uint32_t n) {
        for (uint32_t s = n; s-- > 0;) oct_kind(D, C, L, T, s, counter);      // (backwards: the order of the candidates must not matter)
    }
    void collapse(const OctLevel& D, const OctLevel& C, const OctLeaves& L, const FhMdcTable* T, uint32_t n) {
        for (uint32_t k = 0; k < n; k++) oct_collapse(D, C, L, T, k);
    }
    void place(const OctLevel& D, const OctLevel& C, const OctLeaves& L, const FhMdcTable* T, Cell* cells, V3* verts, const float* mat, uint32_t n) {
        for (uint32_t s = n; s-- > 0;) oct_place(D, C, L, T, s, cells, verts, mat);
    }
    void leaf_verts(const OctLeaves& L, V3* verts, const float* mat, uint32_t n) {
        for (uint32_t i = 0; i < n; i++) oct_leaf_verts(L, i, verts, mat);
    }
};
struct Run { HostX x; OctOut out; uint32_t collapsed = 0; };
}  // namespace

extern "C" {
// levels flattened: n_cells[d] classes / slots of level d one after the other; n_amb[d] bounds (6 floats) of its ambiguous cells by slot
void* fh_asm_run(uint32_t depth, uint32_t n_levels, const uint32_t* n_cells, const uint8_t* cls, const uint32_t* slot, const uint32_t* n_amb, const float* bounds,
                 const void* rec, uint32_t n_rec, const void* table, const float* mat) {
    Run* r = new Run();
    std::vector<OctLevel> lv(n_levels);
    std::vector<std::vector<FhMeshCell>> amb(n_levels);
    size_t co = 0, bo = 0;
    for (uint32_t d = 0; d < n_levels; d++) {
        lv[d].cls = cls + co; lv[d].slot = slot + co; lv[d].n_amb = n_amb[d];
        amb[d].resize(n_amb[d]);
        for (uint32_t s = 0; s < n_amb[d]; s++) { for (int k = 0; k < 6; k++) amb[d][s].b[k] = bounds[(bo + s) * 6 + k]; amb[d][s].path = 0; }
        lv[d].amb = amb[d].data();
        co += n_cells[d]; bo += n_amb[d];
    }
    if (oct_assemble(r->x, depth, lv.data(), n_levels, (const FhMeshLeaf*)rec, n_rec, (const FhMdcTable*)table, mat, &r->out) != OCT_OK) { delete r; return nullptr; }
    for (uint32_t d = 0; d + 1 < n_levels; d++)
        for (uint32_t s = 0; lv[d].res && s < lv[d].n_amb; s++) r->collapsed += lv[d].res[s].kind == C_LEAF;
    return r;
}
void fh_asm_counts(const void* h, uint32_t* out /* root kind, mask, index, blocks, vertices, collapsed cells */) {
    const Run* r = (const Run*)h;
    out[0] = r->out.root.kind; out[1] = r->out.root.mask; out[2] = r->out.root.index; out[3] = r->out.n_blocks; out[4] = r->out.n_verts; out[5] = r->collapsed;
}
void fh_asm_copy(const void* h, uint32_t* cells /* [blocks][8][3] kind, mask, index */, float* verts) {
    const Run* r = (const Run*)h;
    for (size_t i = 0; i < (size_t)r->out.n_blocks * 8; i++) { const Cell& c = r->out.cells[i]; cells[3 * i] = c.kind; cells[3 * i + 1] = c.mask; cells[3 * i + 2] = c.index; }
    if (r->out.n_verts) memcpy(verts, r->out.verts, (size_t)r->out.n_verts * sizeof(V3));
}
void fh_asm_free(void* h) { delete (Run*)h; }
uint32_t fh_asm_sizes(uint32_t which) { return which == 0 ? sizeof(FhMeshLeaf) : (which == 1 ? sizeof(FhMdcTable) : sizeof(OctCollapsed)); }
}
