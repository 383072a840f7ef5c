// Fragment of capi.hip (meshing (the only fragment the mesh path owns: tools/src_hash.py leaves it out of the render path's hash)); not a stand-alone header: included by capi.hip only.
// ---- meshing: the evaluation side of fidget_mesh::Octree::build (fidget-mesh/src/octree.rs) --------------------------------
// CELL_TO_VERT_TO_EDGES of fidget-mesh/build.rs:26-160: per corner mask, the inside -> outside edges grouped into cell vertices
// by connected region (filled regions first, then empty ones, each in ascending order of their corner sets)
static void build_mdc_table(FhMdcTable& T) {
    auto next = [](int a) { return (a << 1) > 4 ? 1 : (a << 1); };
    for (int i = 0; i < 256; i++) {
        int region_of[2][8];
        for (int pass = 0; pass < 2; pass++) {
            int* r = region_of[pass];
            for (int j = 0; j < 8; j++) r[j] = 1 << j;
            for (bool changed = true; changed;) {
                changed = false;
                for (int f = 0; f < 8; f++) {
                    if ((((i >> f) & 1) != 0) != (pass == 0)) continue;
                    for (int axis : {1, 2, 4}) {
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
        std::vector<std::pair<int, std::vector<std::pair<int, int>>>> verts;
        for (int rev = 0; rev < 2; rev++)
            for (int t : {1, 2, 4}) {
                const int u = next(t), v = next(u);
                for (int b = 0; b < 2; b++)
                    for (int a = 0; a < 2; a++) {
                        int start = (a * u) | (b * v), end = start | t;
                        if (rev) std::swap(start, end);
                        if (!(((i >> start) & 1) && !((i >> end) & 1))) continue;
                        auto it = std::find_if(verts.begin(), verts.end(), [&](auto& kv) { return kv.first == regions[start]; });
                        if (it == verts.end()) { verts.push_back({regions[start], {}}); it = verts.end() - 1; }
                        it->second.push_back({start, end});
                    }
            }
        std::sort(verts.begin(), verts.end(), [](auto& a, auto& b) { return a.first < b.first; });
        T.n_verts[i] = (uint8_t)verts.size();
        int ne = 0;
        for (int k = 0; k < 4; k++) T.per_vert[i][k] = 0;
        for (size_t vi = 0; vi < verts.size(); vi++) {
            T.per_vert[i][vi] = (uint8_t)verts[vi].second.size();
            for (auto& e : verts[vi].second) { T.edge[i][ne][0] = (uint8_t)e.first; T.edge[i][ne][1] = (uint8_t)e.second; ne++; }
        }
        T.n_edges[i] = (uint8_t)ne;
    }
}
struct fhip_mesh {
    // leaf records, in pinned host memory (the device writes them there in chunks while the leaf kernel is still running)
    struct PinnedLeaves {
        FhMeshLeaf* p = nullptr;
        size_t n = 0;
        bool borrowed = false;      // the context's cached area (fhip_mesh_build): not kept with the mesh
        // fhip_mesh_merge: the records stay where the parts' buffers hold them; segment k covers records seg_start[k] .. seg_start[k + 1] - 1
        std::vector<const FhMeshLeaf*> seg_p;
        std::vector<size_t> seg_start;
        const FhMeshLeaf& operator[](size_t i) const {
            if (seg_p.empty()) return p[i];
            size_t k = 0;
            while (k + 1 < seg_p.size() && i >= seg_start[k + 1]) k++;
            return seg_p[k][i - seg_start[k]];
        }
        const FhMeshLeaf* data() const { return p; }
        size_t size() const { return n; }
        ~PinnedLeaves() { if (p && !borrowed) (void)hipHostFree(p); }
    } leaves;
    uint64_t cells_evaluated = 0, full = 0, empty = 0, ambiguous_leaves = 0;
    std::vector<uint64_t> per_level;   // cells evaluated at each depth
    // per level, per evaluated cell: class (1 empty 2 full 3 ambiguous) and, for ambiguous cells, their index among the level's
    // ambiguous cells (= parent index of their children / leaf record index)
    std::vector<std::vector<uint8_t>> cls;
    std::vector<std::vector<uint32_t>> slot;
    fhmesh::VertVec vertices;                            // fhip_mesh_build: Mesh::vertices
    fhmesh::TriVec triangles;                            // ... Mesh::triangles
    uint64_t octree_cells = 0, octree_verts = 0;
    uint64_t sub_skipped = 0;                            // ... or how many there would have been, when they were not worth using
    uint64_t sub_tapes = 0, sub_ops = 0;                 // tapes simplified at the split level and their ops together (0: the root tape everywhere)
    uint32_t depth = 0, part = 0, n_parts = 1;           // fhip_mesh_sample_part: which of the root's octants this one covers
};
// Assembly of the octree from the device's results, as Octree::recurse unwinds (octree.rs:556-583), then Octree::walk_dual
struct MeshAssembler {
    const fhip_mesh& M;
    uint32_t depth;
    fhmesh::Octree o;
    fhmesh::Cell build(uint32_t d, size_t i, const float* b, fhmesh::Hermite* hermite) {
        fhmesh::Cell res;
        const uint8_t c = M.cls[d][i];
        if (c == 2) { res.kind = fhmesh::C_FULL; return res; }
        if (c == 1) { res.kind = fhmesh::C_EMPTY; return res; }
        const uint32_t s = M.slot[d][i];
        if (d == depth) {       // leaf() (octree.rs:590-862) with the device's samples
            const FhMeshLeaf& lf = M.leaves[s];
            if (lf.mask == 0) { res.kind = fhmesh::C_EMPTY; return res; }
            if (lf.mask == 255) { res.kind = fhmesh::C_FULL; return res; }
            const fhmesh::Tables& T = fhmesh::tables();
            uint32_t ii = 0, vi = 0;
            for (auto& vs : T.v2e[lf.mask]) {
                bool forced = false;
                for (auto& e : vs) {
                    const uint32_t k = std::min<uint32_t>(ii, 11);
                    const float* g = lf.grad[k];
                    if (g[0] != g[0] || g[1] != g[1] || g[2] != g[2] || g[3] != g[3]) { forced = true; hermite->qef_err = fhmesh::QEF_ERR_INVALID; break; }
                    fhmesh::LeafIntersection& li = hermite->inter[fhmesh::to_undirected(e.first, e.second)];
                    li.pos[0] = lf.pos[k][0]; li.pos[1] = lf.pos[k][1]; li.pos[2] = lf.pos[k][2]; li.pos[3] = 1.0f;
                    for (int q = 0; q < 4; q++) li.grad[q] = g[q];
                    ii++;
                }
                if (!forced) hermite->qef_err = lf.qef_err[vi];
                vi++;
            }
            res.kind = fhmesh::C_LEAF; res.mask = (uint8_t)lf.mask; res.index = (uint32_t)o.verts.size();
            for (uint32_t v = 0; v < lf.n_verts; v++) o.verts.push_back(fhmesh::V3{lf.vert[v][0], lf.vert[v][1], lf.vert[v][2]});
            for (uint32_t e = 0; e < lf.n_edges; e++) o.verts.push_back(fhmesh::V3{lf.pos[e][0], lf.pos[e][1], lf.pos[e][2]});
            return res;
        }
        const size_t index = o.cells.size();
        o.cells.push_back(std::array<fhmesh::Cell, 8>());
        fhmesh::Hermite hc[8];
        for (int corner = 0; corner < 8; corner++) {
            float cb[6];
            for (int k = 0; k < 3; k++) {
                const float mid = (b[2 * k] + b[2 * k + 1]) / 2.0f;        // cell.rs:184-194
                if (corner & (1 << k)) { cb[2 * k] = mid; cb[2 * k + 1] = b[2 * k + 1]; } else { cb[2 * k] = b[2 * k]; cb[2 * k + 1] = mid; }
            }
            const fhmesh::Cell ch = build(d + 1, (size_t)s * 8 + corner, cb, &hc[corner]);
            o.cells[index][corner] = ch;
        }
        return o.check_done(b, index, hc, hermite);
    }
};
// The same assembly by independent subtrees on the host's threads, with the sequential recursion's result cell for cell and
// vertex for vertex (as Octree::build_inner_mt does with its thread pool, octree.rs:94-210, but spliced in recursion order):
// the ambiguous cells of level L are built each into an octree of its own, then the levels above them are assembled
// sequentially and take the subtrees in the order the recursion reaches them (cell / vertex indices shifted to where the
// recursion would have put them - check_done's bookkeeping only ever looks at the end of the arrays, which a subtree owns).
struct ParallelMeshAssembler {
    const fhip_mesh& M;
    uint32_t depth, L;
    fhmesh::Octree o;
    struct Task { size_t i; float b[6]; };
    struct Sub { fhmesh::Octree o; fhmesh::Cell root; fhmesh::Hermite h; size_t co = 0, vo = 0; };
    static fhmesh::Cell shift(fhmesh::Cell x, size_t co, size_t vo) {
        if (x.kind == fhmesh::C_BRANCH) x.index += (uint32_t)co;
        else if (x.kind == fhmesh::C_LEAF) x.index += (uint32_t)vo;
        return x;
    }
    std::vector<Task> tasks;
    std::vector<Sub> subs;
    size_t next = 0;
    static void child_bounds(const float* b, int corner, float* cb) {
        for (int k = 0; k < 3; k++) {
            const float mid = (b[2 * k] + b[2 * k + 1]) / 2.0f;        // cell.rs:184-194
            if (corner & (1 << k)) { cb[2 * k] = mid; cb[2 * k + 1] = b[2 * k + 1]; } else { cb[2 * k] = b[2 * k]; cb[2 * k + 1] = mid; }
        }
    }
    void plan(uint32_t d, size_t i, const float* b) {
        if (M.cls[d][i] != 3) return;
        if (d == L) { Task t; t.i = i; for (int k = 0; k < 6; k++) t.b[k] = b[k]; tasks.push_back(t); return; }
        const uint32_t s = M.slot[d][i];
        for (int corner = 0; corner < 8; corner++) { float cb[6]; child_bounds(b, corner, cb); plan(d + 1, (size_t)s * 8 + corner, cb); }
    }
    fhmesh::Cell top(uint32_t d, size_t i, const float* b, fhmesh::Hermite* hermite) {
        fhmesh::Cell res;
        const uint8_t c = M.cls[d][i];
        if (c == 2) { res.kind = fhmesh::C_FULL; return res; }
        if (c == 1) { res.kind = fhmesh::C_EMPTY; return res; }
        if (d == L) {       // splice the subtree
            Sub& S = subs[next++];
            // (room now, contents later and in parallel: nothing above this level ever reads inside a subtree)
            const size_t co = o.cells.size(), vo = o.verts.size();
            S.co = co; S.vo = vo;
            o.cells.resize(co + S.o.cells.size());
            o.verts.resize(vo + S.o.verts.size());
            *hermite = S.h;
            return shift(S.root, co, vo);
        }
        const uint32_t s = M.slot[d][i];
        const size_t index = o.cells.size();
        o.cells.push_back(std::array<fhmesh::Cell, 8>());
        fhmesh::Hermite hc[8];
        for (int corner = 0; corner < 8; corner++) {
            float cb[6];
            child_bounds(b, corner, cb);
            const fhmesh::Cell ch = top(d + 1, (size_t)s * 8 + corner, cb, &hc[corner]);
            o.cells[index][corner] = ch;
        }
        return o.check_done(b, index, hc, hermite);
    }
    fhmesh::Cell run(const float* rb, fhmesh::Hermite* h) {
        fhmesh::tables();
        const bool times = getenv("FHIP_MESH_TIMES") != nullptr;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t0 = now();
        plan(0, 0, rb);
        subs.resize(tasks.size());
        fhmesh::parallel_for(tasks.size(), [&](size_t k) {
            MeshAssembler A{M, depth, {}};
            subs[k].root = A.build(L, tasks[k].i, tasks[k].b, &subs[k].h);
            subs[k].o = std::move(A.o);
        });
        const double t1 = now();
        size_t total_c = 64, total_v = 64;
        for (auto& S : subs) { total_c += S.o.cells.size() + 1; total_v += S.o.verts.size(); }
        o.cells.reserve(total_c + 600 * tasks.size() / 512 + 4096);
        o.verts.reserve(total_v + 4096);
        const fhmesh::Cell root = top(0, 0, rb, h);
        const double t2 = now();
        fhmesh::parallel_for(subs.size(), [&](size_t k) {
            Sub& S = subs[k];
            // (a top-level collapse may have cut the arrays back below this subtree: then it is unreachable and not copied)
            if (S.co + S.o.cells.size() <= o.cells.size())
                for (size_t i = 0; i < S.o.cells.size(); i++) for (int q = 0; q < 8; q++) o.cells[S.co + i][q] = shift(S.o.cells[i][q], S.co, S.vo);
            if (S.vo + S.o.verts.size() <= o.verts.size() && !S.o.verts.empty())
                memcpy(&o.verts[S.vo], S.o.verts.data(), S.o.verts.size() * sizeof(fhmesh::V3));
            S.o = fhmesh::Octree();
        });
        if (times) fprintf(stderr, "fhip mesh assembly: %zu subtrees below level %u %.4f s, levels above + room %.4f s, splice %.4f s\n", tasks.size(), L, t1 - t0, t2 - t1, now() - t2);
        return root;
    }
};
// the root's octants part `part` of `n_parts` evaluates: octant o belongs to part o * n_parts / 8 (8 parts: one octant each, as
// Octree::build_inner_mt hands the root's children to its workers, octree.rs:109-123; 2 parts: the z halves)
static uint32_t mesh_part_mask(uint32_t part, uint32_t n_parts) {
    uint32_t m = 0;
    for (uint32_t o = 0; o < 8; o++) if (o * n_parts / 8 == part) m |= 1u << o;
    return m;
}
struct MeshTimes { bool on; double t_start, t_cells, t_leaf, t_copy; uint32_t n_leaf_cells; };
static void mesh_assemble(fhip_ctx* ctx, fhip_mesh* M, uint32_t depth, bool has_mat, const float* mat, MeshTimes& T);
// The octree assembled on the device (mesh_collapse.hpp oct_assemble; kernels in mesh.hip): arrays in HBM, one launch per pass and level
struct OctDevX {
    hipStream_t st;
    std::vector<void*> owned;
    hipError_t err = hipSuccess;
    void chk(hipError_t e) { if (e != hipSuccess && err == hipSuccess) err = e; }
    void* alloc(size_t b) {
        void* p = nullptr;
        const hipError_t e = hipMalloc(&p, b ? b : 4);
        if (e != hipSuccess) { chk(e); return nullptr; }
        owned.push_back(p);
        return p;
    }
    void zero(void* p, size_t b) { chk(hipMemsetAsync(p, 0, b, st)); }
    void read(void* d, const void* s, size_t b) { chk(hipMemcpyAsync(d, s, b, hipMemcpyDeviceToHost, st)); chk(hipStreamSynchronize(st)); }
    void kind(const fhmesh::OctLevel& D, const fhmesh::OctLevel& C, const fhmesh::OctLeaves& L, const FhMdcTable* T, uint32_t* counter, uint32_t n) {
        if (!n) return;
        hipLaunchKernelGGL(fhm::k_oct_kind, dim3((n + 255) / 256), dim3(256), 0, st, D, C, L, T, counter, n);
        chk(hipGetLastError());
    }
    void collapse(const fhmesh::OctLevel& D, const fhmesh::OctLevel& C, const fhmesh::OctLeaves& L, const FhMdcTable* T, uint32_t n) {
        if (!n) return;
        hipLaunchKernelGGL(fhm::k_oct_collapse, dim3((n + 63) / 64), dim3(64), 0, st, D, C, L, T, n);
        chk(hipGetLastError());
    }
    void place(const fhmesh::OctLevel& D, const fhmesh::OctLevel& C, const fhmesh::OctLeaves& L, const FhMdcTable* T, fhmesh::Cell* cells, fhmesh::V3* verts, const float* mat, uint32_t n) {
        if (!n) return;
        hipLaunchKernelGGL(fhm::k_oct_place, dim3((n + 255) / 256), dim3(256), 0, st, D, C, L, T, cells, verts, mat, n);
        chk(hipGetLastError());
    }
    void leaf_verts(const fhmesh::OctLeaves& L, fhmesh::V3* verts, const float* mat, uint32_t n) {
        if (!n) return;
        hipLaunchKernelGGL(fhm::k_oct_leaf_verts, dim3((n + 255) / 256), dim3(256), 0, st, L, verts, mat, n);
        chk(hipGetLastError());
    }
    void release() { for (void* p : owned) (void)hipFree(p); owned.clear(); }
};
// Octree::walk_dual on the device: where mesh_walk.hpp's arrays live and how its passes run (one kernel launch per pass)
struct WalkDevX {
    hipStream_t st;
    std::vector<void*> owned;
    hipError_t err = hipSuccess;
    void chk(hipError_t e) { if (e != hipSuccess && err == hipSuccess) err = e; }
    void* alloc(size_t b) {
        void* p = nullptr;
        const hipError_t e = hipMalloc(&p, b ? b : 4);
        if (e != hipSuccess) { chk(e); return nullptr; }
        owned.push_back(p);
        return p;
    }
    void free(void* p) {
        for (size_t i = owned.size(); i-- > 0;) if (owned[i] == p) { owned.erase(owned.begin() + (long)i); break; }
        (void)hipFree(p);
    }
    void forget(void* p) { for (size_t i = owned.size(); i-- > 0;) if (owned[i] == p) { owned.erase(owned.begin() + (long)i); break; } }   // the caller keeps it
    void release() { for (void* p : owned) (void)hipFree(p); owned.clear(); }
    void zero(void* p, size_t b) { chk(hipMemsetAsync(p, 0, b, st)); }
    void fill_ff(void* p, size_t b) { if (b) chk(hipMemsetAsync(p, 0xFF, b, st)); }
    void read(void* d, const void* s, size_t b) { chk(hipMemcpyAsync(d, s, b, hipMemcpyDeviceToHost, st)); chk(hipStreamSynchronize(st)); }
    void write(void* d, const void* s, size_t b) { chk(hipMemcpyAsync(d, s, b, hipMemcpyHostToDevice, st)); chk(hipStreamSynchronize(st)); }
    static dim3 grid(uint64_t n) { return dim3((unsigned)((n + 255) / 256)); }
    bool scan(const uint32_t* in, uint32_t n, uint32_t* out) {
        const uint32_t nb = (n + 1 + fhm::FH_SCAN_PER_BLOCK - 1) / fhm::FH_SCAN_PER_BLOCK;
        if (nb == 1) {
            hipLaunchKernelGGL(fhm::k_scan_block, dim3(1), dim3(256), 0, st, in, n, out, (uint32_t*)nullptr);
            chk(hipGetLastError());
            return err == hipSuccess;
        }
        uint32_t* sums = (uint32_t*)alloc((size_t)nb * 4);
        uint32_t* sums_off = (uint32_t*)alloc((size_t)(nb + 1) * 4);
        if (!sums || !sums_off) return false;
        hipLaunchKernelGGL(fhm::k_scan_block, dim3(nb), dim3(256), 0, st, in, n, out, sums);
        chk(hipGetLastError());
        if (!scan(sums, nb, sums_off)) return false;
        hipLaunchKernelGGL(fhm::k_scan_add, grid((uint64_t)n + 1), dim3(256), 0, st, out, n, (const uint32_t*)sums_off);
        chk(hipGetLastError());
        free(sums); free(sums_off);
        return err == hipSuccess;
    }
    void count(const fhmesh::WalkTree& o, const fhmesh::WalkItem* items, uint32_t n, uint32_t* cnt, uint32_t* live) {
        hipLaunchKernelGGL(fhm::k_walk_count, grid(n), dim3(256), 0, st, o, items, n, cnt, live);
        chk(hipGetLastError());
    }
    void expand(const fhmesh::WalkTree& o, const fhmesh::WalkItem* items, uint32_t n, const uint32_t* off, fhmesh::WalkItem* next) {
        hipLaunchKernelGGL(fhm::k_walk_expand, grid(n), dim3(256), 0, st, o, items, n, off, next);
        chk(hipGetLastError());
    }
    void first_min(const fhmesh::WalkItem* recs, uint32_t n, uint32_t* first) {
        hipLaunchKernelGGL(fhm::k_walk_first, grid((uint64_t)n * 5), dim3(256), 0, st, recs, n, first);
        chk(hipGetLastError());
    }
    void rec_counts(const fhmesh::WalkItem* recs, uint32_t n, const uint32_t* first, uint32_t* nn, uint32_t* nt) {
        hipLaunchKernelGGL(fhm::k_walk_rec_counts, grid(n), dim3(256), 0, st, recs, n, first, nn, nt);
        chk(hipGetLastError());
    }
    void rec_number(const fhmesh::WalkItem* recs, uint32_t n, uint32_t* first, const uint32_t* vb, const fhmesh::V3* octree_verts, fhmesh::V3* verts) {
        hipLaunchKernelGGL(fhm::k_walk_rec_number, grid(n), dim3(256), 0, st, recs, n, first, vb, octree_verts, verts);
        chk(hipGetLastError());
    }
    void rec_triangles(const fhmesh::WalkItem* recs, uint32_t n, const uint32_t* first, const uint32_t* tb, uint64_t* tris) {
        hipLaunchKernelGGL(fhm::k_walk_rec_triangles, grid(n), dim3(256), 0, st, recs, n, first, tb, tris);
        chk(hipGetLastError());
    }
};
// ... and on the host: the same passes as plain loops (fhip_debug_walk_dual mode 3: how they are tested without a GPU)
struct WalkHostX {
    void* alloc(size_t b) { return malloc(b ? b : 4); }
    void free(void* p) { ::free(p); }
    void zero(void* p, size_t b) { memset(p, 0, b); }
    void fill_ff(void* p, size_t b) { memset(p, 0xFF, b); }
    void read(void* d, const void* s, size_t b) { memcpy(d, s, b); }
    void write(void* d, const void* s, size_t b) { memcpy(d, s, b); }
    bool scan(const uint32_t* in, uint32_t n, uint32_t* out) { uint32_t run = 0; for (uint32_t i = 0; i < n; i++) { out[i] = run; run += in[i]; } out[n] = run; return true; }
    void count(const fhmesh::WalkTree& o, const fhmesh::WalkItem* items, uint32_t n, uint32_t* cnt, uint32_t* live) {
        for (uint32_t i = n; i-- > 0;) { cnt[i] = fhmesh::wk_count(o, items[i]); if ((items[i].hdr & 3u) != fhmesh::WK_REC) *live = 1; }      // (any order)
    }
    void expand(const fhmesh::WalkTree& o, const fhmesh::WalkItem* items, uint32_t n, const uint32_t* off, fhmesh::WalkItem* next) {
        for (uint32_t i = n; i-- > 0;) if (off[i + 1] != off[i]) fhmesh::wk_expand(o, items[i], next + off[i]);
    }
    void first_min(const fhmesh::WalkItem* recs, uint32_t n, uint32_t* first) {
        for (uint64_t i = (uint64_t)n * 5; i-- > 0;) { uint32_t& f = first[recs[i / 5].a[i % 5]]; if ((uint32_t)i < f) f = (uint32_t)i; }
    }
    void rec_counts(const fhmesh::WalkItem* recs, uint32_t n, const uint32_t* first, uint32_t* nn, uint32_t* nt) {
        for (uint32_t i = 0; i < n; i++) fhmesh::wk_rec_counts(recs[i], i, first, &nn[i], &nt[i]);
    }
    void rec_number(const fhmesh::WalkItem* recs, uint32_t n, uint32_t* first, const uint32_t* vb, const fhmesh::V3* octree_verts, fhmesh::V3* verts) {
        for (uint32_t i = n; i-- > 0;) fhmesh::wk_rec_number(recs[i], i, first, vb[i], octree_verts, verts);
    }
    void rec_triangles(const fhmesh::WalkItem* recs, uint32_t n, const uint32_t* first, const uint32_t* tb, uint64_t* tris) {
        for (uint32_t i = n; i-- > 0;) fhmesh::wk_rec_triangles(recs[i], first, tb[i], tris);
    }
};
static const fhmesh::WalkTable& walk_table() {       // CELL_TO_EDGE_TO_VERT out of host_mesh.hpp's tables
    static const fhmesh::WalkTable* const W = [] {
        fhmesh::WalkTable* w = new fhmesh::WalkTable();
        const fhmesh::Tables& T = fhmesh::tables();
        for (int m = 0; m < 256; m++) {
            w->any[m][0] = w->any[m][1] = -1;
            for (int e = 0; e < 12; e++) { w->e2v[m][e][0] = (int8_t)T.e2v[m][e][0]; w->e2v[m][e][1] = (int8_t)T.e2v[m][e][1]; }
            for (int e = 0; e < 12; e++) if (T.e2v[m][e][0] >= 0) { w->any[m][0] = (int8_t)T.e2v[m][e][0]; w->any[m][1] = (int8_t)T.e2v[m][e][1]; break; }
        }
        return w;
    }();
    return *W;
}
static hipError_t mesh_assemble_device(fhip_ctx* ctx, fhip_mesh* M, uint32_t depth, std::vector<fhmesh::OctLevel>& lv, const FhMeshLeaf* rec, uint32_t n_rec, const FhMdcTable* table,
                                       bool has_mat, const float* mat, MeshTimes& T, std::string& why);
enum MeshMode { MESH_SAMPLE, MESH_BUILD, MESH_PART };
static fhip_status mesh_run(fhip_ctx* ctx, const fhip_tape* tape, uint32_t depth, const float* world_to_model, const int32_t* axis_slots,
                            const uint64_t* var_keys, const float* var_values, uint32_t n_vars, MeshMode mode, uint32_t part, uint32_t n_parts, fhip_mesh** out) {
    if (!out) return FHIP_ERR_BAD_TAPE;
    *out = nullptr;
    const bool assemble = mode == MESH_BUILD;
    // fhip_mesh_build assembles the octree on the device: the levels' arrays and the leaf records stay in HBM, the host gets the finished
    // octree for the dual walk.  (Option mesh_device_assembly 0: on the host's threads from copies of both, as fhip_mesh_merge does.)
    const bool dev_asm = assemble && n_parts == 1 && ctx->opt.mesh_device_assembly;
    const bool keep = mode != MESH_SAMPLE && !dev_asm;
    if (depth > 20) return fail(ctx, FHIP_ERR_UNSUPPORTED, "octree depth above 20");
    if (n_parts < 1 || n_parts > 8 || part >= n_parts) return fail(ctx, FHIP_ERR_UNSUPPORTED, "mesh parts: 1..8, part < n_parts");
    const fh::HostTape& t = tape->t;
    if (t.n_outputs != 1) return fail(ctx, FHIP_ERR_BAD_TAPE, "shape tapes have exactly one output");
    (void)hipSetDevice(ctx->device);
    { fhip_status ts_ = tape_to_device(ctx, tape); if (ts_) return ts_; }
    FhRender R;
    memset(&R, 0, sizeof(R));
    fhip_status st = bind_inputs(ctx, tape, axis_slots, var_keys, var_values, n_vars, R);
    if (st) return st;
    FhMeshParams P;
    memset(&P, 0, sizeof(P));
    P.tape = tape->d_ops; P.len = (uint32_t)t.ops.size(); P.n_regs = std::max<uint32_t>(t.n_regs, 1);
    bool ident = true;
    if (world_to_model) for (int i = 0; i < 16; i++) { P.mat[i] = world_to_model[i]; ident &= world_to_model[i] == ((i % 5 == 0) ? 1.0f : 0.0f); }
    P.has_mat = (world_to_model && !ident) ? 1 : 0;     // octree.rs:487-492: no transform at all for the identity
    for (int s = 0; s < FH_MAX_INPUTS; s++) { P.in_kind[s] = R.in_kind[s]; P.in_value[s] = R.in_value[s]; }
    const size_t lds_iv = (size_t)P.n_regs * WAVE * 8, lds_leaf = (size_t)P.n_regs * WAVE * 16;
    if (lds_leaf + 1024 > FH_LDS_MAX) return fail(ctx, FHIP_ERR_UNSUPPORTED, "register file exceeds LDS");
    {   // (function attributes are per device; contexts on several host threads may arrive here together)
        static std::mutex attr_lock;
        static bool attr_done[64] = {};
        std::lock_guard<std::mutex> guard(attr_lock);
        const int d = ctx->device & 63;
        if (!attr_done[d]) {
            (void)hipFuncSetAttribute((const void*)fhm::k_mesh_cells, hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS_MAX);
            (void)hipFuncSetAttribute((const void*)fhm::k_mesh_choices, hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS_MAX);
            (void)hipFuncSetAttribute((const void*)fhm::k_mesh_leaf, hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS_MAX - 2048);
            (void)hipFuncSetAttribute((const void*)fhm::k_mesh_corners, hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS_MAX);
            (void)hipFuncSetAttribute((const void*)fhm::k_mesh_edges, hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS_MAX);
            (void)hipFuncSetAttribute((const void*)fhm::k_mesh_grads, hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS_MAX);
            attr_done[d] = true;
        }
    }
    fhip_mesh* M = new fhip_mesh();
    M->depth = depth; M->part = part; M->n_parts = n_parts;
    const bool times = getenv("FHIP_MESH_TIMES") != nullptr;       // diagnostic: phase wall times on stderr
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = now();
    double t_cells = 0, t_leaf = 0, t_copy = 0;
    DevBuf& leaves = ctx->mesh_leaves;      // (kept with the context between builds)
    DevBuf bufs[2], counters, table, d_cls, d_slot, edge_list, edge_count, edge_br, edge_vars, edge_vals, sub_ops, sub_tab, sub_choices, sub_ops2, sub_tab2;
    std::vector<DevBuf> lv_cls, lv_slot, lv_amb;        // dev_asm: every level's classes, slots and ambiguous cells stay
    if (dev_asm) { lv_cls.resize(depth + 1); lv_slot.resize(depth + 1); lv_amb.resize(depth + 1); }
    std::vector<uint32_t> lv_n_amb;
    auto cleanup = [&] {
        bufs[0].release(); bufs[1].release(); counters.release(); table.release(); d_cls.release(); d_slot.release();
        edge_list.release(); edge_count.release(); edge_br.release(); edge_vars.release(); edge_vals.release();
        sub_ops.release(); sub_tab.release(); sub_choices.release(); sub_ops2.release(); sub_tab2.release();
        for (auto* v : {&lv_cls, &lv_slot, &lv_amb}) for (DevBuf& b : *v) b.release();
    };
#define MESH_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { cleanup(); delete M; return fail(ctx, FHIP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } } while (0)
    MESH_TRY(counters.ensure(16));
    FhMeshCell root;
    for (int k = 0; k < 3; k++) { root.b[2 * k] = -1.0f; root.b[2 * k + 1] = 1.0f; }     // CellBounds::new (cell.rs:171-176)
    root.path = 1;
    MESH_TRY(bufs[0].ensure(sizeof(FhMeshCell)));
    MESH_TRY(hipMemcpyAsync(bufs[0].p, &root, sizeof(root), hipMemcpyHostToDevice, ctx->stream));
    // (option mesh_simplify_min_ops, default 256: shorter tapes are evaluated as they are - gyroid-sphere's 28 ops gain nothing and keep the
    // assembly bulk interpreter for their leaf samples; 0 = never.  The level: 4 - 4 096 cells at most, a 16th of the region across, where
    // prospero.vm's 6 363 ops are down to a few hundred - or two above the leaves of a shallower octree)
    const uint32_t split_level = (ctx->opt.mesh_simplify_min_ops > 0 && t.ops.size() >= (size_t)ctx->opt.mesh_simplify_min_ops && t.n_choices > 0 && depth >= 3)
                                     ? std::min<uint32_t>(4, depth - 2) : 0;
    // ... and once more at depth - 2 (at most level 7: a table of 8^7 entries), from the first split's tapes, when the first one was taken:
    // prospero.vm's level-4 tapes still hold ~600 ops, and the leaf samples of a depth-8 build walked them 16 M times
    const uint32_t split_level2 = (split_level == 4 && depth >= 7) ? std::min<uint32_t>(7, depth - 2) : 0;
    std::vector<fh::HostTape> sub_keep;           // the first split's tapes, by table index (kept for the second)
    std::vector<int32_t> sub_of;                  // first-split table index -> index into sub_keep, -1: the root tape
    uint32_t n_in = 1;      // cells in bufs[cur] to evaluate (level 0) or whose 8 children to evaluate
    int cur = 0;
    uint32_t n_leaf_cells = 0;
    for (uint32_t d = 0; d <= depth; d++) {
        const uint64_t n64 = d == 0 ? 1 : (uint64_t)n_in * 8;
        if (n64 > (1ull << 30)) { cleanup(); delete M; return fail(ctx, FHIP_ERR_OVERFLOW, "octree level above 2^30 cells"); }
        const uint32_t n = (uint32_t)n64;
        DevBuf& out_cells = dev_asm ? lv_amb[d] : bufs[cur ^ 1];
        const void* in_cells = (dev_asm && d > 0) ? lv_amb[d - 1].p : bufs[cur].p;
        MESH_TRY(out_cells.ensure((size_t)n * sizeof(FhMeshCell)));
        MESH_TRY(hipMemsetAsync(counters.p, 0, 16, ctx->stream));
        if (keep) { MESH_TRY(d_cls.ensure(n)); MESH_TRY(d_slot.ensure((size_t)n * 4)); }
        if (dev_asm) { MESH_TRY(lv_cls[d].ensure(n)); MESH_TRY(lv_slot[d].ensure((size_t)n * 4)); }
        uint8_t* const cls_p = dev_asm ? (uint8_t*)lv_cls[d].p : (keep ? (uint8_t*)d_cls.p : nullptr);
        uint32_t* const slot_p = dev_asm ? (uint32_t*)lv_slot[d].p : (keep ? (uint32_t*)d_slot.p : nullptr);
        const uint32_t child_mask = (d == 1 && n_parts > 1) ? mesh_part_mask(part, n_parts) : 0xFFu;      // (level 1 = the root's 8 children)
        hipLaunchKernelGGL(fhm::k_mesh_cells, dim3((n + WAVE - 1) / WAVE), dim3(WAVE), lds_iv, ctx->stream, P, (const FhMeshCell*)in_cells, n, d == 0 ? 0 : 1,
                           (FhMeshCell*)out_cells.p, (uint32_t*)counters.p, n, cls_p, slot_p, child_mask);
        MESH_TRY(hipGetLastError());
        uint32_t c[4];
        MESH_TRY(hipMemcpyAsync(c, counters.p, 16, hipMemcpyDeviceToHost, ctx->stream));
        if (keep) {
            M->cls.emplace_back(n); M->slot.emplace_back(n);
            MESH_TRY(hipMemcpyAsync(M->cls.back().data(), d_cls.p, n, hipMemcpyDeviceToHost, ctx->stream));
            MESH_TRY(hipMemcpyAsync(M->slot.back().data(), d_slot.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        }
        MESH_TRY(hipStreamSynchronize(ctx->stream));
        const uint32_t n_here = child_mask == 0xFFu ? n : (uint32_t)__builtin_popcount(child_mask);
        M->cells_evaluated += n_here; M->full += c[1]; M->empty += c[2];
        M->per_level.push_back(n_here);
        cur ^= 1;
        n_in = c[0];
        lv_n_amb.push_back(c[0]);
        if (d == depth) n_leaf_cells = c[0];
        if (n_in == 0) break;
        if (d == split_level && split_level > 0) {
            // Tape simplification down the octree (octree.rs:546-553), once: the choices of the root tape over every ambiguous cell of this
            // level (k_mesh_choices), VmData::simplify under them on the host's threads, the simplified tapes back as one array with a
            // table indexed by the cell's path; every launch from here on gives a lane the tape of its cell's ancestor at this level.
            const uint32_t na = c[0], nch = t.n_choices;
            MESH_TRY(sub_choices.ensure((size_t)na * nch));
            hipLaunchKernelGGL(fhm::k_mesh_choices, dim3((na + WAVE - 1) / WAVE), dim3(WAVE), lds_iv, ctx->stream, P, (const FhMeshCell*)out_cells.p, na, nch, (uint8_t*)sub_choices.p);
            MESH_TRY(hipGetLastError());
            std::vector<uint8_t> ch((size_t)na * nch);
            std::vector<FhMeshCell> amb(na);
            MESH_TRY(hipMemcpyAsync(ch.data(), sub_choices.p, ch.size(), hipMemcpyDeviceToHost, ctx->stream));
            MESH_TRY(hipMemcpyAsync(amb.data(), out_cells.p, (size_t)na * sizeof(FhMeshCell), hipMemcpyDeviceToHost, ctx->stream));
            MESH_TRY(hipStreamSynchronize(ctx->stream));
            std::vector<fh::HostTape> sub(na);
            std::vector<uint8_t> ok(na, 0);
            fhmesh::parallel_for(na, [&](size_t j) { ok[j] = simplify_host(t, ch.data() + j * nch, sub[j]) ? 1 : 0; });
            const size_t n_tab = (size_t)1 << (3 * split_level);
            std::vector<uint2> tab(n_tab, make_uint2(0, 0));
            std::vector<uint64_t> ops;
            sub_of.assign(n_tab, -1);
            for (uint32_t j = 0; j < na; j++) {
                if (!ok[j] || sub[j].ops.empty() || sub[j].ops.size() >= t.ops.size()) continue;     // (nothing gained: the root tape)
                const uint64_t idx = amb[j].path - ((uint64_t)1 << (3 * split_level));
                if (idx >= n_tab) continue;
                tab[(size_t)idx] = make_uint2((uint32_t)ops.size(), (uint32_t)sub[j].ops.size());
                ops.insert(ops.end(), sub[j].ops.begin(), sub[j].ops.end());
                M->sub_tapes++; M->sub_ops += sub[j].ops.size();
                if (split_level2) { sub_of[(size_t)idx] = (int32_t)sub_keep.size(); sub_keep.push_back(std::move(sub[j])); }
            }
            // ... where it pays: the lanes of a wave then walk tapes of their own through the generic interpreter, and a tape that fits the
            // assembly bulk interpreter (<= 32 registers) gives that up for its leaf samples - bear.vm's smooth blend keeps 3/4 of its ops
            // at this level and meshes twice as fast WITHOUT (measured, profiles/r04g); prospero.vm keeps 1/20 and gains 16x
            const double kept = M->sub_tapes ? (double)M->sub_ops / ((double)M->sub_tapes * (double)t.ops.size()) : 1.0;
            const bool bulk_capable = ctx->use_asm && P.n_regs <= 32;
            if (kept >= (bulk_capable ? 0.25 : 0.75)) { ops.clear(); M->sub_skipped = M->sub_tapes; M->sub_tapes = 0; sub_keep.clear(); }
            if (!ops.empty()) {
                MESH_TRY(sub_ops.ensure(ops.size() * 8));
                MESH_TRY(sub_tab.ensure(n_tab * sizeof(uint2)));
                MESH_TRY(hipMemcpyAsync(sub_ops.p, ops.data(), ops.size() * 8, hipMemcpyHostToDevice, ctx->stream));
                MESH_TRY(hipMemcpyAsync(sub_tab.p, tab.data(), n_tab * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream));
                MESH_TRY(hipStreamSynchronize(ctx->stream));
                P.sub_ops = (const uint64_t*)sub_ops.p; P.sub_tab = (const uint2*)sub_tab.p; P.split_level = split_level;
            }
        }
        // (only where the first split left tapes worth pruning again: 128 ops on average - colonnade.vm's are ~50 and a second split cost it 20 %)
        if (d == split_level2 && split_level2 > 0 && P.sub_tab && !sub_keep.empty() && M->sub_ops >= 128 * M->sub_tapes) {
            // The second split: the choices of every ambiguous cell of this level over the tape it inherited (its level-4 ancestor's),
            // VmData::simplify of THAT tape under them, a second table for everything below.
            const uint32_t na = c[0];
            uint32_t nch = 1;
            for (const fh::HostTape& st : sub_keep) nch = std::max(nch, st.n_choices);
            if ((size_t)na * nch <= ((size_t)3 << 30)) {
                MESH_TRY(sub_choices.ensure((size_t)na * nch));
                hipLaunchKernelGGL(fhm::k_mesh_choices, dim3((na + WAVE - 1) / WAVE), dim3(WAVE), lds_iv, ctx->stream, P, (const FhMeshCell*)out_cells.p, na, nch, (uint8_t*)sub_choices.p);
                MESH_TRY(hipGetLastError());
                std::vector<uint8_t> ch((size_t)na * nch);
                std::vector<FhMeshCell> amb(na);
                MESH_TRY(hipMemcpyAsync(ch.data(), sub_choices.p, ch.size(), hipMemcpyDeviceToHost, ctx->stream));
                MESH_TRY(hipMemcpyAsync(amb.data(), out_cells.p, (size_t)na * sizeof(FhMeshCell), hipMemcpyDeviceToHost, ctx->stream));
                MESH_TRY(hipStreamSynchronize(ctx->stream));
                std::vector<fh::HostTape> sub2(na);
                std::vector<uint8_t> ok(na, 0);
                const int up = 3 * (int)(split_level2 - split_level);
                const uint64_t base1 = (uint64_t)1 << (3 * split_level);
                fhmesh::parallel_for(na, [&](size_t j) {
                    const uint64_t i1 = (amb[j].path >> up) - base1;
                    const int32_t k1 = i1 < sub_of.size() ? sub_of[(size_t)i1] : -1;
                    if (k1 < 0) return;          // (its ancestor kept the root tape: so does it)
                    const fh::HostTape& parent = sub_keep[(size_t)k1];
                    ok[j] = simplify_host(parent, ch.data() + j * nch, sub2[j]) && !sub2[j].ops.empty() && sub2[j].ops.size() < parent.ops.size() ? 1 : 0;
                });
                const size_t n_tab2 = (size_t)1 << (3 * split_level2);
                std::vector<uint2> tab2(n_tab2, make_uint2(0, 0));
                std::vector<uint64_t> ops2;
                uint64_t n2 = 0;
                for (uint32_t j = 0; j < na; j++) {
                    if (!ok[j]) continue;
                    const uint64_t idx = amb[j].path - ((uint64_t)1 << (3 * split_level2));
                    if (idx >= n_tab2 || ops2.size() + sub2[j].ops.size() >= ((size_t)1 << 32)) continue;
                    tab2[(size_t)idx] = make_uint2((uint32_t)ops2.size(), (uint32_t)sub2[j].ops.size());
                    ops2.insert(ops2.end(), sub2[j].ops.begin(), sub2[j].ops.end());
                    n2++;
                }
                if (!ops2.empty()) {
                    MESH_TRY(sub_ops2.ensure(ops2.size() * 8));
                    MESH_TRY(sub_tab2.ensure(n_tab2 * sizeof(uint2)));
                    MESH_TRY(hipMemcpyAsync(sub_ops2.p, ops2.data(), ops2.size() * 8, hipMemcpyHostToDevice, ctx->stream));
                    MESH_TRY(hipMemcpyAsync(sub_tab2.p, tab2.data(), n_tab2 * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream));
                    MESH_TRY(hipStreamSynchronize(ctx->stream));
                    P.sub_ops2 = (const uint64_t*)sub_ops2.p; P.sub_tab2 = (const uint2*)sub_tab2.p; P.split_level2 = split_level2;
                    if (times) fprintf(stderr, "fhip mesh: tape simplified again at level %u: %llu cells with tapes of their own, %.1f ops on average\n", split_level2,
                                       (unsigned long long)n2, (double)ops2.size() / (double)n2);
                }
            }
        }
    }
    M->ambiguous_leaves = n_leaf_cells;
    t_cells = now() - t_start;
    if (times && (M->sub_tapes || M->sub_skipped))
        fprintf(stderr, "fhip mesh: tape simplified at level %u: %llu cells with tapes of their own, %.1f ops on average (root tape: %zu)%s\n", split_level,
                (unsigned long long)(M->sub_tapes + M->sub_skipped), (double)M->sub_ops / (double)(M->sub_tapes + M->sub_skipped), t.ops.size(),
                M->sub_skipped ? " - not used: too little gained" : "");
    FhMdcTable mdc;
    if (n_leaf_cells || dev_asm) {
        build_mdc_table(mdc);
        MESH_TRY(table.ensure(sizeof(mdc)));
        MESH_TRY(hipMemcpyAsync(table.p, &mdc, sizeof(mdc), hipMemcpyHostToDevice, ctx->stream));
    }
    // one chunk of leaf cells sampled into records: as passes in which every lane has a point of its own (corners, the edge search over the
    // chunk's list of edges, gradients), or - FHIP_MESH_LEAF_PASSES=0 - one wavefront per cell (k_mesh_leaf); then the cell vertices' QEFs
    const uint32_t LEAF_CH = 1u << 19;
    const char* const lp_env = getenv("FHIP_MESH_LEAF_PASSES");        // diagnostic: 0 = k_mesh_leaf, the kernel the passes are checked against
    const bool leaf_passes = !(lp_env && lp_env[0] == '0');
    const char* const be_env = getenv("FHIP_MESH_BULK_EDGES");          // diagnostic: 0 = the edge search by k_mesh_edges (the generic interpreter)
    const bool bulk_edges = leaf_passes && ctx->use_asm && P.n_regs <= 32 && !(be_env && be_env[0] == '0') && !P.sub_tab;     // (one tape per launch)
    uint32_t n_slots = std::max<uint32_t>(t.n_vars, 1);
    for (uint32_t sl = 0; sl < FH_MAX_INPUTS; sl++) if (P.in_kind[sl] < 3) n_slots = std::max(n_slots, sl + 1);
    const size_t lds_f32 = (size_t)P.n_regs * WAVE * 4;
    auto sample_chunk = [&](const FhMeshCell* cells, FhMeshLeaf* recs, uint32_t cnt) -> hipError_t {
        hipError_t e = hipSuccess;
        auto ck = [&](hipError_t x) { if (x != hipSuccess && e == hipSuccess) e = x; };
        if (leaf_passes && cnt < (1u << 28)) {
            ck(edge_list.ensure((size_t)LEAF_CH * 12 * 4));
            ck(edge_count.ensure(4));
            if (e != hipSuccess) return e;
            ck(hipMemsetAsync(edge_count.p, 0, 4, ctx->stream));
            if (bulk_edges) {      // the corners through the assembly bulk interpreter too (k_mesh_corners: 48 of a 350 ms build through the generic one)
                const uint32_t np = cnt * 8u;
                ck(edge_vars.ensure((size_t)n_slots * np * 4));
                ck(edge_vals.ensure((size_t)np * 4));
                if (e != hipSuccess) return e;
                for (uint32_t sl = 0; sl < n_slots; sl++)
                    if (P.in_kind[sl] >= 3) {
                        hipLaunchKernelGGL(fhm::k_mesh_fill, dim3((np + 255) / 256), dim3(256), 0, ctx->stream, (float*)edge_vars.p + (size_t)sl * np, P.in_value[sl], np);
                        ck(hipGetLastError());
                    }
                hipLaunchKernelGGL(fhm::k_mesh_corner_points, dim3((np + 255) / 256), dim3(256), 0, ctx->stream, P, cells, cnt, (float*)edge_vars.p);
                ck(hipGetLastError());
                struct { const uint64_t* tape; const float* vars; float* out; uint32_t len, n; } kc = {tape->d_ops, (const float*)edge_vars.p, (float*)edge_vals.p, P.len, np};
                const bool plain_c = tape_asm_ok(t);
                const uint32_t per_c = P.n_regs <= 16 ? 256 : 128;
                const int which_c = P.n_regs <= 16 ? (plain_c ? FH_ASM_FLOAT_16x4 : FH_ASM_FLOAT_16x4_T) : (plain_c ? FH_ASM_FLOAT_32x2 : FH_ASM_FLOAT_32x2_T);
                ck(launch_asm(ctx, which_c, (np + per_c - 1) / per_c, &kc, sizeof(kc)));
                hipLaunchKernelGGL(fhm::k_mesh_corner_masks, dim3((cnt + 7) / 8), dim3(WAVE), 0, ctx->stream, cells, cnt, (const float*)edge_vals.p, (const FhMdcTable*)table.p, recs,
                                   (uint32_t*)edge_count.p, (uint32_t*)edge_list.p);
                ck(hipGetLastError());
            } else {
                hipLaunchKernelGGL(fhm::k_mesh_corners, dim3((cnt + 7) / 8), dim3(WAVE), lds_f32, ctx->stream, P, cells, cnt, (const FhMdcTable*)table.p, recs,
                                   (uint32_t*)edge_count.p, (uint32_t*)edge_list.p);
                ck(hipGetLastError());
            }
            uint32_t n_edges = 0;
            ck(hipMemcpyAsync(&n_edges, edge_count.p, 4, hipMemcpyDeviceToHost, ctx->stream));
            ck(hipStreamSynchronize(ctx->stream));
            if (e == hipSuccess && n_edges && bulk_edges) {
                // the four rounds as passes over the chunk's edges, the samples' values from the assembly bulk interpreter (mesh_edges.hpp)
                const uint32_t n = n_edges * 16u;
                ck(edge_br.ensure((size_t)n_edges * sizeof(fhmesh::EdgeBracket)));
                ck(edge_vars.ensure((size_t)n_slots * n * 4));
                ck(edge_vals.ensure((size_t)n * 4));
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(fhm::k_mesh_edge_begin, dim3((n_edges + 255) / 256), dim3(256), 0, ctx->stream, (const FhMdcTable*)table.p, (const FhMeshLeaf*)recs,
                                   (const uint32_t*)edge_list.p, n_edges, (fhmesh::EdgeBracket*)edge_br.p);
                ck(hipGetLastError());
                for (uint32_t sl = 0; sl < n_slots; sl++)
                    if (P.in_kind[sl] >= 3) {
                        hipLaunchKernelGGL(fhm::k_mesh_fill, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (float*)edge_vars.p + (size_t)sl * n, P.in_value[sl], n);
                        ck(hipGetLastError());
                    }
                struct { const uint64_t* tape; const float* vars; float* out; uint32_t len, n; } ka = {tape->d_ops, (const float*)edge_vars.p, (float*)edge_vals.p, P.len, n};
                const bool plain = tape_asm_ok(t);
                const uint32_t per = P.n_regs <= 16 ? 256 : 128;
                const int which = P.n_regs <= 16 ? (plain ? FH_ASM_FLOAT_16x4 : FH_ASM_FLOAT_16x4_T) : (plain ? FH_ASM_FLOAT_32x2 : FH_ASM_FLOAT_32x2_T);
                for (int round = 0; round < 4 && e == hipSuccess; round++) {
                    hipLaunchKernelGGL(fhm::k_mesh_edge_points, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, P, (const FhMeshLeaf*)recs, (const uint32_t*)edge_list.p,
                                       (const fhmesh::EdgeBracket*)edge_br.p, n_edges, (float*)edge_vars.p, n);
                    ck(hipGetLastError());
                    ck(launch_asm(ctx, which, (n + per - 1) / per, &ka, sizeof(ka)));
                    hipLaunchKernelGGL(fhm::k_mesh_edge_narrow, dim3((n_edges + 255) / 256), dim3(256), 0, ctx->stream, (fhmesh::EdgeBracket*)edge_br.p, (const float*)edge_vals.p, n_edges);
                    ck(hipGetLastError());
                }
                hipLaunchKernelGGL(fhm::k_mesh_edge_end, dim3((n_edges + 255) / 256), dim3(256), 0, ctx->stream, recs, (const uint32_t*)edge_list.p, (const fhmesh::EdgeBracket*)edge_br.p, n_edges);
                ck(hipGetLastError());
                hipLaunchKernelGGL(fhm::k_mesh_grads, dim3((n_edges + WAVE - 1) / WAVE), dim3(WAVE), lds_leaf, ctx->stream, P, recs, (const uint32_t*)edge_list.p, n_edges);
                ck(hipGetLastError());
            } else if (e == hipSuccess && n_edges) {
                hipLaunchKernelGGL(fhm::k_mesh_edges, dim3((n_edges + 3) / 4), dim3(WAVE), lds_f32, ctx->stream, P, (const FhMdcTable*)table.p, recs, (const uint32_t*)edge_list.p, n_edges);
                ck(hipGetLastError());
                hipLaunchKernelGGL(fhm::k_mesh_grads, dim3((n_edges + WAVE - 1) / WAVE), dim3(WAVE), lds_leaf, ctx->stream, P, recs, (const uint32_t*)edge_list.p, n_edges);
                ck(hipGetLastError());
            }
        } else {
            hipLaunchKernelGGL(fhm::k_mesh_leaf, dim3(cnt), dim3(WAVE), lds_leaf, ctx->stream, P, cells, cnt, (const FhMdcTable*)table.p, recs);
            ck(hipGetLastError());
        }
        hipLaunchKernelGGL(fhm::k_mesh_leaf_qef, dim3((cnt + WAVE - 1) / WAVE), dim3(WAVE), 0, ctx->stream, (const FhMdcTable*)table.p, recs, cnt);
        ck(hipGetLastError());
        return e;
    };
    if (n_leaf_cells && dev_asm) {      // the records stay in HBM
        MESH_TRY(leaves.ensure((size_t)n_leaf_cells * sizeof(FhMeshLeaf)));
        const uint32_t CH = LEAF_CH;
        const void* const leaf_cells = lv_amb[depth].p;
        for (uint32_t off = 0; off < n_leaf_cells; off += CH) {
            const uint32_t cnt = std::min<uint32_t>(CH, n_leaf_cells - off);
            const hipError_t se = sample_chunk((const FhMeshCell*)leaf_cells + off, (FhMeshLeaf*)leaves.p + off, cnt);
            MESH_TRY(se);
        }
        if (times) { MESH_TRY(hipStreamSynchronize(ctx->stream)); t_leaf = now() - t_start - t_cells; }
    } else if (n_leaf_cells) {
        MESH_TRY(leaves.ensure((size_t)n_leaf_cells * sizeof(FhMeshLeaf)));
        // in chunks: the records of chunk k travel to the host (second stream) while chunk k + 1 is sampled
        const size_t leaf_bytes = (size_t)n_leaf_cells * sizeof(FhMeshLeaf);
        if (assemble) {     // the records are only needed until the octree is assembled: the context's cached landing area
            if (ctx->mesh_pinned_cap < leaf_bytes) {
                if (ctx->mesh_pinned) (void)hipHostFree(ctx->mesh_pinned);
                ctx->mesh_pinned = nullptr; ctx->mesh_pinned_cap = 0;
                MESH_TRY(hipHostMalloc(&ctx->mesh_pinned, leaf_bytes + leaf_bytes / 8, hipHostMallocDefault));
                ctx->mesh_pinned_cap = leaf_bytes + leaf_bytes / 8;
            }
            M->leaves.p = (FhMeshLeaf*)ctx->mesh_pinned;
            M->leaves.borrowed = true;
        } else
            MESH_TRY(hipHostMalloc((void**)&M->leaves.p, leaf_bytes, hipHostMallocDefault));
        M->leaves.n = n_leaf_cells;
        const uint32_t CH = LEAF_CH;
        std::vector<hipEvent_t> evs;
        hipStream_t const copy_stream = ctx->stream2 ? ctx->stream2 : ctx->stream;
        bool ok = true;
        hipError_t first_err = hipSuccess;
        auto chk = [&](hipError_t e) { if (e != hipSuccess && ok) { ok = false; first_err = e; } };
        for (uint32_t off = 0; off < n_leaf_cells && ok; off += CH) {
            const uint32_t cnt = std::min<uint32_t>(CH, n_leaf_cells - off);
            chk(sample_chunk((const FhMeshCell*)bufs[cur].p + off, (FhMeshLeaf*)leaves.p + off, cnt));
            hipEvent_t ev = nullptr;
            chk(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            if (ev) evs.push_back(ev);
            chk(hipEventRecord(ev, ctx->stream));
            chk(hipStreamWaitEvent(copy_stream, ev, 0));
            chk(hipMemcpyAsync(M->leaves.p + off, (FhMeshLeaf*)leaves.p + off, (size_t)cnt * sizeof(FhMeshLeaf), hipMemcpyDeviceToHost, copy_stream));
        }
        if (times) { chk(hipStreamSynchronize(ctx->stream)); t_leaf = now() - t_start - t_cells; }
        chk(hipStreamSynchronize(ctx->stream));
        chk(hipStreamSynchronize(copy_stream));
        for (hipEvent_t e : evs) (void)hipEventDestroy(e);
        MESH_TRY(first_err);
    }
    if (dev_asm) {
        std::vector<fhmesh::OctLevel> lv(lv_n_amb.size());
        for (size_t d = 0; d < lv.size(); d++) {
            lv[d].cls = (const uint8_t*)lv_cls[d].p; lv[d].slot = (const uint32_t*)lv_slot[d].p;
            lv[d].amb = (const FhMeshCell*)lv_amb[d].p; lv[d].n_amb = lv_n_amb[d];
        }
        MeshTimes MT{times, t_start, t_cells, t_leaf, 0.0, n_leaf_cells};
        std::string why;
        const hipError_t ae = mesh_assemble_device(ctx, M, depth, lv, (const FhMeshLeaf*)leaves.p, n_leaf_cells, (const FhMdcTable*)table.p, P.has_mat != 0, P.mat, MT, why);
        if (ae != hipSuccess && !why.empty()) { cleanup(); delete M; return fail(ctx, FHIP_ERR_OVERFLOW, why); }
        MESH_TRY(ae);
        cleanup();
        *out = M;
        return FHIP_OK;
    }
#undef MESH_TRY
    cleanup();
    t_copy = now() - t_start - t_cells - t_leaf;
    MeshTimes MT{times, t_start, t_cells, t_leaf, t_copy, n_leaf_cells};
    if (assemble) mesh_assemble(ctx, M, depth, P.has_mat != 0, P.mat, MT);
    else if (times)
        fprintf(stderr, "fhip mesh depth %u (part %u of %u): cells %.4f s (%llu evaluated), leaf kernel %.4f s (%u leaves), copies %.4f s\n", depth, part, n_parts,
                t_cells, (unsigned long long)M->cells_evaluated, t_leaf, n_leaf_cells, t_copy);
    *out = M;
    return FHIP_OK;
}
// Octree assembly (cell collapse included) and dual walk on the host's threads, from the classes / slots / leaf records in M
static void mesh_cache_release(void* octree) { delete (fhmesh::Octree*)octree; }
static void mesh_assemble(fhip_ctx* ctx, fhip_mesh* M, uint32_t depth, bool has_mat, const float* mat, MeshTimes& T) {
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    double t_asm = 0, t_walk = 0;
    {
        const float rb[6] = {-1.0f, 1.0f, -1.0f, 1.0f, -1.0f, 1.0f};
        fhmesh::Hermite h;
        // (a level the recursion never reached - everything above it was decided - has no arrays: only levels 0 .. cls.size()-1 are indexed)
        const uint32_t split = std::min<uint32_t>(depth, getenv("FHIP_MESH_SPLIT") ? (uint32_t)atoi(getenv("FHIP_MESH_SPLIT")) : 5u);
        const bool par = split >= 1 && M->cls.size() > split && fhmesh::mesh_threads() > 1;
        struct { fhmesh::Octree o; } A;
        if (par) {
            ParallelMeshAssembler PA{*M, depth, split, {}, {}, {}, 0};
            if (ctx && ctx->mesh_octree_cache) {       // the arrays of the last build: their room, not their contents
                PA.o = std::move(*(fhmesh::Octree*)ctx->mesh_octree_cache);
                PA.o.cells.clear(); PA.o.verts.clear(); PA.o.root = fhmesh::Cell();
            }
            PA.o.root = PA.run(rb, &h);
            A.o = std::move(PA.o);
        } else {
            MeshAssembler SA{*M, depth, {}};
            SA.o.root = SA.build(0, 0, rb, &h);
            A.o = std::move(SA.o);
        }
        if (has_mat)       // octree.rs:58-65: vertices back to model space (nalgebra transform_point)
            for (auto& v : A.o.verts) {
                const float x = v.x, y = v.y, z = v.z;
                const float n = ((mat[12] * x + mat[13] * y) + mat[14] * z) + mat[15];
                float a = ((mat[0] * x + mat[1] * y) + mat[2] * z) + mat[3];
                float b = ((mat[4] * x + mat[5] * y) + mat[6] * z) + mat[7];
                float c = ((mat[8] * x + mat[9] * y) + mat[10] * z) + mat[11];
                if (n != 0.0f) { a = a / n; b = b / n; c = c / n; }
                v.x = a; v.y = b; v.z = c;
            }
        t_asm = now() - t0;
        M->leaves.p = nullptr; M->leaves.n = 0;      // (borrowed from the context or from the parts' buffers: gone with the assembly)
        M->leaves.seg_p.clear(); M->leaves.seg_start.clear();
        fhmesh::ParallelWalker W(A.o);
        if (ctx) { W.scratch = &ctx->mesh_first; W.scratch_cap = &ctx->mesh_first_cap; }
        W.run();
        t_walk = now() - t0 - t_asm;
        M->octree_cells = A.o.cells.size(); M->octree_verts = A.o.verts.size();
        M->vertices.swap(W.vertices);
        M->triangles.swap(W.triangles);
        if (ctx && par) {
            if (!ctx->mesh_octree_cache) ctx->mesh_octree_cache = new fhmesh::Octree();
            *(fhmesh::Octree*)ctx->mesh_octree_cache = std::move(A.o);
        }
    }
    if (T.on)
        fprintf(stderr, "fhip mesh depth %u: cells %.4f s (%llu evaluated), leaf kernel %.4f s (%u leaves), copies %.4f s, assembly %.4f s, dual walk %.4f s, total %.4f s\n",
                depth, T.t_cells, (unsigned long long)M->cells_evaluated, T.t_leaf, T.n_leaf_cells, T.t_copy, t_asm, t_walk, now() - T.t_start);
}
// fhip_mesh_build's second half: the octree assembled in HBM (check_done / collapse / places, mesh_collapse.hpp), then Octree::walk_dual on the
// device too (mesh_walk.hpp; the finished mesh travels to the host through the context's pinned landing area) - or, option mesh_device_walk 0
// and for octrees beyond the device walk's 32-bit numbers, its blocks of cells copied to the host, the walk on the host's threads over them,
// and the mesh's vertices - the walk knows which of the octree's they are - gathered on the device.  Neither the leaf records (528 bytes each) nor the octree's vertices (at depth 10:
// 191 M, of which the mesh uses 7.5 M) leave the device.
static hipError_t mesh_assemble_device(fhip_ctx* ctx, fhip_mesh* M, uint32_t depth, std::vector<fhmesh::OctLevel>& lv, const FhMeshLeaf* rec, uint32_t n_rec, const FhMdcTable* table,
                                       bool has_mat, const float* mat, MeshTimes& T, std::string& why) {
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    OctDevX x{ctx->stream, {}, hipSuccess};
    auto give_up = [&](hipError_t e) { (void)hipStreamSynchronize(ctx->stream); x.release(); return e; };
    float* d_mat = nullptr;
    if (has_mat) {
        d_mat = (float*)x.alloc(64);
        if (d_mat) x.chk(hipMemcpyAsync(d_mat, mat, 64, hipMemcpyHostToDevice, ctx->stream));
    }
    fhmesh::OctOut oo;
    const int rc = x.err != hipSuccess ? (int)fhmesh::OCT_NO_MEMORY : fhmesh::oct_assemble(x, depth, lv.data(), (uint32_t)lv.size(), rec, n_rec, table, d_mat, &oo);
    if (rc == fhmesh::OCT_TOO_MANY_VERTICES) { why = "the octree has more than 2^32 vertices"; return give_up(hipErrorInvalidValue); }
    if (rc != fhmesh::OCT_OK || x.err != hipSuccess) return give_up(x.err != hipSuccess ? x.err : hipErrorOutOfMemory);
    // Octree::walk_dual on the device too (mesh_walk.hpp; option mesh_device_walk, on): cells and octree vertices are read where the assembly
    // left them, the mesh's vertices and triangles come to the host through the context's pinned landing area.  Octrees beyond its limits
    // (2^24 blocks, 2^31 / 5 quads) and any failure fall back to the host's threads below.
    if (ctx->opt.mesh_device_walk) {
        WalkDevX w{ctx->stream, {}, hipSuccess};
        fhmesh::WalkTable* d_tab = (fhmesh::WalkTable*)w.alloc(sizeof(fhmesh::WalkTable));
        fhmesh::WalkOut wo;
        int wrc = fhmesh::WALK_NO_MEMORY;
        if (d_tab) {
            w.write(d_tab, &walk_table(), sizeof(fhmesh::WalkTable));
            wrc = fhmesh::walk_dual_passes(w, oo.cells, oo.n_blocks, oo.root, oo.verts, oo.n_verts, d_tab, &wo);
        }
        if (wrc == fhmesh::WALK_OK && w.err == hipSuccess) {
            const double t_asm = now() - t0;
            const size_t vbytes = (size_t)wo.n_verts * sizeof(fhmesh::V3), tbytes = (size_t)wo.n_tris * 24, need = ((vbytes + 255) & ~(size_t)255) + tbytes + 256;
            if (ctx->mesh_pinned_cap < need) {
                if (ctx->mesh_pinned) (void)hipHostFree(ctx->mesh_pinned);
                ctx->mesh_pinned = nullptr; ctx->mesh_pinned_cap = 0;
                if (hipHostMalloc(&ctx->mesh_pinned, need + need / 8, hipHostMallocDefault) == hipSuccess) ctx->mesh_pinned_cap = need + need / 8;
            }
            if (ctx->mesh_pinned_cap >= need) {
                char* pv = (char*)ctx->mesh_pinned;
                char* pt = pv + ((vbytes + 255) & ~(size_t)255);
                if (vbytes) w.chk(hipMemcpyAsync(pv, wo.verts, vbytes, hipMemcpyDeviceToHost, ctx->stream));
                if (tbytes) w.chk(hipMemcpyAsync(pt, wo.tris, tbytes, hipMemcpyDeviceToHost, ctx->stream));
                w.chk(hipStreamSynchronize(ctx->stream));
                if (w.err == hipSuccess) {
                    M->vertices.resize(wo.n_verts);
                    M->triangles.resize(wo.n_tris);
                    const size_t CH = (size_t)4 << 20, nv_ch = (vbytes + CH - 1) / CH, nt_ch = (tbytes + CH - 1) / CH;
                    fhmesh::parallel_for(nv_ch + nt_ch, [&](size_t i) {
                        if (i < nv_ch) memcpy((char*)M->vertices.data() + i * CH, pv + i * CH, std::min(CH, vbytes - i * CH));
                        else { const size_t j = i - nv_ch; memcpy((char*)M->triangles.data() + j * CH, pt + j * CH, std::min(CH, tbytes - j * CH)); }
                    });
                    w.release();
                    x.release();
                    M->octree_cells = oo.n_blocks; M->octree_verts = oo.n_verts;
                    if (T.on)
                        fprintf(stderr, "fhip mesh depth %u: cells %.4f s (%llu evaluated), leaf kernel %.4f s (%u leaves), assembly on the device %.4f s (%u blocks, %u vertices), "
                                        "dual walk on the device + the mesh to the host %.4f s (%u levels, %llu items, %u quads), total %.4f s\n",
                                depth, T.t_cells, (unsigned long long)M->cells_evaluated, T.t_leaf, T.n_leaf_cells, t_asm, oo.n_blocks, oo.n_verts, now() - t0 - t_asm,
                                wo.levels, (unsigned long long)wo.items, wo.n_records, now() - T.t_start);
                    return hipSuccess;
                }
            }
        }
        (void)hipStreamSynchronize(ctx->stream);
        w.release();        // (the host's walk takes over: too big an octree for the passes' 32-bit numbers, or no memory for them)
    }
    const size_t cell_bytes = (size_t)oo.n_blocks * 8 * sizeof(fhmesh::Cell);
    if (ctx->mesh_pinned_cap < cell_bytes + 256) {
        if (ctx->mesh_pinned) (void)hipHostFree(ctx->mesh_pinned);
        ctx->mesh_pinned = nullptr; ctx->mesh_pinned_cap = 0;
        const size_t room = cell_bytes + cell_bytes / 8 + 256;
        const hipError_t e = hipHostMalloc(&ctx->mesh_pinned, room, hipHostMallocDefault);
        if (e != hipSuccess) return give_up(e);
        ctx->mesh_pinned_cap = room;
    }
    fhmesh::Octree o;
    o.root = oo.root;
    o.cells_view = (const std::array<fhmesh::Cell, 8>*)ctx->mesh_pinned; o.n_cells_view = oo.n_blocks;
    o.verts_view = nullptr; o.n_verts_view = oo.n_verts;      // (never read: the walk gathers through the device)
    if (cell_bytes) x.chk(hipMemcpyAsync(ctx->mesh_pinned, oo.cells, cell_bytes, hipMemcpyDeviceToHost, ctx->stream));
    x.chk(hipStreamSynchronize(ctx->stream));
    if (x.err != hipSuccess) return give_up(x.err);
    const double t_asm = now() - t0;
    fhmesh::ParallelWalker W(o);
    W.scratch = &ctx->mesh_first; W.scratch_cap = &ctx->mesh_first_cap;
    W.gather = [&](const uint32_t* idx, size_t n, fhmesh::V3* out) {
        if (!n) return true;
        uint32_t* d_idx = (uint32_t*)x.alloc(n * 4);
        fhmesh::V3* d_out = (fhmesh::V3*)x.alloc(n * sizeof(fhmesh::V3));
        if (!d_idx || !d_out) return false;
        x.chk(hipMemcpyAsync(d_idx, idx, n * 4, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(fhm::k_oct_gather, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const fhmesh::V3*)oo.verts, (const uint32_t*)d_idx, d_out, (uint32_t)n);
        x.chk(hipGetLastError());
        x.chk(hipMemcpyAsync(out, d_out, n * sizeof(fhmesh::V3), hipMemcpyDeviceToHost, ctx->stream));
        x.chk(hipStreamSynchronize(ctx->stream));
        return x.err == hipSuccess;
    };
    W.run();
    const double t_walk = now() - t0 - t_asm;
    if (W.gather_failed) return give_up(x.err != hipSuccess ? x.err : hipErrorOutOfMemory);
    x.release();
    M->octree_cells = oo.n_blocks; M->octree_verts = oo.n_verts;
    M->vertices.swap(W.vertices);
    M->triangles.swap(W.triangles);
    if (T.on)
        fprintf(stderr, "fhip mesh depth %u: cells %.4f s (%llu evaluated), leaf kernel %.4f s (%u leaves), assembly on the device + cells to the host %.4f s (%u blocks, %u vertices), "
                        "dual walk + the mesh's vertices gathered %.4f s, total %.4f s\n",
                depth, T.t_cells, (unsigned long long)M->cells_evaluated, T.t_leaf, T.n_leaf_cells, t_asm, oo.n_blocks, oo.n_verts, t_walk, now() - T.t_start);
    return hipSuccess;
}
fhip_status fhip_mesh_sample(fhip_ctx* ctx, const fhip_tape* tape, uint32_t depth, const float* world_to_model, const int32_t* axis_slots,
                             const uint64_t* var_keys, const float* var_values, uint32_t n_vars, fhip_mesh** out) {
    return mesh_run(ctx, tape, depth, world_to_model, axis_slots, var_keys, var_values, n_vars, MESH_SAMPLE, 0, 1, out);
}
// Octree::build + Octree::walk_dual (octree.rs:48-68, 219-225): fhip_mesh_sample, then the octree assembled from the device's
// results (cell collapse included) and the dual walk, both on the device (mesh_assemble_device)
fhip_status fhip_mesh_build(fhip_ctx* ctx, const fhip_tape* tape, uint32_t depth, const float* world_to_model, const int32_t* axis_slots,
                            const uint64_t* var_keys, const float* var_values, uint32_t n_vars, fhip_mesh** out) {
    return mesh_run(ctx, tape, depth, world_to_model, axis_slots, var_keys, var_values, n_vars, MESH_BUILD, 0, 1, out);
}
// ---- the build sharded by the root's octants (Octree::build_inner_mt, octree.rs:94-210, across GPUs): every part runs the
// device side for its octants; the parts' results travel as flat buffers to one place, where fhip_mesh_merge puts the level
// arrays together (slots of later parts shifted by the ambiguous cells before them) and runs assembly and dual walk
fhip_status fhip_mesh_sample_part(fhip_ctx* ctx, const fhip_tape* tape, uint32_t depth, const float* world_to_model, const int32_t* axis_slots,
                                  const uint64_t* var_keys, const float* var_values, uint32_t n_vars, uint32_t part, uint32_t n_parts, fhip_mesh** out) {
    return mesh_run(ctx, tape, depth, world_to_model, axis_slots, var_keys, var_values, n_vars, MESH_PART, part, n_parts, out);
}
namespace {
struct MeshPartHeader {       // followed by n_levels u64 level sizes, then per level {cls bytes padded to 8, slot words padded to 8}, then the leaf records
    uint32_t magic, version, depth, part, n_parts, n_levels, leaf_size, pad;
    uint64_t n_leaves, cells_evaluated, full, empty;
};
constexpr uint32_t MESH_PART_MAGIC = 0x504d4846u;     // "FHMP"
inline uint64_t pad8(uint64_t n) { return (n + 7) & ~7ull; }
}
uint64_t fhip_mesh_part_bytes(const fhip_mesh* m) {
    uint64_t n = sizeof(MeshPartHeader) + 8ull * m->cls.size();
    for (auto& c : m->cls) n += pad8(c.size()) + pad8(4ull * c.size());
    return n + (uint64_t)m->leaves.size() * sizeof(FhMeshLeaf);
}
void fhip_mesh_part_export(const fhip_mesh* m, void* out) {
    char* p = (char*)out;
    MeshPartHeader h;
    memset(&h, 0, sizeof(h));
    h.magic = MESH_PART_MAGIC; h.version = 1; h.depth = m->depth; h.part = m->part; h.n_parts = m->n_parts; h.n_levels = (uint32_t)m->cls.size();
    h.leaf_size = (uint32_t)sizeof(FhMeshLeaf); h.n_leaves = m->leaves.size(); h.cells_evaluated = m->cells_evaluated; h.full = m->full; h.empty = m->empty;
    memcpy(p, &h, sizeof(h)); p += sizeof(h);
    for (auto& c : m->cls) { const uint64_t n = c.size(); memcpy(p, &n, 8); p += 8; }
    for (size_t d = 0; d < m->cls.size(); d++) {
        const size_t n = m->cls[d].size();
        memset(p, 0, pad8(n)); memcpy(p, m->cls[d].data(), n); p += pad8(n);
        memset(p, 0, pad8(4 * n)); memcpy(p, m->slot[d].data(), 4 * n); p += pad8(4 * n);
    }
    if (m->leaves.size()) memcpy(p, m->leaves.data(), m->leaves.size() * sizeof(FhMeshLeaf));
}
fhip_status fhip_mesh_merge(fhip_ctx* ctx, const void* const* parts, const uint64_t* part_bytes, uint32_t n_parts, const float* world_to_model, fhip_mesh** out) {
    if (!out) return FHIP_ERR_BAD_TAPE;
    *out = nullptr;
    if (!parts || !part_bytes || n_parts < 1 || n_parts > 8) return fail(ctx, FHIP_ERR_UNSUPPORTED, "mesh merge: 1..8 parts");
    struct View { MeshPartHeader h; const uint64_t* level_n; std::vector<const uint8_t*> cls; std::vector<const uint32_t*> slot; const FhMeshLeaf* leaves; };
    std::vector<View> V(n_parts);
    for (uint32_t k = 0; k < n_parts; k++) {       // part k of the array must BE part k
        const char* p = (const char*)parts[k];
        if (!p || part_bytes[k] < sizeof(MeshPartHeader)) return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: short part");
        View& v = V[k];
        memcpy(&v.h, p, sizeof(v.h));
        if (v.h.magic != MESH_PART_MAGIC || v.h.version != 1 || v.h.leaf_size != sizeof(FhMeshLeaf)) return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: not a mesh part of this library");
        if (v.h.n_parts != n_parts || v.h.part != k || v.h.depth != V[0].h.depth || v.h.n_levels < 1 || v.h.n_levels > 21)
            return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: parts do not belong together (part index, part count or depth)");
        uint64_t need = sizeof(MeshPartHeader) + 8ull * v.h.n_levels;
        if (part_bytes[k] < need) return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: short part");
        v.level_n = (const uint64_t*)(p + sizeof(MeshPartHeader));
        const char* q = p + need;
        for (uint32_t d = 0; d < v.h.n_levels; d++) {
            const uint64_t n = v.level_n[d];
            if (n > (1ull << 30)) return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: level size");
            need += pad8(n) + pad8(4 * n);
            if (part_bytes[k] < need) return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: short part");
            v.cls.push_back((const uint8_t*)q); q += pad8(n);
            v.slot.push_back((const uint32_t*)q); q += pad8(4 * n);
        }
        if (part_bytes[k] < need + v.h.n_leaves * sizeof(FhMeshLeaf)) return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: short part");
        v.leaves = (const FhMeshLeaf*)q;
    }
    const uint32_t depth = V[0].h.depth;
    fhip_mesh* M = new fhip_mesh();
    M->depth = depth;
    // the root: evaluated by every part, with the same result
    for (uint32_t k = 0; k < n_parts; k++)
        if (V[k].level_n[0] != 1 || V[k].cls[0][0] != V[0].cls[0][0]) { delete M; return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: the parts disagree about the root cell"); }
    const bool whole = n_parts == 1 || V[0].h.n_levels == 1;       // nothing below the root (decided, or a leaf at depth 0): part 0 has it all
    const uint32_t np = whole ? 1 : n_parts;
    uint32_t levels = 0;
    for (uint32_t k = 0; k < np; k++) levels = std::max(levels, V[k].h.n_levels);
    M->cells_evaluated = 1; M->full = 0; M->empty = 0;
    for (uint32_t k = 0; k < np; k++) { M->cells_evaluated += V[k].h.cells_evaluated - 1; M->full += V[k].h.full; M->empty += V[k].h.empty; }
    if (!whole && V[0].cls[0][0] != 3) { delete M; return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: levels below a decided root"); }
    std::vector<uint64_t> shift(np, 0);       // slots of part k at the level before: + shift[k]
    M->cls.resize(levels); M->slot.resize(levels);
    for (uint32_t d = 0; d < levels; d++) {
        std::vector<uint8_t>& C = M->cls[d];
        std::vector<uint32_t>& S = M->slot[d];
        std::vector<uint64_t> amb(np, 0);
        if (d == 0) { C.assign(1, V[0].cls[0][0]); S.assign(1, V[0].slot[0][0]); if (!whole) S[0] = 0; amb.assign(np, 0); }
        else if (d == 1 && !whole) {     // the root's eight children, each from the part that owns it
            C.assign(8, 0); S.assign(8, 0xFFFFFFFFu);
            for (uint32_t k = 0; k < np; k++) {
                if (V[k].h.n_levels < 2 || V[k].level_n[1] != 8) { delete M; return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: a part without the root's children"); }
                const uint32_t mask = mesh_part_mask(k, n_parts);
                for (uint32_t o = 0; o < 8; o++) {
                    const uint8_t c = V[k].cls[1][o];
                    if (((mask >> o) & 1u) != (c != 0 ? 1u : 0u)) { delete M; return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: a part covers the wrong octants"); }
                    if (c == 3) amb[k]++;
                }
            }
            uint64_t off = 0;
            for (uint32_t k = 0; k < np; k++) {
                for (uint32_t o = 0; o < 8; o++) if (V[k].cls[1][o]) { C[o] = V[k].cls[1][o]; S[o] = V[k].cls[1][o] == 3 ? (uint32_t)(off + V[k].slot[1][o]) : 0xFFFFFFFFu; }
                shift[k] = off; off += amb[k];
            }
            continue;
        } else {
            // children of the level above's ambiguous cells: part k's array sits at 8 * (its slots' shift at the level above)
            uint64_t total = 0;
            for (uint32_t k = 0; k < np; k++) total += V[k].h.n_levels > d ? V[k].level_n[d] : 0;
            C.resize(total); S.resize(total);
            // (two passes over the parts, each on the host's threads: the ambiguous cells of every part, then the copies with
            //  the slots shifted by the ambiguous cells of the parts before)
            std::vector<uint64_t> n_of(np, 0), at_of(np, 0), next_shift(np, 0);
            uint64_t at = 0;
            for (uint32_t k = 0; k < np; k++) {
                n_of[k] = V[k].h.n_levels > d ? V[k].level_n[d] : 0;
                if (at != shift[k] * 8 && n_of[k]) { delete M; return fail(ctx, FHIP_ERR_BAD_TAPE, "mesh merge: level arrays do not line up"); }
                at_of[k] = at; at += n_of[k];
            }
            fhmesh::parallel_for(np, [&](size_t k) {
                uint64_t a = 0;
                const uint8_t* c = n_of[k] ? V[k].cls[d] : nullptr;
                for (uint64_t i = 0; i < n_of[k]; i++) a += c[i] == 3;
                amb[k] = a;
            });
            uint64_t off = 0;
            for (uint32_t k = 0; k < np; k++) { next_shift[k] = off; off += amb[k]; }
            constexpr uint64_t CHUNK = 1u << 20;
            std::vector<std::array<uint64_t, 3>> jobs;      // part, first cell, cells
            for (uint32_t k = 0; k < np; k++) for (uint64_t i = 0; i < n_of[k]; i += CHUNK) jobs.push_back({k, i, std::min(CHUNK, n_of[k] - i)});
            fhmesh::parallel_for(jobs.size(), [&](size_t j) {
                const uint32_t k = (uint32_t)jobs[j][0];
                const uint8_t* c = V[k].cls[d] + jobs[j][1];
                const uint32_t* sl = V[k].slot[d] + jobs[j][1];
                uint8_t* co = C.data() + at_of[k] + jobs[j][1];
                uint32_t* so = S.data() + at_of[k] + jobs[j][1];
                const uint32_t sh = (uint32_t)next_shift[k];
                for (uint64_t i = 0; i < jobs[j][2]; i++) { co[i] = c[i]; so[i] = c[i] == 3 ? sh + sl[i] : 0xFFFFFFFFu; }
            });
            shift = next_shift;
            continue;
        }
    }
    // leaf records: in part order (= slot order at the leaf depth), left where they are
    uint64_t n_leaves = 0;
    M->leaves.seg_start.push_back(0);
    for (uint32_t k = 0; k < np; k++) {
        M->leaves.seg_p.push_back(V[k].leaves);
        n_leaves += V[k].h.n_leaves;
        M->leaves.seg_start.push_back(n_leaves);
    }
    M->leaves.n = n_leaves;
    M->ambiguous_leaves = n_leaves;
    for (uint32_t d = 0; d < levels; d++) M->per_level.push_back(M->cls[d].size());
    float mat[16];
    bool ident = true;
    if (world_to_model) for (int i = 0; i < 16; i++) { mat[i] = world_to_model[i]; ident &= world_to_model[i] == ((i % 5 == 0) ? 1.0f : 0.0f); }
    MeshTimes MT{getenv("FHIP_MESH_TIMES") != nullptr, std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(), 0, 0, 0, (uint32_t)n_leaves};
    mesh_assemble(ctx, M, depth, world_to_model && !ident, mat, MT);
    *out = M;
    return FHIP_OK;
}
void fhip_debug_walk_dual(const uint32_t* cells, uint64_t n_cells, const uint32_t* root, const float* verts, uint64_t n_verts, int parallel,
                          uint64_t counts[2], uint64_t* tris, float* verts_out) {
    fhmesh::Octree o;
    auto cell = [](const uint32_t* w) { fhmesh::Cell c; c.kind = (uint8_t)w[0]; c.mask = (uint8_t)w[1]; c.index = w[2]; return c; };
    o.root = cell(root);
    o.cells.resize(n_cells);
    for (uint64_t i = 0; i < n_cells; i++) for (int k = 0; k < 8; k++) o.cells[i][k] = cell(cells + (i * 8 + k) * 3);
    o.verts.resize(n_verts);
    for (uint64_t i = 0; i < n_verts; i++) o.verts[i] = fhmesh::V3{verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]};
    fhmesh::TriVec t;
    fhmesh::VertVec v;
    if (parallel == 3) {     // the passes fhip_mesh_build runs on the device (mesh_walk.hpp), here as loops on the host
        WalkHostX hx;
        fhmesh::WalkOut wo;
        const int rc = fhmesh::walk_dual_passes(hx, (const fhmesh::Cell*)o.cells.data(), (uint32_t)o.cells.size(), o.root, o.verts.data(), (uint32_t)o.verts.size(), &walk_table(), &wo);
        counts[0] = counts[1] = 0;
        if (rc != fhmesh::WALK_OK) { counts[0] = ~0ull; return; }
        counts[0] = wo.n_tris; counts[1] = wo.n_verts;
        if (tris && wo.tris) memcpy(tris, wo.tris, (size_t)wo.n_tris * 24);
        if (verts_out && wo.verts) memcpy(verts_out, wo.verts, (size_t)wo.n_verts * 12);
        hx.free(wo.tris); hx.free(wo.verts);
        return;
    }
    if (parallel == 2) {     // as fhip_mesh_build runs it: the cells through a view, the octree's vertices never read - the mesh's are gathered afterwards
        fhmesh::Octree w;
        w.root = o.root;
        w.cells_view = o.cells.data(); w.n_cells_view = o.cells.size(); w.n_verts_view = o.verts.size();
        fhmesh::ParallelWalker W(w);
        W.gather = [&](const uint32_t* idx, size_t n, fhmesh::V3* out) { for (size_t i = 0; i < n; i++) out[i] = o.verts[idx[i]]; return true; };
        W.run();
        t.swap(W.triangles); v.swap(W.vertices);
    } else if (parallel) { fhmesh::ParallelWalker W(o); W.run(); t.swap(W.triangles); v.swap(W.vertices); }
    else { fhmesh::Walker W(o); W.cell(fhmesh::CellRef()); t.swap(W.triangles); v.swap(W.vertices); }
    counts[0] = t.size(); counts[1] = v.size();
    if (tris) memcpy(tris, t.data(), t.size() * 24);
    if (verts_out) memcpy(verts_out, v.data(), v.size() * 12);
}
void fhip_mesh_vertices(const fhip_mesh* m, float* out) { memcpy(out, m->vertices.data(), m->vertices.size() * 12); }
void fhip_mesh_triangles(const fhip_mesh* m, uint64_t* out) { memcpy(out, m->triangles.data(), m->triangles.size() * 24); }
const float* fhip_mesh_vertices_ptr(const fhip_mesh* m) { return (const float*)m->vertices.data(); }
const uint64_t* fhip_mesh_triangles_ptr(const fhip_mesh* m) { return (const uint64_t*)m->triangles.data(); }
void fhip_mesh_free(fhip_mesh* m) { delete m; }
// out = {cells evaluated (= interval evaluations), Full, Empty, ambiguous cells at the leaf depth (= calls of leaf()), bytes per leaf record, levels}
void fhip_mesh_counts(const fhip_mesh* m, uint64_t out[8]) {
    out[0] = m->cells_evaluated; out[1] = m->full; out[2] = m->empty; out[3] = m->ambiguous_leaves; out[4] = sizeof(FhMeshLeaf);
    out[5] = m->per_level.size(); out[6] = m->vertices.size(); out[7] = m->triangles.size();
}
void fhip_mesh_leaves(const fhip_mesh* m, void* out) { memcpy(out, m->leaves.data(), m->leaves.size() * sizeof(FhMeshLeaf)); }
