// ORACLE — TEST INFRASTRUCTURE ONLY (see types.hpp header).
//
// Flat C API over the oracle so tests (ctypes) and bench.py's cpu_baseline leg
// can drive it.  Nothing in fidget_amd/ links or loads this library.
#include <chrono>

#include "render.hpp"
#include "effects.hpp"
#include "mesh.hpp"

namespace orc {
thread_local uint64_t g_invalid_intervals = 0;
}
using namespace orc;

struct OrcShape {
    VmDataP d;
};

// memcpy whose source may be an empty vector's null data(): undefined behaviour even for zero bytes (found by UBSan, tests/test_sanitizers.py)
static inline void copy_bytes(void* dst, const void* src, size_t n) {
    if (n) std::memcpy(dst, src, n);
}

extern "C" {

// ---- Context -------------------------------------------------------------
void* orc_ctx_new() { return new Context(); }
void orc_ctx_free(void* c) { delete (Context*)c; }
uint32_t orc_ctx_len(void* c) { return (uint32_t)((Context*)c)->len(); }
uint32_t orc_ctx_x(void* c) { return ((Context*)c)->x(); }
uint32_t orc_ctx_y(void* c) { return ((Context*)c)->y(); }
uint32_t orc_ctx_z(void* c) { return ((Context*)c)->z(); }
uint32_t orc_ctx_var(void* c, uint64_t index) { return ((Context*)c)->var(Var{3, index}); }
uint32_t orc_ctx_constant(void* c, float f) { return ((Context*)c)->constant(f); }
// opcode numbering = UnaryOpcode / BinaryOpcode declaration order (context/op.rs)
uint32_t orc_ctx_unary(void* c, int op, uint32_t a) {
    Context* x = (Context*)c;
    switch ((UnaryOpcode)op) {
        case U_NEG: return x->neg(a);
        case U_ABS: return x->abs(a);
        case U_RECIP: return x->recip(a);
        case U_SQRT: return x->sqrt(a);
        case U_SQUARE: return x->square(a);
        case U_FLOOR: return x->floor(a);
        case U_CEIL: return x->ceil(a);
        case U_ROUND: return x->round(a);
        case U_SIN: return x->sin(a);
        case U_COS: return x->cos(a);
        case U_TAN: return x->tan(a);
        case U_ASIN: return x->asin(a);
        case U_ACOS: return x->acos(a);
        case U_ATAN: return x->atan(a);
        case U_EXP: return x->exp(a);
        case U_LN: return x->ln(a);
        case U_NOT: return x->not_(a);
        case U_RAND: return x->rand(a);
    }
    return BAD_NODE;
}
uint32_t orc_ctx_binary(void* c, int op, uint32_t a, uint32_t b) {
    Context* x = (Context*)c;
    switch ((BinaryOpcode)op) {
        case B_ADD: return x->add(a, b);
        case B_SUB: return x->sub(a, b);
        case B_MUL: return x->mul(a, b);
        case B_DIV: return x->div(a, b);
        case B_ATAN: return x->atan2(a, b);
        case B_MIN: return x->min(a, b);
        case B_MAX: return x->max(a, b);
        case B_COMPARE: return x->compare(a, b);
        case B_MOD: return x->modulo(a, b);
        case B_AND: return x->and_(a, b);
        case B_OR: return x->or_(a, b);
        case B_MIX: return x->mix(a, b);
    }
    return BAD_NODE;
}
uint32_t orc_ctx_less_than(void* c, uint32_t a, uint32_t b) { return ((Context*)c)->less_than(a, b); }
uint32_t orc_ctx_less_than_or_equal(void* c, uint32_t a, uint32_t b) { return ((Context*)c)->less_than_or_equal(a, b); }
uint32_t orc_ctx_if_nonzero_else(void* c, uint32_t q, uint32_t a, uint32_t b) { return ((Context*)c)->if_nonzero_else(q, a, b); }
// Returns the root node or 0xFFFFFFFF on a parse error
uint32_t orc_ctx_from_text(void* c, const char* text) {
    try {
        return Context::from_text(*(Context*)c, std::string(text));
    } catch (const std::exception& e) {
        fprintf(stderr, "oracle: from_text: %s\n", e.what());
        return BAD_NODE;
    }
}
// Context::eval_xyz-like scalar evaluation (context/mod.rs:798-856)
float orc_ctx_eval_xyz(void* c, uint32_t root, float x, float y, float z) {
    Context* ctx = (Context*)c;
    std::vector<float> cache(ctx->len(), 0.0f);
    std::vector<uint8_t> done(ctx->len(), 0);
    // nodes are created children-first, so a forward sweep is a valid order
    for (uint32_t i = 0; i <= root && i < ctx->len(); i++) {
        const NodeOp& o = ctx->ops[i];
        switch (o.kind) {
            case N_INPUT: cache[i] = o.var.kind == 0 ? x : o.var.kind == 1 ? y : o.var.kind == 2 ? z : NANF; break;
            case N_CONST: cache[i] = o.c; break;
            case N_UNARY: cache[i] = eval_unary((UnaryOpcode)o.opcode, cache[o.a]); break;
            case N_BINARY: cache[i] = eval_binary((BinaryOpcode)o.opcode, cache[o.a], cache[o.b]); break;
        }
    }
    return cache[root];
}

// ---- Shapes (VmData) -------------------------------------------------------
void* orc_shape_new(void* c, const uint32_t* roots, int n_roots, uint32_t n_regs) {
    std::vector<Node> r(roots, roots + n_roots);
    VmDataP d = vmdata_new(*(Context*)c, r, n_regs);
    if (!d) return nullptr;
    return new OrcShape{d};
}
void orc_shape_free(void* s) { delete (OrcShape*)s; }
uint32_t orc_shape_len(void* s) { return (uint32_t)((OrcShape*)s)->d->len(); }
uint32_t orc_shape_ssa_len(void* s) { return (uint32_t)((OrcShape*)s)->d->ssa.tape.size(); }
uint32_t orc_shape_choice_count(void* s) { return (uint32_t)((OrcShape*)s)->d->choice_count(); }
uint32_t orc_shape_output_count(void* s) { return (uint32_t)((OrcShape*)s)->d->output_count(); }
uint32_t orc_shape_slot_count(void* s) { return (uint32_t)((OrcShape*)s)->d->slot_count(); }
uint32_t orc_shape_var_count(void* s) { return (uint32_t)((OrcShape*)s)->d->vars->len(); }
// index of X/Y/Z (axis 0..2) in the variable list, or -1
int orc_shape_axis_index(void* s, int axis) {
    const VarMap& v = *((OrcShape*)s)->d->vars;
    return axis == 0 ? v.x : axis == 1 ? v.y : v.z;
}
int orc_shape_var_index(void* s, uint64_t index) { return ((OrcShape*)s)->d->vars->get(Var{3, index}); }

// Dump a tape as 6 x u32 records: {op, form, out, a, b, idx} + imm bits in a
// parallel array.  which = 0: SSA tape (root first); 1: register tape in
// evaluation order (iter_asm).
uint32_t orc_shape_dump(void* s, int which, uint32_t* rec, uint32_t* imm, uint32_t cap) {
    const VmData& d = *((OrcShape*)s)->d;
    const std::vector<TOp>& t = which == 0 ? d.ssa.tape : d.asm_.tape;
    uint32_t n = (uint32_t)t.size();
    for (uint32_t i = 0; i < n && i < cap; i++) {
        const TOp& o = which == 0 ? t[i] : t[n - 1 - i];
        rec[i * 6 + 0] = o.op; rec[i * 6 + 1] = o.form; rec[i * 6 + 2] = o.out;
        rec[i * 6 + 3] = o.a; rec[i * 6 + 4] = o.b; rec[i * 6 + 5] = o.idx;
        imm[i] = f2u(o.imm);
    }
    return n;
}

void* orc_shape_simplify(void* s, const uint8_t* choices, uint32_t n, uint32_t n_regs) {
    VmDataP d = vmdata_simplify(*((OrcShape*)s)->d, choices, n, n_regs);
    if (!d) return nullptr;  // BadChoiceSlice
    return new OrcShape{d};
}

// Bytecode::new; returns number of words (call with cap=0 to size)
uint32_t orc_shape_bytecode(void* s, uint32_t* words, uint32_t cap, uint32_t* reg_count, uint32_t* mem_count) {
    Bytecode bc = bytecode_new(*((OrcShape*)s)->d);
    if (bc.reserved_register) return 0;
    if (reg_count) *reg_count = bc.reg_count;
    if (mem_count) *mem_count = bc.mem_count;
    for (uint32_t i = 0; i < bc.data.size() && i < cap; i++) words[i] = bc.data[i];
    return (uint32_t)bc.data.size();
}

// ---- Evaluators ------------------------------------------------------------
// Interval tracing eval: vars = nvars x {lo,hi}; out = output_count x {lo,hi};
// choices = choice_count bytes.  Returns 1 if a trace is reported, 0 if not,
// -1 on BadVarSlice.
int orc_eval_interval(void* s, const float* vars, uint32_t nvars, float* out, uint8_t* choices) {
    const VmData& d = *((OrcShape*)s)->d;
    TracingEval<Interval> e;
    std::vector<Interval> v(nvars);
    for (uint32_t i = 0; i < nvars; i++) { v[i].lo = vars[2 * i]; v[i].hi = vars[2 * i + 1]; }
    int r = e.eval(d, v.data(), nvars);
    if (r < 0) return r;
    for (size_t i = 0; i < e.out.size(); i++) { out[2 * i] = e.out[i].lo; out[2 * i + 1] = e.out[i].hi; }
    if (choices) copy_bytes(choices, e.choices.data(), e.choices.size());
    return r;
}
int orc_eval_point(void* s, const float* vars, uint32_t nvars, float* out, uint8_t* choices) {
    const VmData& d = *((OrcShape*)s)->d;
    TracingEval<float> e;
    int r = e.eval(d, vars, nvars);
    if (r < 0) return r;
    for (size_t i = 0; i < e.out.size(); i++) out[i] = e.out[i];
    if (choices) copy_bytes(choices, e.choices.data(), e.choices.size());
    return r;
}
// Bulk f32: vars[i] -> n floats; out = output_count x n (output-major).
// Returns -1 BadVarSlice.
int orc_eval_float_slice(void* s, const float* const* vars, uint32_t nvars, uint32_t n, float* out) {
    const VmData& d = *((OrcShape*)s)->d;
    BulkEval<float> e;
    int r = e.eval(d, vars, nvars, n);
    if (r < 0) return r;
    for (size_t o = 0; o < e.out.size(); o++) copy_bytes(out + o * n, e.out[o].data(), n * 4);
    return 0;
}
// Bulk grad: vars[i] -> n x {v,dx,dy,dz}
int orc_eval_grad_slice(void* s, const float* const* vars, uint32_t nvars, uint32_t n, float* out) {
    const VmData& d = *((OrcShape*)s)->d;
    BulkEval<Grad> e;
    std::vector<const Grad*> v(nvars);
    for (uint32_t i = 0; i < nvars; i++) v[i] = (const Grad*)vars[i];
    int r = e.eval(d, v.data(), nvars, n);
    if (r < 0) return r;
    for (size_t o = 0; o < e.out.size(); o++) copy_bytes(out + o * n * 4, e.out[o].data(), n * 16);
    return 0;
}
uint64_t orc_invalid_intervals() { return g_invalid_intervals; }
void orc_reset_invalid_intervals() { g_invalid_intervals = 0; }

// ---- Geometry ----------------------------------------------------------------
// screen_to_world for 2D (3x3, size = {w,h}) or 3D (4x4, size = {w,h,d})
void orc_screen_to_world(const uint32_t* size, int n, float* out) { screen_to_world(size, n, out); }
void orc_mat_mul(const float* a, const float* b, int dim, float* out) { mat_mul(a, b, dim, out); }
void orc_transform_point(const float* mat4, float x, float y, float z, float* out) {
    Mat4 m;
    copy_bytes(m.m, mat4, 64);
    transform_f32(m, x, y, z, out, out + 1, out + 2);
}
void orc_transform_interval(const float* mat4, const float* xyz /*3x{lo,hi}*/, float* out /*3x{lo,hi}*/) {
    Mat4 m;
    copy_bytes(m.m, mat4, 64);
    Interval o[3];
    transform_interval(m, Interval(xyz[0], xyz[1]), Interval(xyz[2], xyz[3]), Interval(xyz[4], xyz[5]), o);
    for (int i = 0; i < 3; i++) { out[2 * i] = o[i].lo; out[2 * i + 1] = o[i].hi; }
}

// ---- Renders -----------------------------------------------------------------
// stats: array of sizeof(RenderStats)/8 uint64 (see orc_stats_names)
static const char* STATS_NAMES =
    "interval_evals,interval_ops,interval_choices,simplify_calls,simplify_in,simplify_out,simplify_kept,"
    "tiles_full,tiles_empty,tiles_ambiguous,tiles_skipped,float_evals,float_points,float_lane_ops,"
    "float_wave_ops,grad_evals,grad_points,grad_lane_ops,grad_tape_ops,invalid_intervals";
const char* orc_stats_names() { return STATS_NAMES; }
uint32_t orc_stats_count() { return (uint32_t)(sizeof(RenderStats) / 8); }

// world_to_model: row-major 3x3 (may be NULL = identity).  Returns 0 ok, -1 unbound vars.
int orc_render2d(void* s, const float* world_to_model, uint32_t w, uint32_t h, float z, int pixel_perfect,
                 const uint32_t* tiles, uint32_t n_tiles, int mode, int threads, float* out, uint64_t* stats,
                 double* seconds, const uint64_t* var_keys, const float* var_vals, uint32_t n_vars) {
    uint32_t size[2] = {w, h};
    float s2w[9], mat[9];
    screen_to_world(size, 2, s2w);
    if (world_to_model) mat_mul(world_to_model, s2w, 3, mat);
    else {
        const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        mat_mul(I, s2w, 3, mat);
    }
    std::vector<uint32_t> ts(tiles, tiles + n_tiles);
    auto t0 = std::chrono::steady_clock::now();
    RenderResult r = render_2d(((OrcShape*)s)->d, mat, w, h, z, pixel_perfect != 0, ts, mode, threads, out, var_keys,
                               var_vals, n_vars);
    auto t1 = std::chrono::steady_clock::now();
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
    if (stats) copy_bytes(stats, &r.stats, sizeof(RenderStats));
    return r.ok ? 0 : -1;
}
// world_to_model: row-major 4x4 (may be NULL).  out = w*h GeometryPixel {f32 normal[3]; u32 depth}
int orc_render3d(void* s, const float* world_to_model, uint32_t w, uint32_t h, uint32_t d, const uint32_t* tiles,
                 uint32_t n_tiles, int mode, int threads, void* out, uint64_t* stats, double* seconds,
                 const uint64_t* var_keys, const float* var_vals, uint32_t n_vars) {
    uint32_t size[3] = {w, h, d};
    float s2w[16], mat[16];
    screen_to_world(size, 3, s2w);
    if (world_to_model) mat_mul(world_to_model, s2w, 4, mat);
    else {
        const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        mat_mul(I, s2w, 4, mat);
    }
    std::vector<uint32_t> ts(tiles, tiles + n_tiles);
    auto t0 = std::chrono::steady_clock::now();
    RenderResult r = render_3d(((OrcShape*)s)->d, mat, w, h, d, ts, mode, threads, (GeometryPixel*)out, var_keys,
                               var_vals, n_vars);
    auto t1 = std::chrono::steady_clock::now();
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
    if (stats) copy_bytes(stats, &r.stats, sizeof(RenderStats));
    return r.ok ? 0 : -1;
}

int orc_max_threads() { return omp_get_max_threads(); }


// ---- effects (fidget-raster/src/effects.rs) ---------------------------------------------
void orc_fx_denoise_normals(const void* img, int w, int h, void* out) { fx_denoise_normals((const GeomPx*)img, w, h, (GeomPx*)out); }
void orc_fx_compute_ssao(const void* img, int w, int h, int d, const float* kernel, int nk, const float* noise, int nn, float* out) {
    fx_compute_ssao((const GeomPx*)img, w, h, d, kernel, nk, noise, nn, out);
}
void orc_fx_blur_ssao(const float* ssao, int w, int h, float* out) { fx_blur_ssao(ssao, w, h, out); }
void orc_fx_apply_shading(const void* img, int w, int h, int d, const float* ssao, uint8_t* out) {
    fx_apply_shading((const GeomPx*)img, w, h, d, ssao, out);
}
void orc_fx_to_rgba_bitmap(const float* img, uint64_t n, int transparent, uint8_t* out) { fx_to_rgba_bitmap(img, n, transparent, out); }
void orc_fx_to_debug_bitmap(const float* img, uint64_t n, uint8_t* out) { fx_to_debug_bitmap(img, n, out); }
void orc_fx_to_rgba_distance(const float* img, uint64_t n, uint8_t* out) { fx_to_rgba_distance(img, n, out); }

// ---- libm sweep: the reference's transcendental opcodes call the platform libm in f32 (glibc here) --------------------
// out[i] = f(x_i), x_i = the float with bit pattern first + i * stride; op: 0 sin 1 cos 2 tan 3 asin 4 acos 5 atan 6 exp 7 ln
// 8 atan2(x_i, w_i) with w_i = the float with bit pattern x_i's bits * 2654435761 + 0x9E3779B9 (a second argument that covers the range)
void orc_math_unary(int op, uint32_t first, uint32_t stride, uint64_t count, float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)count; i++) {
        const float x = u2f(first + (uint32_t)i * stride);
        float y;
        switch (op) {
            case 0: y = sinf(x); break;
            case 1: y = cosf(x); break;
            case 2: y = tanf(x); break;
            case 3: y = asinf(x); break;
            case 4: y = acosf(x); break;
            case 5: y = atanf(x); break;
            case 6: y = expf(x); break;
            case 8: y = atan2f(x, u2f((first + (uint32_t)i * stride) * 2654435761u + 0x9E3779B9u)); break;
            default: y = logf(x); break;
        }
        out[i] = y;
    }
}

// ---- fidget-mesh: Octree::build / walk_dual (octree.rs:48-68, 219-225) ------------------------------------
struct OrcMesh {
    mesh::Octree o;
    mesh::MeshOut m;
    bool walked = false;
};
// world_to_model: row-major 4x4 or NULL (= identity); returns NULL if a variable has no value
static void* mesh_build(void* s, const float* world_to_model, uint32_t depth, int mode, const uint64_t* var_keys, const float* var_vals, uint32_t n_vars, int threads,
                        int keep_samples) {
    VmDataP shape = ((OrcShape*)s)->d;
    Axes axes(*shape->vars);
    if (!bind_vars(*shape->vars, var_keys, var_vals, n_vars, axes)) return nullptr;
    Mat4 m;
    bool ident = true;
    if (world_to_model) {
        for (int i = 0; i < 16; i++) { m.m[i] = world_to_model[i]; ident &= (world_to_model[i] == ((i % 5 == 0) ? 1.0f : 0.0f)); }
    }
    OrcMesh* out = new OrcMesh();
    if (threads > 0) {      // Settings::threads = Some(pool): Octree::build_inner_mt (octree.rs:94-210)
        out->o = mesh::build_mt(shape, depth, (world_to_model && !ident) ? &m : nullptr, axes, mode, threads, keep_samples != 0, nullptr);
    } else {
        mesh::Builder b(depth, (world_to_model && !ident) ? &m : nullptr, axes, mode);
        b.keep_samples = keep_samples != 0;
        RenderHandle root(shape);
        mesh::Hermite h;
        b.recurse(&root, mesh::CellIndex(), &h);
        out->o = std::move(b.o);
    }
    if (world_to_model && !ident) {      // octree.rs:58-65: vertices back to model space
        for (auto& v : out->o.verts) transform_f32(m, v.x, v.y, v.z, &v.x, &v.y, &v.z);
    }
    return out;
}
void* orc_mesh_build(void* s, const float* world_to_model, uint32_t depth, int mode, const uint64_t* var_keys, const float* var_vals, uint32_t n_vars) {
    return mesh_build(s, world_to_model, depth, mode, var_keys, var_vals, n_vars, 0, 1);
}
// threads > 0: the multithreaded constructor (the reference's Settings::threads); keep_samples 0: no per-leaf sampling records
void* orc_mesh_build_mt(void* s, const float* world_to_model, uint32_t depth, int mode, const uint64_t* var_keys, const float* var_vals, uint32_t n_vars, int threads,
                        int keep_samples) {
    return mesh_build(s, world_to_model, depth, mode, var_keys, var_vals, n_vars, threads, keep_samples);
}
void orc_mesh_free(void* h) { delete (OrcMesh*)h; }
// counts: cells (groups of 8), verts, leaf samples, interval evaluations, root kind, root mask, root index
void orc_mesh_counts(void* h, uint64_t out[8]) {
    OrcMesh* m = (OrcMesh*)h;
    out[0] = m->o.cells.size(); out[1] = m->o.verts.size(); out[2] = m->o.samples.size(); out[3] = m->o.interval_evals;
    out[4] = m->o.root.kind; out[5] = m->o.root.mask; out[6] = m->o.root.index; out[7] = 0;
}
void orc_mesh_verts(void* h, float* out) { OrcMesh* m = (OrcMesh*)h; copy_bytes(out, m->o.verts.data(), m->o.verts.size() * 12); }
// cells: per cell 3 x u32 (kind, mask, index), 8 per group
void orc_mesh_cells(void* h, uint32_t* out) {
    OrcMesh* m = (OrcMesh*)h;
    size_t k = 0;
    for (auto& g : m->o.cells) for (auto& c : g) { out[k++] = c.kind; out[k++] = c.mask; out[k++] = c.index; }
}
// leaf samples: bounds f32[6] (x.lo x.hi y.lo y.hi z.lo z.hi), u32 mask, n_edges, n_verts; u16 inter[12][3]; f32 pos[12][3]; f32 grad[12][4]
// (dx dy dz v); f32 vert[4][3]
void orc_mesh_samples(void* h, float* bounds, uint32_t* info, uint16_t* inter, float* pos, float* grad, float* vert) {
    OrcMesh* m = (OrcMesh*)h;
    size_t i = 0;
    for (auto& s : m->o.samples) {
        for (int k = 0; k < 3; k++) { bounds[i * 6 + 2 * k] = s.bounds.b[k].lo; bounds[i * 6 + 2 * k + 1] = s.bounds.b[k].hi; }
        info[i * 3] = s.mask; info[i * 3 + 1] = s.n_edges; info[i * 3 + 2] = s.n_verts;
        copy_bytes(inter + i * 36, s.inter, 72); copy_bytes(pos + i * 36, s.pos, 144); copy_bytes(grad + i * 48, s.grad, 192);
        copy_bytes(vert + i * 12, s.vert, 48);
        i++;
    }
}
// walk_dual: returns (triangles, vertices) counts in out[0..1]; then copy with orc_mesh_dual_copy
void orc_mesh_walk_dual(void* h, uint64_t out[2]) {
    OrcMesh* m = (OrcMesh*)h;
    if (!m->walked) { mesh::Walker w(m->o); w.cell(mesh::CellIndex()); m->m = std::move(w.out); m->walked = true; }
    out[0] = m->m.triangles.size(); out[1] = m->m.vertices.size();
}
void orc_mesh_dual_copy(void* h, uint64_t* tris, float* verts) {
    OrcMesh* m = (OrcMesh*)h;
    copy_bytes(tris, m->m.triangles.data(), m->m.triangles.size() * 24);
    copy_bytes(verts, m->m.vertices.data(), m->m.vertices.size() * 12);
}
// CELL_TO_VERT_TO_EDGES / CELL_TO_EDGE_TO_VERT (build.rs): for mask: out[0] = vertices, then per vertex: count, (start, end)...; e2v[12][2]
void orc_mesh_table(int mask, int32_t* v2e, int32_t* e2v) {
    const mesh::Tables& T = mesh::tables();
    size_t k = 0;
    v2e[k++] = (int32_t)T.v2e[mask].size();
    for (auto& vs : T.v2e[mask]) { v2e[k++] = (int32_t)vs.size(); for (auto& e : vs) { v2e[k++] = e.first; v2e[k++] = e.second; } }
    for (int e = 0; e < 12; e++) { e2v[2 * e] = T.e2v[mask][e][0]; e2v[2 * e + 1] = T.e2v[mask][e][1]; }
}
// QuadraticErrorSolver (qef.rs): n intersections (pos[3], grad[4]) -> solve
void orc_qef_solve(const float* pos, const float* grad, int n, float* out_pos, float* out_err) {
    mesh::Qef q;
    for (int i = 0; i < n; i++) q.add_intersection(pos + 3 * i, grad + 4 * i);
    q.solve(out_pos, out_err);
}
}  // extern "C"
