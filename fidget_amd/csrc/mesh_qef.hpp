// The quadratic error function of a cell vertex (fidget-mesh/src/qef.rs:45-126), shared by the device leaf kernel (mesh.hip)
// and the host-side octree assembly (host_mesh.hpp: merged QEFs of collapsed cells): ONE definition, so that a leaf vertex and
// a collapsed cell's vertex come out of the same arithmetic.  The reference solves with nalgebra's SVD; here the symmetric 3x3
// A^T A is diagonalised by cyclic Jacobi rotations in f64 (the oracle does the same: oracle/src/mesh.hpp).
#pragma once
#include <math.h>
#if defined(__HIPCC__)
#define FHQ_HD __host__ __device__
#else
#define FHQ_HD
#endif
#define FHQ_SQRTF sqrtf
#define FHQ_FABSF fabsf
namespace fhq {
struct Qef {
    float ata[3][3], atb[3], btb, mass[4];
    FHQ_HD void init() {
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) ata[i][j] = 0.0f; atb[i] = 0.0f; }
        btb = 0.0f;
        for (int i = 0; i < 4; i++) mass[i] = 0.0f;
    }
    FHQ_HD void merge(const Qef& o) {      // AddAssign (qef.rs:20-27)
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) ata[i][j] += o.ata[i][j]; atb[i] += o.atb[i]; }
        btb += o.btb;
        for (int i = 0; i < 4; i++) mass[i] += o.mass[i];
    }
    FHQ_HD void add(const float* pos, const float* grad) {
        mass[0] += pos[0]; mass[1] += pos[1]; mass[2] += pos[2]; mass[3] += 1.0f;
        const float nn = FHQ_SQRTF(0.0f + ((grad[0] * grad[0] + grad[1] * grad[1]) + grad[2] * grad[2]));
        const float n[3] = {grad[0] / nn, grad[1] / nn, grad[2] / nn};
        const float d = (n[0] * pos[0] + n[1] * pos[1]) + n[2] * pos[2];
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) ata[i][j] += n[i] * n[j];
            atb[i] += n[i] * d;
        }
        btb += d * d;
    }
    FHQ_HD void solve(float* pos, float* err) const {
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
                    if (fabs(a[p][q]) < 1e-300) continue;
                    const double th = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                    const double tt = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                    const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
                    for (int k = 0; k < 3; k++) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq; }
                    for (int k = 0; k < 3; k++) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk; }
                    for (int k = 0; k < 3; k++) { const double vkp = v[k][p], vkq = v[k][q]; v[k][p] = c * vkp - s * vkq; v[k][q] = s * vkp + c * vkq; }
                }
        }
        int order[3] = {0, 1, 2};
        for (int i = 0; i < 2; i++)          // stable selection sort, descending |eigenvalue| (std::sort on 3 elements in the oracle)
            for (int j = i + 1; j < 3; j++)
                if (fabs(a[order[j]][order[j]]) > fabs(a[order[i]][order[i]])) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
        float sv[3];
        for (int i = 0; i < 3; i++) sv[i] = (float)fabs(a[order[i]][order[i]]);
        const float cutoff = FHQ_FABSF(sv[0]) * 1e-3f;
        int rank = 3;
        for (int i = 0; i < 3; i++) if (FHQ_FABSF(sv[i]) < cutoff) { rank = i; break; }
        const float eps = rank < 3 ? sv[rank] : 0.0f;
        double sol[3] = {0, 0, 0};
        for (int k = 0; k < 3; k++) {
            const int e = order[k];
            if (!((float)fabs(a[e][e]) > eps)) continue;
            const double proj = (v[0][e] * b[0] + v[1][e] * b[1] + v[2][e] * b[2]) / a[e][e];
            for (int i = 0; i < 3; i++) sol[i] += v[i][e] * proj;
        }
        for (int i = 0; i < 3; i++) pos[i] = (float)sol[i] + center[i];
        float ap[3];
        for (int i = 0; i < 3; i++) ap[i] = (ata[i][0] * pos[0] + ata[i][1] * pos[1]) + ata[i][2] * pos[2];
        float e = ((pos[0] * ap[0] + pos[1] * ap[1]) + pos[2] * ap[2]) - 2.0f * ((pos[0] * atb[0] + pos[1] * atb[1]) + pos[2] * atb[2]);
        e += btb;
        *err = e > 1e-6f ? e : 1e-6f;
    }
};


}  // namespace fhq
