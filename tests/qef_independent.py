"""An oracle-INDEPENDENT check of the mesher's vertices (TEST INFRASTRUCTURE).

Product and oracle place a leaf vertex with the same arithmetic (mesh_qef.hpp / oracle/src/mesh.hpp: the QEF's 3x3 A^T A decomposed by
cyclic Jacobi in f64) where the reference calls nalgebra's f32 SVD (fidget-mesh/src/qef.rs:67-126; nalgebra is not in /root/reference
and is not restated): `device mesh == oracle mesh` therefore says nothing about the solve itself.  This does: from the leaf records
(the intersections' positions and gradients, which ARE pinned bit for bit) it accumulates each vertex's QEF exactly as qef.rs:49-61
does (f32, in the intersections' order), solves it with LAPACK's f64 SVD under qef.rs's rank rule - another algorithm in another
precision than the product's - and measures how far the product's vertex is from that solution, in units of the leaf cell's size.
A rank decision that falls the other way (a singular value within rounding of the 1e-3 cutoff) shows up as a deviation of a
fraction of a cell; rounding of the solve itself as ~1e-7."""
import numpy as np

F32 = np.float32


def per_vertex_counts(mdc_table):
    """[256][4] intersections per vertex of a corner mask, from CELL_TO_VERT_TO_EDGES (the caller passes oracle.mdc_table or any
    other source of that fixed table)"""
    t = np.zeros((256, 4), np.int64)
    for m in range(256):
        v2e, _ = mdc_table(m)
        for vi, es in enumerate(v2e):
            t[m, vi] = len(es)
    return t


def check(leaves, counts, max_vertices=None, seed=0):
    """leaves: MESH_LEAF records (fidget_amd.mesh_sample); counts: per_vertex_counts(...).  Returns a dict of figures."""
    lv = leaves[leaves["n_verts"] > 0]
    if max_vertices is not None and len(lv) > max_vertices:
        lv = lv[np.random.default_rng(seed).choice(len(lv), max_vertices, replace=False)]
    n = len(lv)
    mask = lv["mask"].astype(np.int64)
    per = counts[mask]                                   # [n][4]
    start = np.concatenate([np.zeros((n, 1), np.int64), np.cumsum(per, axis=1)[:, :3]], axis=1)
    size = (lv["bounds"][:, 1] - lv["bounds"][:, 0]).astype(np.float64)
    res = {"leaves": int(n), "vertices": 0, "forced_points": 0}
    devs, near, svs = [], [], []
    for vi in range(4):
        sel = np.flatnonzero(lv["n_verts"] > vi)
        if not len(sel):
            continue
        cnt, st = per[sel, vi], start[sel, vi]
        ata = np.zeros((len(sel), 3, 3), F32); atb = np.zeros((len(sel), 3), F32); mass = np.zeros((len(sel), 4), F32)
        forced = np.zeros(len(sel), bool)
        with np.errstate(all="ignore"):
            for k in range(int(cnt.max())):
                on = (k < cnt) & ~forced
                idx = np.minimum(st + k, 11)
                p = lv["pos"][sel, idx]                      # [m][3] f32
                g = lv["grad"][sel, idx]                     # [m][4]
                bad = on & np.isnan(g).any(axis=1)
                forced |= bad                                # (octree.rs:818-823: the vertex is that point, no QEF)
                on &= ~bad
                nn = np.sqrt(((g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1]) + g[:, 2] * g[:, 2]).astype(F32)).astype(F32)
                nrm = (g[:, :3] / nn[:, None]).astype(F32)
                d = ((nrm[:, 0] * p[:, 0] + nrm[:, 1] * p[:, 1]) + nrm[:, 2] * p[:, 2]).astype(F32)
                w = on[:, None]
                ata = np.where(w[:, :, None], (ata + (nrm[:, :, None] * nrm[:, None, :]).astype(F32)).astype(F32), ata)
                atb = np.where(w, (atb + (nrm * d[:, None]).astype(F32)).astype(F32), atb)
                mass = np.where(w, (mass + np.concatenate([p, np.ones((len(sel), 1), F32)], axis=1)).astype(F32), mass)
        ok = ~forced & (mass[:, 3] > 0)
        res["forced_points"] += int(forced.sum())
        if not ok.any():
            continue
        ata, atb, mass = ata[ok], atb[ok], mass[ok]
        got = lv["vert"][sel[ok], vi].astype(np.float64)
        with np.errstate(all="ignore"):
            center = (mass[:, :3] / mass[:, 3:4]).astype(F32)
            b = (atb - np.einsum("nij,nj->ni", ata, center).astype(F32)).astype(F32).astype(np.float64)
            u, s, vt = np.linalg.svd(ata.astype(np.float64))            # LAPACK, f64, singular values in descending order
            cutoff = s[:, 0] * 1e-3
            below = s < cutoff[:, None]
            rank = np.where(below.any(axis=1), below.argmax(axis=1), 3)
            eps = np.where(rank < 3, s[np.arange(len(s)), np.minimum(rank, 2)], 0.0)
            keep = s > eps[:, None]
            proj = np.einsum("nji,nj->ni", u, b)                          # u^T b
            coef = np.where(keep, proj / np.where(keep, s, 1.0), 0.0)
            sol = np.einsum("nji,nj->ni", vt, coef)                       # v coef
            want = sol + center.astype(np.float64)
        dev = np.abs(got - want).max(axis=1) / size[sel[ok]]
        devs.append(dev)
        # how close a rank decision was: the distance of any singular value to the cutoff, relative to the cutoff
        near.append((np.abs(s - cutoff[:, None]) / cutoff[:, None]).min(axis=1))
        res["vertices"] += int(ok.sum())
    dev = np.concatenate(devs) if devs else np.zeros(0)
    nr = np.concatenate(near) if near else np.zeros(0)
    res.update({"max_deviation_cell_fraction": float(dev.max()) if len(dev) else 0.0,
                "p999_deviation_cell_fraction": float(np.percentile(dev, 99.9)) if len(dev) else 0.0,
                "median_deviation_cell_fraction": float(np.median(dev)) if len(dev) else 0.0,
                "over_1e-4_of_a_cell": int((dev > 1e-4).sum()), "over_1e-2_of_a_cell": int((dev > 1e-2).sum()),
                "rank_decisions_within_1e-4_of_the_cutoff": int((nr < 1e-4).sum())})
    return res
