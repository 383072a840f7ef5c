#!/usr/bin/env python3
"""GPU box: one assembly variant (FHIP_INTERP_CO=<code object>, or the embedded one) on the general path - the leaf kernel's time per
launch (HIP events around every launch, profiled frames), queued frames per ms, and a hash of the image (must not change between
variants).  usage: tools/variant_run.py [model.vm] [size] [general|default] [stats]   (driver: tools/variants.py)"""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import fidget_amd as F

model = sys.argv[1] if len(sys.argv) > 1 else "prospero.vm"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
general = (sys.argv[3] if len(sys.argv) > 3 else "general") == "general"
want_stats = len(sys.argv) > 4 and sys.argv[4] == "stats"
stream = torch.cuda.current_stream()
hip = F.HipContext(0, stream.cuda_stream)
if not os.environ.get("VR_LANES"):       # (VR_LANES=1: the library's own arrangement of queued frames, as bench.py measures it)
    hip.set_option("frame_lanes", 0)
if general:
    hip.set_option("no_column_inv", 1)
shape = F.Shape.from_vm(os.path.join(ROOT, "models", model), hip=hip)
out = torch.zeros((n, n, 4), dtype=torch.int32, device="cuda")
for _ in range(8):            # arena growth, buffer sets
    F.render3d(shape, n, out=out)
    hip.sync()
res = {"co": os.environ.get("FHIP_INTERP_CO", "embedded"), "model": model, "n": n, "general": general}
res["sha"] = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
hip.profile(True)
kern = {}
for _ in range(4):
    F.render3d(shape, n, out=out)
    for k, (ms, cnt) in hip.profile_read_kernels().items():
        if cnt:
            kern.setdefault(k, [0.0, 0])
            kern[k][0] += ms
            kern[k][1] += cnt
    res["leaf_stats"] = hip.leaf_stats()
hip.profile(False)
res["kernel_ms_per_launch"] = {k: round(v[0] / v[1], 4) for k, v in kern.items()}
res["kernel_launches_per_frame"] = {k: v[1] / 4 for k, v in kern.items()}
K = 40 if not os.environ.get("VR_LANES") else 400
for _ in range(5 if not os.environ.get("VR_LANES") else 150):      # (the arrangement tuner's windows first)
    F.render3d(shape, n, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    F.render3d(shape, n, out=out)
torch.cuda.synchronize()
res["queued_ms_per_frame"] = round((time.perf_counter() - t0) / K * 1e3, 4)
if want_stats:
    lv = hip.last_leaves()
    regs, ln = lv["regs"].astype(int), lv["len"].astype(int)
    cls = np.where(regs <= 8, 0, np.where(regs <= 16, 1, np.where(regs <= 32, 2, 3)))
    passes = np.array([1, 2, 4, 0])[cls]
    st = {"leaves_last_slab": int(len(lv)), "len_mean": float(ln.mean()), "len_p50_p90_p99_max": [float(v) for v in np.percentile(ln, [50, 90, 99, 100])],
          "regs_mean": float(regs.mean()), "regs_hist": {str(r): int((regs == r).sum()) for r in range(0, 40) if (regs == r).any()},
          "over_64_ops": int((ln > 64).sum())}
    for c, nm in enumerate(("le8", "le16", "le32", "gt32")):
        m = cls == c
        st[nm] = {"leaves": int(m.sum()), "ops": int(ln[m].sum()), "ops_x_max_passes": int((ln[m] * passes[m]).sum()), "len_mean": float(ln[m].mean()) if m.any() else 0}
    # opcode statistics of a sample of tapes: unigrams, in-place, and adjacent pairs (with whether the second op reads the first's output)
    import collections
    rng = np.random.default_rng(1)
    pick = rng.choice(len(lv), size=min(3000, len(lv)), replace=False)
    NAMES = "OUTPUT INPUT COPY_REG COPY_IMM NEG ABS RECIP SQRT SQUARE FLOOR CEIL ROUND SIN COS TAN ASIN ACOS ATAN EXP LN NOT RAND".split()
    BIN = "ADD SUB MUL DIV ATAN2 COMPARE MIX MOD MIN MAX AND OR".split()

    def name(op):
        if op < 22: return NAMES[op]
        if op < 34: return BIN[op - 22] + "_RR"
        if op < 46: return BIN[op - 34] + "_RI"
        return ["SUB", "DIV", "ATAN2", "COMPARE", "MIX", "MOD"][op - 46] + "_IR"
    uni, inpl, pairs, tri = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter()
    tot = 0
    dump = []
    for i in pick:
        ops = hip.arena_ops(lv["off"][i], lv["len"][i])
        w0 = (ops & 0xFFFFFFFF).astype(np.uint32); w1 = (ops >> 32).astype(np.uint32)
        op = (w0 & 0xFF).astype(int); o = ((w0 >> 8) & 0xFFF).astype(int); a = (w0 >> 20).astype(int)
        if len(dump) < 40:
            dump.append([[name(int(op[k])), int(o[k]), int(a[k]), int(w1[k]) if 22 <= op[k] < 34 else -1] for k in range(len(ops))])
        desc = []
        for k in range(len(ops)):
            nm = name(int(op[k])); uni[nm] += 1; tot += 1
            ip = (op[k] > 3 and a[k] == o[k])
            if ip: inpl[nm] += 1
            desc.append(nm + ("*" if ip else ""))
        for k in range(len(ops) - 1):
            reads = (op[k + 1] not in (1, 3)) and (a[k + 1] == o[k] or (22 <= op[k + 1] < 34 and w1[k + 1] == o[k]))
            pairs[(desc[k], desc[k + 1], bool(reads))] += 1
        for k in range(len(ops) - 2):
            tri[(desc[k], desc[k + 1], desc[k + 2])] += 1
    st["ops_sampled"] = tot
    st["unigrams_pct(in place pct)"] = {k: [round(100 * v / tot, 2), round(100 * inpl[k] / v, 1)] for k, v in uni.most_common()}
    st["pairs_pct (a, b, b reads a)"] = [[list(k), round(100 * v / tot, 2)] for k, v in pairs.most_common(40)]
    st["triples_pct"] = [[list(k), round(100 * v / tot, 2)] for k, v in tri.most_common(25)]
    st["sample_tapes"] = dump
    res["stats"] = st
print(json.dumps(res))
