#!/usr/bin/env python3
"""GPU box: find where the 3D normals of the random shapes (tests/test_render_random.py) leave the oracle's.
For every differing pixel: the render's normal on both sides, the trait-level gradient of the FULL tape at the
same point on both sides, and the first sub-expression (construction order) whose gradient differs."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fidget_amd as F
import oracle as O
from test_render_random import build


class Rec:
    """records every node a Context constructor returns"""
    def __init__(self, ctx):
        self.ctx, self.log = ctx, []
    def __getattr__(self, name):
        f = getattr(self.ctx, name)
        def g(*a):
            r = f(*a)
            self.log.append((name, a, r))
            return r
        return g


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def main():
    out = []
    sizes = [int(s) for s in (sys.argv[1:] or ["64", "128", "200"])]
    for seed in range(12):
        for size in sizes:
            fr, orr = Rec(F.Context()), Rec(O.Context())
            fn, on = build(fr, seed), build(orr, seed)
            a = F.render3d(F.Shape(fr.ctx, fn), size)[0]
            b = O.render3d(O.Shape(orr.ctx, on), size)[0]
            dn = (bits(a["normal"]) != bits(b["normal"])).any(axis=2) & ~(np.isnan(a["normal"]).any(axis=2) & np.isnan(b["normal"]).any(axis=2))
            ys, xs = np.nonzero(dn)
            rec = {"seed": seed, "size": size, "depth_diff": int((a["depth"] != b["depth"]).sum()), "normal_diff_pixels": int(dn.sum()),
                   "max_abs": float(np.nanmax(np.abs(a["normal"] - b["normal"]))) if dn.any() else 0.0, "pixels": []}
            mat = F.screen_to_world([size, size, size])
            for y, x in list(zip(ys, xs))[:3]:
                d = int(a["depth"][y, x])
                p = F.transform_point(mat, float(x), float(y), float(d - 1))
                px = {"xy": [int(x), int(y)], "depth": d, "hip": a["normal"][y, x].tolist(), "oracle": b["normal"][y, x].tolist(), "model_point": p.tolist()}
                gf = F.Shape(fr.ctx, fn).eval_grad_slice([p[0]], [p[1]], [p[2]])[0]
                go = O.Shape(orr.ctx, on).eval_grad_slice([p[0]], [p[1]], [p[2]])[0]
                px["full_tape_hip"], px["full_tape_oracle"] = gf.tolist(), go.tolist()
                px["full_tape_equal"] = bool((bits(gf) == bits(go)).all())
                px["render_equals_full_tape_hip"] = bool((bits(gf[1:]) == bits(a["normal"][y, x])).all())
                px["render_equals_full_tape_oracle"] = bool((bits(go[1:]) == bits(b["normal"][y, x])).all())
                # first differing sub-expression
                for (name, args, nf), (_, _, no) in zip(fr.log, orr.log):
                    if name in ("x", "y", "z", "constant"):
                        continue
                    try:
                        sf = F.Shape(fr.ctx, nf).eval_grad_slice([p[0]], [p[1]], [p[2]])[0]
                        so = O.Shape(orr.ctx, no).eval_grad_slice([p[0]], [p[1]], [p[2]])[0]
                    except Exception as e:   # constants fold to nodes that are not ops
                        continue
                    if not (bits(sf) == bits(so)).all():
                        px["first_diff"] = {"op": name, "args": [float(v) if isinstance(v, float) else int(v) for v in args],
                                            "hip": sf.tolist(), "oracle": so.tolist()}
                        # the operands' values on the device
                        ops = []
                        for v in args:
                            if isinstance(v, float):
                                ops.append(v)
                            else:
                                try:
                                    ops.append(F.Shape(fr.ctx, v).eval_grad_slice([p[0]], [p[1]], [p[2]])[0].tolist())
                                except Exception:
                                    ops.append(None)
                        px["first_diff"]["operands_hip"] = ops
                        break
                rec["pixels"].append(px)
            out.append(rec)
            print(json.dumps(rec), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bisect_normals.json"), "w"), indent=1)


main()
