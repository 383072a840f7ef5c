import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fidget_amd as F, oracle as O
m, n = sys.argv[1], int(sys.argv[2])
ref = O.render3d(O.Shape.from_vm("models/" + m), n)[0]
sh = F.Shape.from_vm("models/" + m)
for r in range(3):
    got = F.render3d(sh, n)[0]
    badm = (got["depth"] != ref["depth"]) | (got["normal"].view(np.uint32) != ref["normal"].view(np.uint32)).any(axis=-1)
    ys, xs = np.nonzero(badm)
    print(m, n, "run", r, "bad pixels", badm.sum(), "depth-bad", (got["depth"] != ref["depth"]).sum())
    for y, x in list(zip(ys, xs))[:5]:
        print("   ", y, x, "got", got["depth"][y, x], got["normal"][y, x], "want", ref["depth"][y, x], ref["normal"][y, x])
