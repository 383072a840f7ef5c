import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fidget_amd as F, oracle as O
m, n = sys.argv[1], int(sys.argv[2])
ref = O.render3d(O.Shape.from_vm("models/" + m), n)[0]
got = F.render3d(F.Shape.from_vm("models/" + m), n)[0]
bad = np.argwhere((got["depth"] != ref["depth"]) | (got["normal"].view(np.uint32) != ref["normal"].view(np.uint32)).any(axis=-1))
print(m, n, "asm_tiles" if not os.environ.get("FHIP_NO_ASM_TILES") else "c++", "bad pixels", len(bad))
for y, x in bad[:6]:
    print("  ", y, x, "got", got["depth"][y, x], got["normal"][y, x], "want", ref["depth"][y, x], ref["normal"][y, x])
