#!/usr/bin/env python3
"""Hash of the RENDER path's device-side sources: ties profiles/traffic_*.json - PMC counters of the render kernels - to the build they
were measured on.  Everything under fidget_amd/csrc counts (generators of the assembly kernels, HIP kernels, the fragments of the C ABI
with the frame driver, headers) except the files only the mesh path uses (MESH_ONLY: capi_mesh.hpp is the C ABI's meshing fragment).  A
change there cannot alter a render kernel or how a frame is launched."""
import glob, hashlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MESH_ONLY = ("mesh.hip", "mesh_collapse.hpp", "mesh_qef.hpp", "mesh_edges.hpp", "mesh_walk.hpp", "host_mesh.hpp", "capi_mesh.hpp")


def source_hash(root=ROOT):
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(root, "fidget_amd", "csrc", "*"))):
        name = os.path.basename(f)
        if os.path.isfile(f) and name.rsplit(".", 1)[-1] in ("py", "hip", "hpp", "h", "cpp") and name not in MESH_ONLY:
            h.update(name.encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
