#!/usr/bin/env python3
"""Hash of the RENDER path's device-side sources: ties profiles/traffic_*.json - PMC counters of the render kernels - to the build they
were measured on.  Everything under fidget_amd/csrc counts (generators of the assembly kernels, HIP kernels, the C ABI with the frame
driver, headers) except what only the mesh path uses: the files in MESH_ONLY and the part of capi.hip between the lines that open its
meshing and its profiling sections.  A change there cannot alter a render kernel or how a frame is launched."""
import glob, hashlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MESH_ONLY = ("mesh.hip", "mesh_collapse.hpp", "mesh_qef.hpp", "host_mesh.hpp")
MESH_BEGIN, MESH_END = b"// ---- meshing: ", b"// ---- profiling "


def render_part(name, data):
    if name != "capi.hip":
        return data
    i, j = data.find(MESH_BEGIN), data.find(MESH_END)
    assert 0 < i < j, "capi.hip: the section markers the source hash cuts at are gone"
    return data[:i] + data[j:]


def source_hash(root=ROOT):
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(root, "fidget_amd", "csrc", "*"))):
        name = os.path.basename(f)
        if os.path.isfile(f) and name.rsplit(".", 1)[-1] in ("py", "hip", "hpp", "h", "cpp") and name not in MESH_ONLY:
            h.update(name.encode())
            h.update(render_part(name, open(f, "rb").read()))
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
