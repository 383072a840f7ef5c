"""tools/src_hash.py ties profiles/traffic_*.json (PMC counters of the render kernels) to the sources of the render path: a change to
a render source must change the hash, a change to the mesh path (its own files, the C ABI's meshing fragment capi_mesh.hpp) must not."""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_hash_covers_the_render_path_only(tmp_path):
    import src_hash
    dst = tmp_path / "fidget_amd" / "csrc"
    shutil.copytree(os.path.join(ROOT, "fidget_amd", "csrc"), dst, ignore=shutil.ignore_patterns("_gen", "*.so", "__pycache__"))
    base = src_hash.source_hash(str(tmp_path))
    assert base == src_hash.source_hash(ROOT)

    def edited(name, find, replace):
        p = dst / name
        old = p.read_bytes()
        assert old.count(find) >= 1, (name, find)
        p.write_bytes(old.replace(find, replace, 1))
        h = src_hash.source_hash(str(tmp_path))
        p.write_bytes(old)
        return h
    # the mesh path: its own files, the C ABI's meshing fragment included
    assert edited("mesh.hip", b"k_mesh_cells", b"k_mesh_cellz") == base
    assert edited("mesh_collapse.hpp", b"OctRes", b"OctRez") == base
    assert edited("host_mesh.hpp", b"ParallelWalker", b"ParallelWalkez") == base
    assert edited("capi_mesh.hpp", b"static hipError_t mesh_assemble_device", b"static hipError_t mesh_assemble_devicf") == base
    # the render path: kernels, generators, the frame driver in capi.hip, shared headers
    assert edited("kernels.hip", b"k_classify3d", b"k_classify3e") != base
    assert edited("gen_interp.py", b"fh_columns", b"fh_columnz") != base
    assert edited("dev_ops.hpp", b"t_sin", b"t_sim") != base
    assert edited("capi_core.hpp", b"FH_ASM_COLUMNS", b"FH_ASM_COLUMNZ") != base
    assert edited("capi_render.hpp", b"upload_frame", b"upload_framf") != base
    assert b"fhip_status fhip_mesh_build(" in (dst / "capi_mesh.hpp").read_bytes()
