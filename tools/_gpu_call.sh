#!/bin/bash
# scratch: the mesh path with the octree assembled on the device
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03y
timeout 900 python -m pytest tests/test_mesh.py tests/test_multi_gpu.py -x -q -m gpu -k "mesh" > gpurun_out/r03y/mesh_tests.log 2>&1
tail -5 gpurun_out/r03y/mesh_tests.log
timeout 600 python tools/mesh_times.py 9 10 > gpurun_out/r03y/mesh_times.log 2>&1
grep "fhip mesh depth\|^10\|^9" gpurun_out/r03y/mesh_times.log
cp gpurun_out/mesh_times.json gpurun_out/r03y/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03y/prof_mesh -o mesh -- python $GRAFT_REPO_ROOT/tools/mesh_times.py 10 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/r03y/prof_mesh -name "*kernel_stats.csv" | head -1); head -12 $f
