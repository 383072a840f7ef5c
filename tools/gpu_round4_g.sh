#!/bin/bash
# GPU call G of round 4: tape simplification down the octree in the mesher; corners through the bulk interpreter
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04g
mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests/test_mesh.py tests/test_mesh_assembly.py tests/test_multi_gpu.py -m gpu -q -x --timeout 600 > $O/mesh_tests.log 2>&1; echo "pytest rc $?" >> $O/mesh_tests.log
tail -6 $O/mesh_tests.log
timeout -k 5 600 python tools/mesh_simplify_times.py 8 > $O/mesh_simplify_times.log 2>&1; grep -v amdgpu.ids $O/mesh_simplify_times.log | cut -c1-600
cp gpurun_out/mesh_simplify_times_depth8.json $O/ 2>/dev/null
FHIP_MESH_TIMES=1 MESH_TIMES_REPS=3 timeout -k 5 200 python tools/mesh_times.py 10 > $O/mesh_times.log 2>&1; grep "fhip mesh depth 10\|build" $O/mesh_times.log | tail -6
