#!/bin/bash
# GPU call D of round 4: the dual walk on the device
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04d
mkdir -p $O
cd $R
export FHIP_MESH_TIMES=1
timeout -k 5 600 python -m pytest tests/test_mesh.py tests/test_mesh_assembly.py tests/test_multi_gpu.py -m gpu -q -x --timeout 300 > $O/mesh_tests.log 2>&1; echo "pytest rc $?" >> $O/mesh_tests.log
grep -v "^fhip" $O/mesh_tests.log | tail -6
MESH_TIMES_REPS=4 timeout -k 5 200 python tools/mesh_times.py 10 > $O/mesh_times.log 2>&1; grep "fhip mesh depth 10\|build" $O/mesh_times.log | tail -8
FHIP_MESH_DEVICE_WALK=0 MESH_TIMES_REPS=2 timeout -k 5 200 python tools/mesh_times.py 10 > $O/mesh_times_host_walk.log 2>&1; grep "fhip mesh depth 10" $O/mesh_times_host_walk.log | tail -2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_mesh; MESH_TIMES_REPS=2 timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_mesh -o m -- python $R/tools/mesh_times.py 10 > $O/mesh_under_rocprof.log 2>&1
find /tmp/p_mesh -name "*kernel_stats.csv" -exec cp {} $O/mesh_kernel_stats.csv \; ; head -24 $O/mesh_kernel_stats.csv | cut -c1-150
