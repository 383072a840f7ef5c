#!/bin/bash
# scratch: per-kernel times of one mesh build (gyroid-sphere, depth 10)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03z
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_mesh
timeout -k 3 42 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_mesh -o m -- python $R/tools/mesh_times.py 10 > $R/gpurun_out/r03z/mesh_under_rocprof.log 2>&1
find /tmp/p_mesh -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r03z/mesh_kernel_stats.csv \;
head -14 $R/gpurun_out/r03z/mesh_kernel_stats.csv
