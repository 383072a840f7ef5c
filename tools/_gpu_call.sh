#!/bin/bash
# scratch: the GPU tests outside the mesh path on the final sources
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03z
timeout -k 5 100 python -m pytest tests -x -q -m gpu -k "not mesh" > gpurun_out/r03z/gpu_suite_not_mesh.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03z/gpu_suite_not_mesh.log
grep -E "passed|failed|rc=" gpurun_out/r03z/gpu_suite_not_mesh.log | tail -3
