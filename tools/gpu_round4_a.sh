#!/bin/bash
# GPU call A of round 4: the whole GPU suite on the libm-identical routines, the 2^32 sweeps, the bench line, mesh times, other configs
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04a
mkdir -p $O
cd $R
export FHIP_MESH_TIMES=1
timeout -k 5 600 python -m pytest tests -m gpu -q -x --timeout 300 > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
tail -5 $O/gpu_tests.log
timeout -k 5 400 python tools/math_sweep.py > $O/math_sweep.log 2>&1; cp gpurun_out/math_sweep.json $O/ 2>/dev/null
tail -10 $O/math_sweep.log
timeout -k 5 400 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
MESH_TIMES_REPS=3 timeout -k 5 200 python tools/mesh_times.py 10 > $O/mesh_times.log 2>&1; tail -8 $O/mesh_times.log
timeout -k 5 300 python tools/config_times.py > $O/config_times.log 2>&1; tail -12 $O/config_times.log
