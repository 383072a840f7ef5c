#!/bin/bash
# GPU call I of round 4: the side stream with level 1 alone (option side_only_l1); the mesh suite after the record-comparison fix
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04i
mkdir -p $O
cd $R
bash tools/sweep_env.sh "FHIP_SIDE_ONLY_L1=0" "FHIP_SIDE_ONLY_L1=1" "FHIP_SIDE_ONLY_L1=0" "FHIP_SIDE_ONLY_L1=1" > $O/side_only_l1.txt 2>&1; cat $O/side_only_l1.txt | cut -c1-200
timeout -k 5 900 python -m pytest tests -m gpu -q -x --timeout 600 > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
tail -4 $O/gpu_tests.log
