#!/bin/bash
# GPU call K: issue priority for the coarse chain's kernels (option chain_prio)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04k
mkdir -p $O
cd $R
bash tools/sweep_env.sh "" "FHIP_FRAME_SETS=3 FHIP_TAIL_ON_MAIN=0" "" "FHIP_V64_WAVES=2" > $O/new_defaults.txt 2>&1; cat $O/new_defaults.txt | cut -c1-200
