#!/bin/bash
# GPU call K: issue priority for the coarse chain's kernels (option chain_prio)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04k
mkdir -p $O
cd $R
bash tools/sweep_env.sh "" "FHIP_DEBUG_ZFILL=1" "FHIP_COL_WAVES=4" "FHIP_COL_WAVES=8" "FHIP_NORMALS_WAVES=2" > $O/interference.txt 2>&1; cat $O/interference.txt | cut -c1-200
