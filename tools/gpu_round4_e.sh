#!/bin/bash
# GPU call E of round 4: frame sets 3 / 4 / 5 (a set is free when its frame is complete: period >= latency / sets)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04e
mkdir -p $O
cd $R
bash tools/sweep_env.sh "" "FHIP_FRAME_SETS=4" "FHIP_FRAME_SETS=5" "FHIP_FRAME_SETS=4 FHIP_SLAB_CONTEXTS=2" > $O/frame_sets.txt 2>&1
cat $O/frame_sets.txt
