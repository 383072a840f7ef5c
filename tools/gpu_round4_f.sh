#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04f
mkdir -p $O
cd $R
timeout -k 5 300 python tools/cu_mask.py 100 > $O/cu_mask.txt 2>&1; cat $O/cu_mask.txt | grep -v amdgpu.ids
