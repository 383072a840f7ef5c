#!/bin/bash
# GPU call J: timeline of pipelined frames (steady state) under rocprofv3 --kernel-trace
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04j
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_tl; timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tl -o t -- python $R/bench.py --steps 30 --warmup 5 --no-cpu --no-general > $O/run.log 2>&1
python $R/tools/timeline.py /tmp/p_tl 2 +20 > $O/timeline_pipelined.txt 2>&1; cat $O/timeline_pipelined.txt | head -150
