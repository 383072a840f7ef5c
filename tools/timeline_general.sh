#!/bin/bash
# GPU box: per-stream timeline of queued GENERAL-path frames (bench.py --only-general --no-cpu under rocprofv3 --kernel-trace, tools/timeline.py)
# -> gpurun_out/timeline_general_<tag>.txt
TAG=${1:-x}
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tlg
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tlg -o t -- python $R/bench.py --steps 40 --warmup 5 --no-cpu --only-general > /tmp/tlg.log 2>&1
python $R/tools/timeline.py /tmp/tlg 2 +80 > $R/gpurun_out/timeline_general_$TAG.txt 2>&1
exit 0
