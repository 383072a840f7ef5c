#!/bin/bash
# GPU box: timeline of queued default-path frames (bench.py --no-general --no-cpu under rocprofv3 --kernel-trace) -> gpurun_out/timeline_<tag>.txt,
# and of frames rendered one at a time (tools/one_frame.py)
TAG=${1:-x}
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tl -o t -- python $R/bench.py --steps 40 --warmup 5 --no-cpu --no-general > /tmp/tl.log 2>&1
python $R/tools/timeline.py /tmp/tl 3 +9 > $R/gpurun_out/timeline_$TAG.txt 2>&1
find /tmp/tl -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/kernel_stats_$TAG.csv \;
exit 0
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl1 -o t -- python $R/tools/one_frame.py > /tmp/tl1.log 2>&1
python $R/tools/timeline.py /tmp/tl1 1 1 > $R/gpurun_out/timeline_one_$TAG.txt 2>&1
tail -3 /tmp/tl.log | cut -c1-300
