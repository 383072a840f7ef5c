#!/bin/bash
# GPU box: per-kernel statistics of lone frames in the few-tiles regime (tools/small_frames.py under rocprofv3 --kernel-trace --stats)
# -> gpurun_out/small_<what>_kernel_stats.csv + the lone-frame times
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for w in ${@:-3d:512 3d:256 3d:128 2d:256 octant 3d:1024}; do
  tag=$(echo $w | tr ':' '_')
  d=/tmp/small_$tag
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o k -- python $R/tools/small_frames.py $w 5 2>&1 | grep "ms per lone"
  f=$(find $d -name '*kernel_stats.csv' 2>/dev/null | head -1)
  echo "== $w"
  if [ -n "$f" ]; then cp "$f" $R/gpurun_out/small_${tag}_kernel_stats.csv; head -14 "$f" | cut -d, -f1-5; fi
done
