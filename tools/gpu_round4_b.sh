#!/bin/bash
# GPU call B of round 4: the frame's head as one launch (k_frame_begin): suite, bench line, one-frame timeline; two contexts; bear per kernel
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
timeout -k 5 600 python -m pytest tests -m gpu -q -x --timeout 300 > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
tail -4 $O/gpu_tests.log
timeout -k 5 400 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json,os
d=json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r04b/bench.json")).read().strip().split("\n")[-1])
print({k:d[k] for k in ("value","ms_per_step","frame_latency_ms","host_output_frame_ms")}, d["general"]["ms_per_step"], d["general"]["frame_latency_ms"], d["c3_bear"]["ms_per_frame"], d["parity"])
PY
timeout -k 5 200 python tools/two_contexts.py > $O/two_contexts.txt 2>&1; cat $O/two_contexts.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_one; timeout -k 5 120 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/p_one -o t -- python $R/tools/one_frame.py > $O/one_frame.log 2>&1
python $R/tools/timeline.py /tmp/p_one 1 1 > $O/timeline_one_frame.txt 2>&1; head -50 $O/timeline_one_frame.txt
rm -rf /tmp/p_bear; timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bear -o b -- python $R/tools/config_profile.py bear3d 20 > $O/bear_profile.log 2>&1
find /tmp/p_bear -name "*kernel_stats.csv" -exec cp {} $O/bear_kernel_stats.csv \; ; head -12 $O/bear_kernel_stats.csv
