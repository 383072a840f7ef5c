#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04o
mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_render_random.py tests/test_gpu_math.py -m gpu -q -x --timeout 600 > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
grep -E "passed|failed|rc" $O/gpu_tests.log | tail -3
timeout -k 5 300 python tools/config_times.py > $O/config_times.log 2>&1; grep "C3\|C2" $O/config_times.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_bear; timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bear -o b -- python $R/tools/config_profile.py bear3d 20 > $O/bear_profile.log 2>&1
find /tmp/p_bear -name "*kernel_stats.csv" -exec cp {} $O/bear_kernel_stats.csv \; ; head -6 $O/bear_kernel_stats.csv | cut -c1-120
