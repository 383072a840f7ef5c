#!/bin/bash
# GPU call C of round 4: the four-sample routines in fh_columns_t
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04c
mkdir -p $O
cd $R
timeout -k 5 400 python -m pytest tests -m gpu -q -x --timeout 300 -k "bear or math or transc or random or kat or mesh_sampl or leaf_samples" > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
tail -4 $O/gpu_tests.log
timeout -k 5 300 python tools/config_times.py > $O/config_times.log 2>&1; grep -A1 "C3\|2048\|colonnade" $O/config_times.log | head -20
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_bear; timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bear -o b -- python $R/tools/config_profile.py bear3d 20 > $O/bear_profile.log 2>&1
find /tmp/p_bear -name "*kernel_stats.csv" -exec cp {} $O/bear_kernel_stats.csv \; ; head -8 $O/bear_kernel_stats.csv
