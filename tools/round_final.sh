#!/bin/bash
# A round's final state on the GPU box, one call (~5 min): the GPU suite, the PMC passes (tools/profile_round.sh -> profiles/traffic_r06.json, which
# bench.py ties to the render sources' hash), the bench line, the other configurations, the mesh timings.  usage: gpurun -- 'bash tools/round_final.sh [tag]'
R=$GRAFT_REPO_ROOT
TAG=${1:-r06z}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 5 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 5 900 python -m pytest tests -m gpu -q -x -n 2 --timeout 600 > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/gpu_tests.log | tail -4
timeout -k 5 900 bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1; tail -30 $O/profile_round.log | cut -c1-200
cp gpurun_out/prof_$TAG/traffic.json profiles/traffic_r06.json 2>/dev/null
cp gpurun_out/prof_$TAG/*.csv gpurun_out/prof_$TAG/*.txt gpurun_out/prof_$TAG/*.json $O/ 2>/dev/null
timeout -k 5 600 python bench.py > $O/bench.json 2> $O/bench.err; cp gpurun_out/bench_details_n1.json $O/ 2>/dev/null; cat $O/bench.json
timeout -k 5 300 python tools/config_times.py > $O/config_times.log 2>&1; cp gpurun_out/other_configs.json $O/; tail -12 $O/config_times.log | cut -c1-300
# the leaf kernel's per-launch time by the profiler and by the library's events over the SAME steady frames (profiled mode, frames alone)
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ss && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ss -o s -- python $R/tools/steady_stats.py prospero.vm 1024 general 30 > $O/steady_stats_general.log 2>&1; find /tmp/ss -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_general_steady.csv \;)
grep "HIP-event" $O/steady_stats_general.log | cut -c1-300; grep '"fh_columns"' $O/kernel_stats_general_steady.csv
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sb && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sb -o s -- python $R/tools/config_profile.py bear3d 60 > /dev/null 2>&1; find /tmp/sb -name "*kernel_stats.csv" -exec cp {} $O/bear_512_kernel_stats.csv \;)
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sm && MESH_TIMES_REPS=3 timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sm -o s -- python $R/tools/mesh_times.py 10 > $O/mesh_times_under_rocprof.log 2>&1; find /tmp/sm -name "*kernel_stats.csv" -exec cp {} $O/mesh_kernel_stats.csv \;)
head -8 $O/mesh_kernel_stats.csv | cut -c1-160
bash tools/fetch_calib.sh > /dev/null 2>&1; cp gpurun_out/fetch_calib.txt $O/ 2>/dev/null
timeout -k 5 100 python tools/small_2d.py > $O/small_2d.log 2>&1; grep "2D" $O/small_2d.log
timeout -k 5 600 python tools/mesh_simplify_times.py 8 > $O/mesh_simplify_times.log 2>&1; grep -v amdgpu.ids $O/mesh_simplify_times.log | cut -c1-400; cp gpurun_out/mesh_simplify_times_depth8.json $O/ 2>/dev/null
