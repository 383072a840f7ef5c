mkdir -p gpurun_out/r03z
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03z/gpu_suite.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03z/gpu_suite.log
grep -n "passed\|failed\|rc=" gpurun_out/r03z/gpu_suite.log | tail -3
python -c "
import __graft_entry__ as g
g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/profile_round.sh r03 > gpurun_out/r03z/profile_round.log 2>&1
python bench.py > gpurun_out/r03z/bench.json 2> gpurun_out/r03z/bench.err; tail -c 150 gpurun_out/r03z/bench.json
python tools/config_times.py > gpurun_out/r03z/config_times.log 2>&1
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pb; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o s -- python $R/tools/config_profile.py bear3d 20 > $R/gpurun_out/r03z/bear_run.log 2>&1; find /tmp/pb -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r03z/bear_kernel_stats.csv \;
