mkdir -p gpurun_out/r03x
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03x/gpu_suite.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03x/gpu_suite.log
grep -n "passed\|failed\|rc=" gpurun_out/r03x/gpu_suite.log | tail -3
python -c "
import __graft_entry__ as g
g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/profile_round.sh r03 > gpurun_out/r03x/profile_round.log 2>&1
python bench.py > gpurun_out/r03x/bench.json 2> gpurun_out/r03x/bench.err; tail -c 200 gpurun_out/r03x/bench.json
python tools/config_times.py > gpurun_out/r03x/config_times.log 2>&1; tail -3 gpurun_out/r03x/config_times.log
bash tools/pmc_stalls.sh general > gpurun_out/r03x/pmc_stalls_general.log 2>&1
