mkdir -p gpurun_out/r03y
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03y/gpu_suite.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03y/gpu_suite.log
grep -n "passed\|failed\|rc=" gpurun_out/r03y/gpu_suite.log | tail -3
python -c "
import __graft_entry__ as g
g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/profile_round.sh r03 > gpurun_out/r03y/profile_round.log 2>&1
python bench.py > gpurun_out/r03y/bench.json 2> gpurun_out/r03y/bench.err; tail -c 150 gpurun_out/r03y/bench.json
