mkdir -p gpurun_out/r03n
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03n/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03n/tests.log
grep -n "passed\|failed\|rc=\|Error\|assert" gpurun_out/r03n/tests.log | tail -8
bash tools/sweep_env.sh "" "" > gpurun_out/r03n/sweep.txt 2>&1; cat gpurun_out/r03n/sweep.txt
