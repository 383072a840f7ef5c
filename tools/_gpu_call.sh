mkdir -p gpurun_out/r03k
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03k/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03k/tests.log
grep -n "passed\|failed\|rc=" gpurun_out/r03k/tests.log | tail -3
bash tools/sweep_env.sh "" "" > gpurun_out/r03k/sweep.txt 2>&1; cat gpurun_out/r03k/sweep.txt
