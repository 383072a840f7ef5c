mkdir -p gpurun_out/r03o
python tools/p2stats.py 2>&1 | grep "level"
timeout 600 python -m pytest tests/test_prune2.py -m gpu -x -q > gpurun_out/r03o/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03o/tests.log
grep -n "passed\|failed\|rc=\|Error\|assert" gpurun_out/r03o/tests.log | tail -5
bash tools/sweep_env.sh "" "FHIP_PRUNE2_L1=0" > gpurun_out/r03o/sweep.txt 2>&1; cat gpurun_out/r03o/sweep.txt
