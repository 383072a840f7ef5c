mkdir -p gpurun_out/r03j
timeout 600 python -m pytest tests/test_spills.py -m gpu -x -q > gpurun_out/r03j/spills.log 2>&1; tail -30 gpurun_out/r03j/spills.log
