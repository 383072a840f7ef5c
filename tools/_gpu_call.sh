mkdir -p gpurun_out/r03v
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03v/gpu_suite.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03v/gpu_suite.log
grep -n "passed\|failed\|rc=\|Error\|assert" gpurun_out/r03v/gpu_suite.log | tail -6
bash tools/sweep_env.sh "" "FHIP_NO_ASM_NORMALS=1" "" "FHIP_NORMALS_WAVES=16"
