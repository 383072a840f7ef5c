#!/bin/bash
# scratch: the round's profile on the final sources, the bench line with it, then the GPU tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03y
timeout -k 5 150 bash tools/profile_round.sh r03 > gpurun_out/r03y/profile_round.log 2>&1
tail -2 gpurun_out/r03y/profile_round.log
cp gpurun_out/prof_r03/traffic.json profiles/traffic_r03.json
timeout -k 5 90 python bench.py > gpurun_out/r03y/bench.json 2> gpurun_out/r03y/bench.err
python - <<'PY'
import json
try:
    r = json.load(open("gpurun_out/r03y/bench.json"))
    print(r["value"], r["ms_per_step"], r.get("frame_latency_ms"), r["roofline"].get("traffic"), r["roofline"].get("traffic_note"))
except Exception as e:
    print("bench:", e)
PY
timeout -k 5 230 python -m pytest tests -x -q -m gpu -k "not mesh" > gpurun_out/r03y/gpu_suite.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03y/gpu_suite.log
tail -4 gpurun_out/r03y/gpu_suite.log
