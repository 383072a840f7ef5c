#!/bin/bash
# GPU call H of round 4: the round's final state - suite, PMC passes (traffic_r04.json), bench line, other configurations, mesh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04n
mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests -m gpu -q -x --timeout 600 > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
tail -4 $O/gpu_tests.log
timeout -k 5 900 bash tools/profile_round.sh r04n > $O/profile_round.log 2>&1; tail -30 $O/profile_round.log | cut -c1-200
cp gpurun_out/prof_r04n/traffic.json profiles/traffic_r04.json 2>/dev/null
cp gpurun_out/prof_r04n/*.csv gpurun_out/prof_r04n/*.txt gpurun_out/prof_r04n/*.json $O/ 2>/dev/null
timeout -k 5 400 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json,os
d=json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r04n/bench.json")).read().strip().split("\n")[-1])
print({k:d[k] for k in ("value","ms_per_step","frame_latency_ms","host_output_frame_ms")}, d["general"]["ms_per_step"], d["general"]["frame_latency_ms"], d["c3_bear"], d["parity"], d["c5_mesh"])
print(json.dumps(d["roofline"])[:900])
PY
timeout -k 5 300 python tools/config_times.py > $O/config_times.log 2>&1; cp gpurun_out/other_configs.json $O/
timeout -k 5 600 python tools/mesh_simplify_times.py 8 > $O/mesh_simplify_times.log 2>&1; grep -v amdgpu.ids $O/mesh_simplify_times.log | cut -c1-400; cp gpurun_out/mesh_simplify_times_depth8.json $O/ 2>/dev/null
FHIP_MESH_TIMES=1 MESH_TIMES_REPS=3 timeout -k 5 200 python tools/mesh_times.py 10 > $O/mesh_times.log 2>&1; grep "fhip mesh depth 10\|build" $O/mesh_times.log | tail -4
