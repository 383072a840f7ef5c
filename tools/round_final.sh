#!/bin/bash
# A round's final state on the GPU box, one call (~4.5 min): the GPU suite, the PMC passes (tools/profile_round.sh -> profiles/traffic_r04.json, which
# bench.py ties to the render sources' hash), the bench line, the other configurations, the mesh timings.  usage: gpurun -- 'bash tools/round_final.sh [tag]'
# (what the letter-named calls of round 4 ran is in DESIGN.md next to their results under profiles/r04*)
R=$GRAFT_REPO_ROOT
TAG=${1:-r04p}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests -m gpu -q -x --timeout 600 > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
tail -4 $O/gpu_tests.log
timeout -k 5 900 bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1; tail -30 $O/profile_round.log | cut -c1-200
cp gpurun_out/prof_$TAG/traffic.json profiles/traffic_r04.json 2>/dev/null
cp gpurun_out/prof_$TAG/*.csv gpurun_out/prof_$TAG/*.txt gpurun_out/prof_$TAG/*.json $O/ 2>/dev/null
export ROUND_TAG=$TAG
timeout -k 5 400 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json,os
d=json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/"+os.environ.get("ROUND_TAG","r04p")+"/bench.json")).read().strip().split("\n")[-1])
print({k:d[k] for k in ("value","ms_per_step","frame_latency_ms","host_output_frame_ms")}, d["general"]["ms_per_step"], d["general"]["frame_latency_ms"], d["c3_bear"], d["parity"], d["c5_mesh"])
print(json.dumps(d["roofline"])[:900])
PY
timeout -k 5 300 python tools/config_times.py > $O/config_times.log 2>&1; cp gpurun_out/other_configs.json $O/
timeout -k 5 600 python tools/mesh_simplify_times.py 8 > $O/mesh_simplify_times.log 2>&1; grep -v amdgpu.ids $O/mesh_simplify_times.log | cut -c1-400; cp gpurun_out/mesh_simplify_times_depth8.json $O/ 2>/dev/null
FHIP_MESH_TIMES=1 MESH_TIMES_REPS=3 timeout -k 5 200 python tools/mesh_times.py 10 > $O/mesh_times.log 2>&1; grep "fhip mesh depth 10\|build" $O/mesh_times.log | tail -4
