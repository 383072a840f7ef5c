mkdir -p gpurun_out/r03i
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03i/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03i/tests.log
grep -n "passed\|failed\|rc=" gpurun_out/r03i/tests.log | tail -3
timeout 600 python bench.py > gpurun_out/r03i/bench.json 2> gpurun_out/r03i/bench.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r03i/bench.json"))
g=d["general"]
print("value", round(d["value"]), "ms", round(d["ms_per_step"],4), "median", round(d["ms_per_step_median"],4), "lat", round(d["frame_latency_ms"],3), "host", round(d["host_output_frame_ms"],3))
print("general ms", round(g["ms_per_step"],4), "lat", round(g["frame_latency_ms"],3), "img eq", g["image_equals_default_path"])
r=d["roofline"]; print("roofline", r["kernel"], r["path"], "frac", round(r["frac"],4), "alu", round(r["alu"]["frac"],4), "avg_launch_ms", round(r["avg_launch_ms"],4), r["launches_per_frame"])
r=d["roofline_tiles"]; print("tiles", r["kernel"], "frac", round(r["frac"],4), "avg_launch_ms", round(r["avg_launch_ms"],4), r["launches_per_frame"])
print("parity", d["parity"], "cpu", round(d["cpu_baseline"]["value"]), "c3", d["c3_bear"]["ms_per_frame"], d["c3_bear"]["normal_max_ulp_of_gradient_scale"])
P
