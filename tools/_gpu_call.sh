mkdir -p gpurun_out/r03m
FHIP_PRUNE2=1 python tools/level_stats.py 1024 > /dev/null 2>&1; cp gpurun_out/level_stats_prospero.vm_1024.json gpurun_out/r03m/level_stats_prune2.json
python tools/level_stats.py 1024 > /dev/null 2>&1; cp gpurun_out/level_stats_prospero.vm_1024.json gpurun_out/r03m/level_stats_prune1.json
python - <<'PY'
import json
for t in ("prune1","prune2"):
    d=json.load(open(f"gpurun_out/r03m/level_stats_{t}.json"))
    for k in ("level1_parents","level2_parents_all_slabs","leaves_last_slab"):
        e=d[k]; print(t,k,e["n"],e.get("len_sum"),e.get("len_p50_90_99_max"),e.get("regs_p50_90_99_max"),e.get("choices_p50_90_99_max"))
PY
