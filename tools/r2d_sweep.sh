mkdir -p gpurun_out/r2d
for c in 2 3 4; do FHIP_SLAB_CONTEXTS=$c timeout 120 python bench.py --no-cpu > gpurun_out/r2d/bench_ctx$c.json 2>/dev/null; done
FHIP_PUSH_WAVES=4 timeout 120 python bench.py --no-cpu > gpurun_out/r2d/bench_push4.json 2>/dev/null
FHIP_AUX_STREAM=1 timeout 120 python bench.py --no-cpu > gpurun_out/r2d/bench_aux.json 2>/dev/null
timeout 120 python bench.py --no-cpu --size 2048 > gpurun_out/r2d/bench_2048.json 2>/dev/null
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2d/bench_*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['asm_kernel_ms_per_frame'].items()})
    except Exception as e: print(f, 'ERR', e)
P
