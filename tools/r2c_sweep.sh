mkdir -p gpurun_out/r2c
timeout 120 python tools/tiles_probe.py > gpurun_out/r2c/probe.json 2> gpurun_out/r2c/probe.err
for w in 4 8 12 16; do FHIP_V32_WAVES=$w timeout 120 python bench.py --no-cpu > gpurun_out/r2c/bench_v32w$w.json 2>/dev/null; done
for w in 2 4 8; do FHIP_V64_WAVES=$w timeout 120 python bench.py --no-cpu > gpurun_out/r2c/bench_v64w$w.json 2>/dev/null; done
FHIP_NO_PIPELINE=1 timeout 120 python bench.py --no-cpu > gpurun_out/r2c/bench_nopipe.json 2>/dev/null
cat gpurun_out/r2c/probe.json
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2c/bench_*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['asm_kernel_ms_per_frame'].items()})
    except Exception as e: print(f, 'ERR', e)
P
