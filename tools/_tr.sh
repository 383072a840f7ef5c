timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -5
for i in 1 2 3; do timeout 100 python bench.py --steps 40 --warmup 3 --no-cpu 2>/dev/null | python -c "
import json,sys,os;d=json.loads(sys.stdin.read());print(d['ms_per_step'], d['kernel_ms_per_frame'])"; done
FHIP_NO_AUX_STREAM=1 timeout 100 python bench.py --steps 40 --warmup 3 --no-cpu 2>/dev/null | python -c "
import json,sys,os;d=json.loads(sys.stdin.read());print('noaux', d['ms_per_step'])"
