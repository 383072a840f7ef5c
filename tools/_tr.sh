timeout 900 python -m pytest tests/test_groups.py tests/test_render_random.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2; do timeout 100 python bench.py --steps 40 --warmup 3 --no-cpu 2>/dev/null | python -c "
import json,sys,os;d=json.loads(sys.stdin.read());print(d['ms_per_step'], d['kernel_ms_per_frame'])"; done
