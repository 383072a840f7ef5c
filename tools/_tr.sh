for i in 1 2; do timeout 100 python bench.py --steps 40 --warmup 3 --no-cpu 2>/dev/null | python -c "
import json,sys,os;d=json.loads(sys.stdin.read());print(d['ms_per_step'], d['kernel_ms_per_frame'])"; done
