python bench.py --no-cpu --steps 100 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('default', r['ms_per_step'], r['ms_per_step_median'], r['frame_latency_ms'], 'general', r['config']['general_path'], 'fh_columns general launch ms', r['roofline']['avg_launch_ms'], 'device_bytes', r['device_bytes'])"
