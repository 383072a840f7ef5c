#!/bin/bash
# GPU box: bench.py under a list of environment settings (one per argument, "A=1 B=2" form; "" = defaults); prints, per setting,
# ms per frame of the default path (mean / median / min of the timed frames), of the general path, and both single-frame latencies
for E in "$@"; do
  R=$(env $E python bench.py --no-cpu --steps 60 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); g=d.get('general') or {}
print('default', round(d['ms_per_step'],4), round(d['ms_per_step_median'],4), round(d['ms_per_step_min'],4), 'general', round(g.get('ms_per_step',0),4), 'latency', round(d['frame_latency_ms'],4), round(g.get('frame_latency_ms',0),4), 'asm', {k:round(v,3) for k,v in d['asm_kernel_ms_per_frame'].items()})")
  echo "[$E] $R"
done
