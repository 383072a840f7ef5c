for e in "FHIP_LANES_TUNE=0" ""; do
  echo "== $e"; env $e python bench.py --no-cpu --no-general --steps 200 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['ms_per_step_median'], r['frame_latency_ms'], r['device_bytes'])"
done
