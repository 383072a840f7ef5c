for e in "FHIP_NO_ROOT_SPLIT=1" "FHIP_NO_ROOT_SPLIT=2" "FHIP_NO_ROOT_SPLIT=3" "FHIP_NO_ROOT_SPLIT=0" "FHIP_NO_ROOT_SPLIT=2 FHIP_TAIL_ON_MAIN=0" "FHIP_NO_ROOT_SPLIT=3 FHIP_TAIL_ON_MAIN=0" "FHIP_NO_ROOT_SPLIT=1 FHIP_FRAME_SETS=5" "FHIP_NO_ROOT_SPLIT=2 FHIP_FRAME_SETS=5"; do
  echo "== $e"; env $e FHIP_LANES_TUNE=0 python bench.py --no-cpu --no-general --steps 200 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['ms_per_step_median'], r['frame_latency_ms'], r['device_bytes'])"
done
