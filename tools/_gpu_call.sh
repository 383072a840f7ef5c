bash tools/sweep_env.sh "FHIP_GROUPS=16" "FHIP_GROUPS=24" "FHIP_GROUPS=32"
FHIP_GROUPS=32 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "headline or column_invariant" 2>&1 | tail -2
