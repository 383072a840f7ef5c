timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -12
python tools/prune2_sizes.py 2>&1 | grep prospero | head -1
FHIP_NO_ASM_TILES_T=1 python tools/prune2_sizes.py 2>&1 | grep prospero | head -1
