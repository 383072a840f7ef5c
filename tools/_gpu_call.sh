timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bear or gyroid or transc" 2>&1 | tail -2
python tools/prune2_sizes.py 2>&1 | grep prospero | head -1
