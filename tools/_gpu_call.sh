timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_render_random.py tests/test_spills.py -m gpu -x -q 2>&1 | tail -3
bash tools/sweep_env.sh "" ""
python tools/prune2_sizes.py 2>&1 | grep prospero | head -1
