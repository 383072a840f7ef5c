python tools/prune2_sizes.py 2>&1 | grep prospero
timeout 600 python -m pytest tests/test_prune2.py tests/test_groups.py -m gpu -x -q 2>&1 | tail -2
bash tools/sweep_env.sh "" ""
