timeout 600 python -m pytest tests/test_prune2.py tests/test_groups.py -m gpu -x -q 2>&1 | tail -3
python tools/p2stats.py 2>&1 | grep "level 0"
bash tools/sweep_env.sh "" ""
