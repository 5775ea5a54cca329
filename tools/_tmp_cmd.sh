cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_traverse_gpu.py -x -q 2>&1 | tail -3
timeout 600 python tools/dev_traverse_time.py 2>/dev/null
