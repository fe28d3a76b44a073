cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q --timeout=600 2>&1 | tail -12 | cut -c1-600
