cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_api.py -m gpu -x -q --timeout=300 2>&1 | tail -15
