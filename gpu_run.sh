cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q --timeout=600 2>&1 | tail -3
