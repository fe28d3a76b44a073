cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "=== pytest"; timeout 1500 python -m pytest tests -m gpu -x -q --timeout=600 2>&1 | tail -3
timeout 600 python scripts/quick_perf.py hop 2>&1 | grep -v amdgpu.ids
echo "--- with pointer hints"; timeout 600 python scripts/quick_perf.py hop hint 2>&1 | grep -v amdgpu.ids
