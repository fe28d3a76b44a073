cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "=== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q --timeout=600 2>&1 | tail -6
echo "=== perf"; timeout 900 python scripts/quick_perf.py fft 2>&1 | grep -v amdgpu.ids
