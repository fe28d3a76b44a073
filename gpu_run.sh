cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "=== pytest"; timeout 1500 python -m pytest tests -m gpu -x -q --timeout=600 2>&1 | tail -3
timeout 600 python scripts/quick_perf.py fft 2>&1 | grep -v amdgpu.ids | head -3 | sed 's/.*opts=/opts=/'
