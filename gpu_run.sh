cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "=== pytest"; timeout 1500 python -m pytest tests -m gpu -x -q --timeout=600 2>&1 | tail -3
echo "=== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "=== bench"; timeout 600 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-1700
