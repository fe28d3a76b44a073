cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "=== pytest"; timeout 1500 python -m pytest tests -m gpu -x -q --timeout=600 2>&1 | tail -3
echo "=== bench all cores"; timeout 900 python bench.py --cpu-all-cores 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/bench_allcores.json; python -c "
import json; d=json.load(open('gpurun_out/bench_allcores.json')); print(d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], d['extras'].get('cpu_baseline_all_cores'))"
