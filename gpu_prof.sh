cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
echo "=== bench (plain)"; timeout 600 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/bench_plain.json; cut -c1-200 gpurun_out/bench_plain.json
echo "=== rocprofv3 kernel-trace stats"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/prof/kt.log 2>&1
grep -v amdgpu.ids gpurun_out/prof/kt.log | grep '"metric"' | cut -c1-200
echo "=== pmc WRITE_SIZE"
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/pmc_w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/prof/pmc_w.log 2>&1
echo "=== pmc FETCH_SIZE"
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/pmc_r -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/prof/pmc_r.log 2>&1
echo "=== pmc SQ"
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/prof/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/prof/pmc_sq.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof/pmc_g -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/prof/pmc_g.log 2>&1
echo "=== batch workload stats"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt_batch -- python3 bench.py --workload batch --steps 5 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/prof/kt_batch.log 2>&1
grep '"metric"' gpurun_out/prof/kt_batch.log | cut -c1-200
ls gpurun_out/prof/*/*/ | head -40
