cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "=== bench N=1"; timeout 600 python bench.py --steps 10 --warmup 2 2>&1 | grep -v amdgpu.ids | tail -1
echo "=== bench 2 ranks on one GPU (gloo, functional)"; SDFT_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 2>&1 | grep -v amdgpu.ids | tail -3
echo "=== bench batch workload on 1 GPU"; timeout 600 python bench.py --workload batch --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1
