"""bench.py's multi-rank branch (one process per GPU, channels sharded over ranks, barrier +
max-over-ranks timing) run for real: two ranks started by torch.distributed.run as a child process.
With two GPUs the backend is nccl (= RCCL); on a one-GPU box both ranks share the GPU over gloo."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(nproc, *extra, env_extra=None):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                 # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_two_rank_bench_shards_channels():
    import torch
    backend = {} if torch.cuda.device_count() >= 2 else {"SDFT_BENCH_BACKEND": "gloo"}
    res = run_bench(2, "--steps", "2", "--warmup", "1", "--samples", "4096", "--no-extras", env_extra=backend)
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["warmup"] == 1
    assert res["config"]["channels_total"] == 128 and res["config"]["samples_per_channel"] == 4096
    assert res["scaling"] == "weak" and res["value"] > 0 and res["ms_per_step"] > 0
    assert res["roofline"]["bound"] == "hbm" and 0 < res["roofline"]["frac"] < 1
    # value = all ranks' samples / max-over-ranks time
    assert abs(res["value"] - 128 * 4096 / (res["ms_per_step"] * 1e-3) / 1e6) <= 0.02 * res["value"]


def test_single_rank_bench_line_contract():
    res = run_bench(1, "--steps", "2", "--warmup", "1", "--samples", "65536", "--cpu-samples", "8192")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in res, key
    assert res["n_gpus"] == 1 and res["vs_baseline"] is None and res["dtype"] == "f64" and res["data"] == "synthetic"
    assert res["cpu_baseline"]["cores"] == 1 and res["cpu_baseline"]["value"] > 0
    assert res["analysis_plus_synthesis_msamples_s"] > 0 and res["analysis_plus_synthesis_msamples_s"] < res["value"]
