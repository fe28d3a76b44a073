"""Many-core CPU baseline: one independent channel (= one oracle plan) per worker process, the
fair CPU counterpart of the GPU's multi-channel configs.  TEST/BENCH
INFRASTRUCTURE ONLY -- used by bench.py's optional `--cpu-all-cores` leg.

    python -m oracle.cpu_bench --procs 32 --n 16384 --m 1024 --combo f32f64 --window hann

Prints one JSON line: aggregate Msamples/s over all workers (samples of all workers / the slowest
worker's best-of-`reps` time).
"""

import argparse
import json
import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def _worker(args):
    idx, n, m, window, combo, reps, cores = args
    import numpy as np
    from oracle import oracle as O
    from sdft_amd.signals import sine_sweep
    try:
        os.sched_setaffinity(0, {cores[idx % len(cores)]})
    except Exception:
        pass
    td, fd, fdx = O.combo_types(combo)
    x = sine_sweep(n, channel=idx, channels=max(len(cores), 1), dtype=td)
    plan = O.best(m, window, 1.0, combo)
    out = np.zeros((n, m), dtype=fdx)                  # pre-touched
    best = float("inf")
    for _ in range(reps):
        plan.reset()
        t0 = time.perf_counter()
        plan.sdft(x, out)
        best = min(best, time.perf_counter() - t0)
    return best, plan.kind


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=0)
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--m", type=int, default=1024)
    ap.add_argument("--window", default="hann")
    ap.add_argument("--combo", default="f32f64")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    cores = sorted(os.sched_getaffinity(0))
    procs = a.procs or len(cores)
    with mp.get_context("fork").Pool(procs) as pool:
        res = pool.map(_worker, [(i, a.n, a.m, a.window, a.combo, a.reps, cores) for i in range(procs)])
    slowest = max(r[0] for r in res)
    print(json.dumps({"value": round(procs * a.n / slowest / 1e6, 4), "unit": "Msamples/s", "cores": procs,
                      "kind": res[0][1], "sample": f"{procs} independent channels, one per core, n={a.n} each, m={a.m}, "
                      f"{a.window}, {a.combo}, best of {a.reps} per worker, slowest worker counts"}))


if __name__ == "__main__":
    main()
