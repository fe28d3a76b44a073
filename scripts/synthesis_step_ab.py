"""Synthesis, float samples from double bins: whole rows read in step by one workgroup per chunk of rows (inverse_rows_kernel,
option inverse_step = 1) against the plan's tuned choice among the streaming forms (inverse_step = -1: not a candidate) and the
tree-sum form -- same bits?  how fast?  Matrices of real analyses (so that the rounding-interval proof meets real sums).
    python scripts/synthesis_step_ab.py [rounds]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep, sweep_batch

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
print(f"device: {torch.cuda.get_device_name(0)}")
for label, m, n, ch, window, lat, reps in (("n=1e6 m=1024 hann", 1024, 1_000_000, 1, "hann", 1.0, 6), ("64 ch x 48000 m=1024 hann", 1024, 48000, 64, "hann", 1.0, 3),
                                           ("64 ch x 48000 m=2048 hann", 2048, 48000, 64, "hann", 1.0, 2), ("n=48000 m=1024 hann", 1024, 48000, 1, "hann", 1.0, 40),
                                           ("n=100000 m=1024 hann", 1024, 100000, 1, "hann", 1.0, 30), ("n=262144 m=1024 hann", 1024, 262144, 1, "hann", 1.0, 12),
                                           ("n=500000 m=1024 blackman", 1024, 500000, 1, "blackman", 1.0, 8), ("n=262144 m=1000 hann", 1000, 262144, 1, "hann", 1.0, 10),
                                           ("n=500000 m=512 hann", 512, 500000, 1, "hann", 1.0, 10), ("n=250000 m=2048 hann", 2048, 250000, 1, "hann", 1.0, 8),
                                           ("n=100000 m=320 hamming", 320, 100000, 1, "hamming", 1.0, 20), ("n=262144 m=1000 blackman latency 0.5", 1000, 262144, 1, "blackman", 0.5, 10)):
    x = torch.from_numpy(sweep_batch(ch, n) if ch > 1 else sine_sweep(n)).cuda()
    plans = []
    d = None
    for vl, opts in (("rows in step", {"inverse_step": 1}), ("tuned streaming forms", {"inverse_step": -1}), ("tree sum, wave per row", {"inverse_step": -1, "inverse_tune": 0, "inverse_verify": 1}),
                     ("tuner with rows in step", {})):
        p = SDFT(m, window, lat, "f32f64", channels=ch)
        p.set_option("async", 1)
        for k, v in opts.items():
            p.set_option(k, v)
        if d is None:
            d = p.sdft(x)
        for _ in range(10):
            p.isdft(d)
        y = p.isdft(d)
        p.synchronize()
        plans.append((vl, p, y))
    same = [bool(torch.equal(plans[1][2].view(torch.int32), q[2].view(torch.int32))) for q in plans]
    res = {vl: [] for vl, _, _ in plans}
    for r in range(rounds):
        for vl, p, y in plans:
            p.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                p.isdft(d, y)
            p.synchronize()
            res[vl].append((time.perf_counter() - t0) / reps)
    b = ch * n * (m * 16 + 4)
    for (vl, p, y), sm in zip(plans, same):
        w = float(np.median(res[vl]))
        print(f"{label:38s} {vl:26s} {w * 1e3:8.3f} ms = {b / w / 1e9:6.0f} GB/s = {b / w / 8e12:5.1%} of peak  form {p.get_option('last_inverse_form')} tuned {p.get_option('last_inverse_tuned')}  same bits: {sm}", flush=True)
        p.close()
    del d, plans, x
    torch.cuda.empty_cache()
