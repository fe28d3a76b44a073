"""The streaming synthesis forms on matrices beyond 4 GiB with ordinary loads (inverse_nt = -1: streaming loads only between 256 MiB and 4 GiB) against
non-temporal loads (inverse_nt = 1); the plan's tuner on in both; interleaved.    python scripts/synthesis_nt_ab.py [rounds]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
print(f"device: {torch.cuda.get_device_name(0)}")
for label, m, n, ch, combo, reps in (("n=1e6 m=1024 f32f64 (streaming forms)", 1024, 1_000_000, 1, "f32f64", 6), ("n=1e6 m=1024 f64f64", 1024, 1_000_000, 1, "f64f64", 6),
                                     ("n=300000 m=1024 f64f64", 1024, 300000, 1, "f64f64", 10), ("n=262144 m=4096 f32f32", 4096, 262144, 1, "f32f32", 8),
                                     ("64 ch x 48000 m=1024 f64f64", 1024, 48000, 64, "f64f64", 3), ("n=1e6 m=1024 f32f32", 1024, 1_000_000, 1, "f32f32", 8)):
    esz = 16 if combo[3:] == "f64" else 8
    cdt = torch.complex128 if esz == 16 else torch.complex64
    d = torch.randn((ch, n, m) if ch > 1 else (n, m), dtype=torch.float32 if esz == 8 else torch.float64, device="cuda").to(cdt)
    plans = []
    for vl, nt in (("ordinary loads", -1), ("non-temporal loads", 1)):
        p = SDFT(m, "hann", 1.0, combo, channels=ch)
        p.set_option("async", 1); p.set_option("inverse_nt", nt); p.set_option("inverse_step", -1)
        for _ in range(10):
            p.isdft(d)
        y = p.isdft(d)
        p.synchronize()
        plans.append((vl, p, y))
    same = bool(torch.equal(plans[0][2], plans[1][2]))
    res = {vl: [] for vl, _, _ in plans}
    for r in range(rounds):
        for vl, p, y in plans:
            p.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                p.isdft(d, y)
            p.synchronize()
            res[vl].append((time.perf_counter() - t0) / reps)
    b = ch * n * (m * esz + 4)
    for vl, p, y in plans:
        w = float(np.median(res[vl]))
        print(f"{label:40s} {vl:20s} {w * 1e3:8.3f} ms = {b / w / 1e9:6.0f} GB/s = {b / w / 8e12:5.1%} of peak  tuned {p.get_option('last_inverse_tuned')}  same bits: {same}", flush=True)
        p.close()
    del d, plans
    torch.cuda.empty_cache()
