"""Synthesis in the reference's order by form: the plan's own choice on the host's calls (logic::FormTuner, default) against the
static choice (inverse_tune = 0) and forced forms (rows per wave, bytes per row segment); interleaved, asynchronous calls.
(Round 5 also measured the waves that run together reading from eight regions of the matrix in turn -- what xcd_map is to the
analysis: +6 % / +4 % / -5.5 % / -5 % by matrix, profiles/r05_store_ceiling_study.txt; not kept.)
    python scripts/synthesis_forms_ab.py [rounds]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
print(f"device: {torch.cuda.get_device_name(0)}")
for label, m, n, ch, combo, reps in (("n=1e6 m=1024 f32f64", 1024, 1_000_000, 1, "f32f64", 6), ("64 ch x 48000 m=1024 f32f64", 1024, 48000, 64, "f32f64", 3),
                                     ("n=44100 m=1000 f64f64", 1000, 44100, 1, "f64f64", 40), ("n=262144 m=4096 f32f32", 4096, 262144, 1, "f32f32", 8),
                                     ("n=600000 m=1024 f32f64", 1024, 600000, 1, "f32f64", 8), ("n=262144 m=1024 f64f64", 1024, 262144, 1, "f64f64", 10)):
    esz = 16 if combo[3:] == "f64" else 8
    cdt = torch.complex128 if esz == 16 else torch.complex64
    d = torch.randn((ch, n, m) if ch > 1 else (n, m), dtype=torch.float32 if esz == 8 else torch.float64, device="cuda").to(cdt)
    plans = []
    for vl, v, rpi, rows, tune in (("tuned", 0, 4, 0, 1), ("static", 0, 4, 0, 0), ("16r 512B", 0, 2, 16, 0), ("16r 256B", 0, 4, 16, 0)) + ((("32r 256B", 0, 4, 32, 0),) if esz == 16 else ()):
        p = SDFT(m, "hann", 1.0, combo, channels=ch)
        p.set_option("async", 1); p.set_option("inverse_rpi", rpi); p.set_option("inverse_rows", rows); p.set_option("inverse_tune", tune)
        for _ in range(10):
            p.isdft(d)
        y = p.isdft(d)
        p.synchronize()
        plans.append((vl, p, y))
    same = all(bool(torch.equal(plans[0][2], q[2])) for q in plans[1:])
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
        print(f"{label:30s} {vl:8s} {w * 1e3:8.3f} ms = {b / w / 1e9:6.0f} GB/s = {b / w / 8e12:5.1%} of peak  form {p.get_option('last_inverse_form')} tuned {p.get_option('last_inverse_tuned')}  same bits: {same}")
        p.close()
    del d, plans
    torch.cuda.empty_cache()
