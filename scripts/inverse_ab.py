"""A/B of the exact-order synthesis forms by call length (inside one process): rows per wave x tiles in flight."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT

# (round 4 also measured 16 rows x 4 tiles in flight -- slower than 4 x 8 below 64 Ki rows and than 16 x 1 above; the instantiation is gone)
FORMS = [("default", {}), ("4 rows x 8 tiles", {"inverse_rows": 4}), ("8 x 4", {"inverse_rows": 8}), ("16 x 1", {"inverse_rows": 16}), ("32 x 1", {"inverse_rows": 32})]


def run(combo, m, n, reps=20):
    cdt = torch.complex128 if combo[3:] == "f64" else torch.complex64
    d = torch.randn((n, m), dtype=cdt, device="cuda")
    esz = 16 if combo[3:] == "f64" else 8
    outs = []
    print(f"{combo} m={m} n={n}  ({n * m * esz / 1e6:.0f} MB)")
    for label, opts in FORMS:
        if label.startswith("32") and combo[3:] == "f32":
            continue
        p = SDFT(m, "hann", 1.0, combo)
        for k, v in opts.items():
            p.set_option(k, v)
        y = p.isdft(d)
        for _ in range(3):
            p.isdft(d, y)
        p.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            p.isdft(d, y)
        p.synchronize()
        w = (time.perf_counter() - t0) / reps
        p.set_option("async", 1)
        t0 = time.perf_counter()
        for _ in range(reps):
            p.isdft(d, y)
        p.synchronize()
        wa = (time.perf_counter() - t0) / reps
        outs.append(y.clone())
        print(f"   {label:18s} form {p.get_option('last_inverse_form')}  sync {w * 1e6:8.1f} us  async {wa * 1e6:8.1f} us = {n * m * esz / wa / 1e9:6.0f} GB/s")
        p.close()
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    print("   bits:", "EQUAL" if same else "DIFFER")


if __name__ == "__main__":
    for combo, m in (("f64f64", 1000), ("f32f32", 1024)):
        for n in (4096, 12000, 24000, 44100, 65536, 131072):
            if n * m * (16 if combo[3:] == "f64" else 8) > 12e9:
                continue
            run(combo, m, n)
