"""Synthesis right after the analysis wrote the matrix (the pattern of bench.py's reference_bench_shape and of every host
of the two reference calls) against synthesis of a matrix that is only read: per call, synchronous wall clock."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT


def run(combo, m, n, **opts):
    td = torch.float64 if combo[:3] == "f64" else torch.float32
    cdt = torch.complex128 if combo[3:] == "f64" else torch.complex64
    x = torch.zeros(n, dtype=td, device="cuda")
    d = torch.empty((n, m), dtype=cdt, device="cuda")
    y = torch.empty(n, dtype=td, device="cuda")
    p = SDFT(m, "hann", 1.0, combo)
    for k, v in opts.items():
        p.set_option(k, v)
    fw, iv, iv2 = [], [], []
    for r in range(14):
        t0 = time.perf_counter(); p.sdft(x, d); t1 = time.perf_counter(); p.isdft(d, y); t2 = time.perf_counter(); p.isdft(d, y); t3 = time.perf_counter()
        if r >= 2:
            fw.append(t1 - t0); iv.append(t2 - t1); iv2.append(t3 - t2)
    esz = 16 if combo[3:] == "f64" else 8
    b = n * m * esz
    print(f"{combo} m={m} n={n} {opts}: sdft {np.median(fw) * 1e6:7.1f} us  isdft after the write {np.median(iv) * 1e6:7.1f} us ({b / np.median(iv) / 1e9:5.0f} GB/s)"
          f"  isdft again {np.median(iv2) * 1e6:7.1f} us ({b / np.median(iv2) / 1e9:5.0f} GB/s)  form {p.get_option('last_inverse_form')}")
    p.close()


if __name__ == "__main__":
    for n in (12000, 44100, 131072, 500000):
        for nt in (0, 1):
            run("f64f64", 1000, n, inverse_nt=nt)
    for nt in (0, 1):
        run("f32f64", 1000, 44100, inverse_nt=nt)
        run("f32f64", 1024, 48000, inverse_nt=nt)
        run("f32f64", 1024, 1000000, inverse_nt=nt)
        run("f32f32", 4096, 262144, inverse_nt=nt)
