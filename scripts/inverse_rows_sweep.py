"""Exact-order synthesis, 4 rows x 8 tiles against 8 rows x 4 tiles per wave, by row count (asynchronous calls, read-only matrix)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT


def t(combo, m, n, rows, d, reps=30):
    p = SDFT(m, "hann", 1.0, combo)
    p.set_option("inverse_rows", rows); p.set_option("inverse_verify", 0); p.set_option("inverse_nt", 0); p.set_option("async", 1)
    y = p.isdft(d)
    for _ in range(3):
        p.isdft(d, y)
    p.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        p.isdft(d, y)
    p.synchronize()
    w = (time.perf_counter() - t0) / reps
    p.close()
    return w


for combo, m in (("f64f64", 1000), ("f32f32", 1024), ("f32f32", 4096)):
    cdt = torch.complex128 if combo[3:] == "f64" else torch.complex64
    esz = 16 if combo[3:] == "f64" else 8
    for n in (16384, 24000, 28000, 32768, 36000, 40000, 44100, 48000, 56000, 60000, 65536):
        if n * m * esz > 3e9: continue
        d = torch.randn((n, m), dtype=cdt, device="cuda")
        a, b, c = t(combo, m, n, 4, d), t(combo, m, n, 8, d), t(combo, m, n, 0, d)
        pick = "8x4" if abs(c - b) < abs(c - a) else "4x8"
        print(f"{combo} m={m} n={n:6d}: 4x8 {a * 1e6:7.1f} us ({n * m * esz / a / 1e9:5.0f} GB/s)  8x4 {b * 1e6:7.1f} us ({n * m * esz / b / 1e9:5.0f} GB/s)  {'8x4' if b < a * 0.98 else ('4x8' if a < b * 0.98 else 'same')} wins | default {c * 1e6:7.1f} us (looks like {pick})")
        del d
