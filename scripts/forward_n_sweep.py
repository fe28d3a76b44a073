"""Analysis (m = 1024, f32f64, default path) by call length: is there a staircase in the number of time chunks?"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
o = torch.empty((262144, m), dtype=torch.complex128, device="cuda")
for n in (36000, 48000, 50000, 52000, 56000, 60000, 66000, 72000, 80000, 90000, 98000, 100000, 110000, 120000, 131072, 160000, 200000, 262144):
    x = torch.from_numpy(sine_sweep(n)).cuda()
    line = []
    for chunk in (0, -1):
        p = SDFT(m, "hann", 1.0, "f32f64")
        p.set_option("async", 1)
        if chunk == -1:
            # one round: 256 chunks (or 512: two rounds) of equal length
            k = (256 if n / 256 >= 96 else 128) * max(1, 1024 // m)
            ln = ((n + k - 1) // k + 7) // 8 * 8
            p.set_option("chunk", ln)
        out = o[:n]
        for _ in range(3):
            p.sdft(x, out)
        p.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            p.sdft(x, out)
        p.synchronize()
        w = (time.perf_counter() - t0) / 20
        line.append(f"chunks {p.get_option('last_chunks'):4d} x {p.get_option('last_chunk_len'):4d}: {w * 1e6:7.1f} us = {n * (m * 16 + 4) / w / 8e12:5.1%}")
        p.close()
    print(f"n={n:6d}  default {line[0]}  |  256 equal chunks {line[1]}")
