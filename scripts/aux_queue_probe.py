"""Does the exact-carry route's overlap (relay on the plan's auxiliary stream beside the forward launch on its main stream)
depend on which hardware queues the runtime deals the two streams?  configs[2] analysis with K other streams created first.
    python scripts/aux_queue_probe.py"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

m, window, n = 4096, "blackman", 262144
x = torch.from_numpy(sine_sweep(n)).cuda()
d = torch.empty((n, m), dtype=torch.complex64, device="cuda")
keep = []
for k in range(0, 9):
    p = SDFT(m, window, 1.0, "f32f32")
    for _ in range(2):
        p.sdft(x, d)
    p.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        p.sdft(x, d)
    p.synchronize()
    w = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{k} other streams alive: {w:7.3f} ms per analysis call  ({n * (m * 8 + 4) / (w * 1e-3) / 8e12:5.1%} of peak)")
    p.close()
    keep.append(torch.cuda.Stream())          # one more stream alive for the next plan
    with torch.cuda.stream(keep[-1]):
        torch.zeros(1, device="cuda")
