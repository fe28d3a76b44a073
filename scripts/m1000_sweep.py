"""m = 1000 (the reference's own test size: rows of 16000 bytes, a partial 16th wave), analysis by chunk length and carry form."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep


def run(m, n, **opts):
    x = torch.from_numpy(sine_sweep(n)).cuda()
    p = SDFT(m, "hann", 1.0, "f32f64")
    for k, v in opts.items():
        p.set_option(k, v)
    d = p.sdft(x)
    p.set_option("async", 1)
    for _ in range(3):
        p.sdft(x, d)
    p.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        p.sdft(x, d)
    p.synchronize()
    w = (time.perf_counter() - t0) / 10
    print(f"m={m} n={n} {str(opts):44s} chunks {p.get_option('last_chunks'):4d} x {p.get_option('last_chunk_len'):5d} self {p.get_option('last_self')}: {w * 1e6:8.1f} us = {n * (m * 16 + 4) / w / 1e9:6.0f} GB/s = {n * (m * 16 + 4) / w / 8e12:5.1%}")
    p.close()
    del d


if __name__ == "__main__":
    for m in (1000, 1024):
        for n in (352800, 48000):
            run(m, n)
            run(m, n, self_carry=0)
            for chunk in (96, 128, 192, 256, 344, 384, 512, 696, 1024, 1384):
                if chunk * 8 <= n:
                    run(m, n, chunk=chunk)
