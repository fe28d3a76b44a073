import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
for n in (24000, 48000, 131072):
    x = torch.from_numpy(sine_sweep(n)).cuda()
    out = torch.empty((n, 1024), dtype=torch.complex128, device="cuda")
    for rep in range(2):
        for opts in ({}, {"fft_carry": 0}):
            p = SDFT(1024, "hann", 1.0, "f32f64"); p.set_option("async", 1)
            for k, v in opts.items(): p.set_option(k, v)
            for _ in range(5): p.sdft(x, out)
            p.synchronize()
            t0 = time.perf_counter()
            for _ in range(100): p.sdft(x, out)
            p.synchronize(); dt = (time.perf_counter() - t0) / 100
            print(f"n={n} {opts} sdft async: {dt*1e6:.1f} us per call (chunks {p.get_option('last_chunks')} x {p.get_option('last_chunk_len')})", flush=True)
            p.close()
