import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
for n in (1024, 4096, 12000):
    x = torch.from_numpy(sine_sweep(n)).cuda(); y = torch.empty_like(x)
    out = torch.empty((n, 1024), dtype=torch.complex128, device="cuda")
    for opts in ({}, {"fft_carry": 0}, {"fuse_delta": 0}):
        p = SDFT(1024, "hann", 1.0, "f32f64"); p.set_option("async", 1)
        for k, v in opts.items(): p.set_option(k, v)
        for what in ("sdft", "process"):
            f = (lambda: p.sdft(x, out)) if what == "sdft" else (lambda: p.process(x, out=y))
            for _ in range(5): f()
            p.synchronize()
            t0 = time.perf_counter()
            for _ in range(200): f()
            p.synchronize(); dt = (time.perf_counter() - t0) / 200
            print(f"n={n} {opts} {what} async: {dt*1e6:.1f} us per call (chunks {p.get_option('last_chunks')})", flush=True)
        p.close()
