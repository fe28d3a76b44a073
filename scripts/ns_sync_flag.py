"""North-star shape (n = 48000, m = 1024, f32f64), synchronous calls: completion by the kernel's word in pinned host memory
(option flag_max raised) against the stream spin."""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

m = 1024
for n in (2048, 4096, 8192, 12000, 24000, 48000, 100000):
    x = torch.from_numpy(sine_sweep(n)).cuda()
    o = torch.empty((n, m), dtype=torch.complex128, device="cuda")
    for label, opts in (("default", {}), ("flag_max 2^28", {"flag_max": 1 << 28}), ("spin 0 (sleeping synchronize)", {"spin": 0}), ("async", {"async": 1})):
        p = SDFT(m, "hann", 1.0, "f32f64")
        for k, v in opts.items():
            p.set_option(k, v)
        xs, os_ = C.c_void_p(x.data_ptr()), C.c_void_p(o.data_ptr())
        for _ in range(5):
            p.api.sdft_n(p._p, n, xs, os_)
        p.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            p.api.sdft_n(p._p, n, xs, os_)
        p.synchronize()
        w = (time.perf_counter() - t0) / 100
        print(f"n={n:6d} {label:32s} {w * 1e6:7.1f} us per call = {n * (m * 16 + 4) / w / 8e12:5.1%} of peak")
        p.close()
