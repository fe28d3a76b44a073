"""North-star shape, asynchronous calls through the raw C-ABI: one matrix / two matrices in turn, pipeline 0 / 1."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sizes = [int(a) for a in sys.argv[2:]] or [12000, 24000, 36000, 48000, 48000, 66000, 131072, 262144]
big = [torch.empty((max(sizes), m), dtype=torch.complex128, device="cuda") for _ in range(2)]
for n in sizes:
    x = torch.from_numpy(sine_sweep(n)).cuda()
    o = [b[:n] for b in big]
    for pipe in (0, 1):
        for bufs in (1, 2):
            p = SDFT(m, "hann", 1.0, "f32f64")
            p.set_option("async", 1)
            p.set_option("pipeline", pipe)
            xp = C.c_void_p(x.data_ptr())
            op = [C.c_void_p(o[0].data_ptr()), C.c_void_p(o[(bufs - 1)].data_ptr())]
            for i in range(6):
                p.api.sdft_n(p._p, n, xp, op[i & 1])
            p.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(50):
                p.api.sdft_n(p._p, n, xp, op[i & 1])
            t1 = time.perf_counter()
            p.synchronize(); torch.cuda.synchronize()
            w = (time.perf_counter() - t0) / 50
            print(f"n={n:6d} pipeline={pipe} matrices={bufs}: {w * 1e6:7.1f} us per call = {n * (m * 16 + 4) / w / 8e12:5.1%}   (host enqueue {1e6 * (t1 - t0) / 50:5.1f} us per call, ordered {p.get_option('pipelined_ordered')})", flush=True)
            p.close()
