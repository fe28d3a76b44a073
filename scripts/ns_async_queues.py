"""Pipelined calls and hardware queues: n = 48 000 / 131 072, two matrices in turn, row streams with and without distinct
priorities, in a clean process and in one that has a dozen other streams alive (what bench.py looks like)."""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

m = 1024
big = [torch.empty((131072, m), dtype=torch.complex128, device="cuda") for _ in range(2)]
crowd = []
for crowded in (0, 1):
    if crowded:
        crowd = [torch.cuda.Stream() for _ in range(12)]
        for s_ in crowd:
            with torch.cuda.stream(s_):
                torch.zeros(16, device="cuda").add_(1)
        torch.cuda.synchronize()
    for n in (48000, 131072):
        x = torch.from_numpy(sine_sweep(n)).cuda()
        for pipe, prio in ((0, 0), (1, 0), (1, 0)):
            p = SDFT(m, "hann", 1.0, "f32f64")
            p.set_option("async", 1)
            p.set_option("pipeline", pipe)
            xp = C.c_void_p(x.data_ptr())
            op = [C.c_void_p(big[0].data_ptr()), C.c_void_p(big[1].data_ptr())]
            for i in range(6):
                p.api.sdft_n(p._p, n, xp, op[i & 1])
            p.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(50):
                p.api.sdft_n(p._p, n, xp, op[i & 1])
            p.synchronize(); torch.cuda.synchronize()
            w = (time.perf_counter() - t0) / 50
            print(f"other streams alive: {12 * crowded:2d}  n={n:6d} pipeline={pipe}: {w * 1e6:7.1f} us per call = {n * (m * 16 + 4) / w / 8e12:5.1%}   (streams: kind*10 + pairs tried = {p.get_option('pipeline_streams')})", flush=True)
            p.close()
