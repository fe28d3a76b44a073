"""Medium calls on host memory (m = 1024; 4 ... 64 MB of matrix) by the number of DMA streams of the pinned ring
(copy_streams: 1 = the plan's stream, 2 = two streams for the 8 MiB ring only, 3 = two streams for every copy of several pieces)."""
import sys, time
import ctypes as C
import numpy as np
sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
m = 1024
for n in (300, 600, 1200, 2000, 4000):
    x = sine_sweep(n)
    out = np.zeros((n, m), dtype=np.complex128)
    y = np.zeros(n, dtype=np.float32)
    row = []
    for streams in (1, 2, 3, 1, 3):
        with SDFT(m, "hann", 1.0, "f32f64") as p:
            p.set_option("copy_streams", streams)
            p.set_option("carry", 0)
            ws, wi = [], []
            for rep in range(12):
                t0 = time.perf_counter(); p.api.sdft_n(p._p, n, C.c_void_p(x.ctypes.data), C.c_void_p(out.ctypes.data)); ws.append(time.perf_counter() - t0)
                t0 = time.perf_counter(); p.api.isdft_n(p._p, n, C.c_void_p(out.ctypes.data), C.c_void_p(y.ctypes.data)); wi.append(time.perf_counter() - t0)
            row.append(f"streams={streams}: {np.median(ws[2:])*1e6:7.0f} / {np.median(wi[2:])*1e6:7.0f} us")
    print(f"n={n:5d} ({n*m*16/2**20:5.1f} MiB) sdft_n / isdft_n  " + "   ".join(row), flush=True)
