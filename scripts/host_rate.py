"""One long call on host memory (m = 1024, n = 100 000: 1.6 GB of matrix into a pre-touched numpy array) by option host_copy.
(Registering the caller's buffer for the duration of one call was measured too: 33 MB 0.84 ms, but 328 MB 18 ms and 1.6 GB 115 ms --
pinning costs as much per byte as the memcpy it would save.)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
m, n = 1024, 100000
x = sine_sweep(n)
for mode, threads, streams in ((0, 2, 2), (0, 2, 1), (0, 0, 2), (1, 0, 1), (0, 3, 2), (0, 3, 1), (0, 1, 2), (0, 2, 2), (0, 2, 1), (1, 0, 1)):
    with SDFT(m, "hann", 1.0, "f32f64") as p:
        p.set_option("host_copy", mode)
        p.set_option("copy_threads", threads)
        p.set_option("copy_streams", streams)
        out = np.empty((n, m), dtype=np.complex128); out[:] = 0
        import ctypes as C
        for rep in range(3):
            t0 = time.perf_counter()
            p.api.sdft_n(p._p, n, C.c_void_p(x.ctypes.data), C.c_void_p(out.ctypes.data))
            w = time.perf_counter() - t0
        y = np.empty(n, dtype=np.float32)
        for rep in range(3):
            t0 = time.perf_counter()
            p.api.isdft_n(p._p, n, C.c_void_p(out.ctypes.data), C.c_void_p(y.ctypes.data))
            wi = time.perf_counter() - t0
        print(f"host_copy={mode} copy_threads={threads} copy_streams={streams}: isdft_n host->host: {wi*1e3:.1f} ms = {n*m*16/wi/1e9:.1f} GB/s;  sdft_n host->host n={n}: {w*1e3:.1f} ms = {n/w/1e6:.2f} Msamples/s = {n*m*16/w/1e9:.1f} GB/s")
