"""Asynchronous synthesis calls back to back on two matrices in turn (m = 1024, f32f64, raw C-ABI): one stream (pipeline = 0)
against the two row streams in turn (default)."""
import ctypes as C, sys, time
import torch
sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
m = 1024
for n in (12000, 48000, 262144, 1000000):
    x = torch.from_numpy(sine_sweep(n)).cuda()
    M = [torch.empty((n, m), dtype=torch.complex128, device="cuda") for _ in range(2)]
    ys = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(2)]
    for rep in range(2):
        for pipe in (0, 1):
            pa = SDFT(m, "hann", 1.0, "f32f64")
            pa.set_option("async", 1); pa.set_option("pipeline", pipe)
            pa.sdft(x, M[0]); pa.sdft(x, M[1]); pa.synchronize()
            mp = [C.c_void_p(M[0].data_ptr()), C.c_void_p(M[1].data_ptr())]; yp = [C.c_void_p(ys[0].data_ptr()), C.c_void_p(ys[1].data_ptr())]
            for i in range(4): pa.api.isdft_n(pa._p, n, mp[i & 1], yp[i & 1])
            pa.synchronize(); torch.cuda.synchronize()
            reps = 20
            t0 = time.perf_counter()
            for i in range(reps): pa.api.isdft_n(pa._p, n, mp[i & 1], yp[i & 1])
            pa.synchronize(); torch.cuda.synchronize()
            w = (time.perf_counter() - t0) / reps
            print(f"n={n:7d} synthesis of two matrices in turn, pipeline={pipe}: {w * 1e6:8.1f} us per call = {n * (m * 16 + 4) / w / 8e12:5.1%}  (pipelined {pa.get_option('pipelined_inverse_calls')})", flush=True)
            pa.close()
