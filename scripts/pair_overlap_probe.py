"""Analysis (writes a matrix) beside synthesis (reads another one): two plans on their own streams against one after the other.
n = 1e6 / 262144, m = 1024, f32f64; per pair: 2 x n x 16 KiB of HBM traffic."""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

m = 1024
for n in (262144, 1000000):
    x = torch.from_numpy(sine_sweep(n)).cuda()
    M = [torch.empty((n, m), dtype=torch.complex128, device="cuda") for _ in range(2)]
    y = torch.empty(n, dtype=torch.float32, device="cuda")
    pa = SDFT(m, "hann", 1.0, "f32f64"); pb = SDFT(m, "hann", 1.0, "f32f64")
    for p in (pa, pb):
        p.set_option("async", 1); p.set_option("pipeline", 0)
    pa.sdft(x, M[0]); pa.sdft(x, M[1]); pa.synchronize()
    xp, yp = C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr())
    mp = [C.c_void_p(M[0].data_ptr()), C.c_void_p(M[1].data_ptr())]
    for mode in ("one after the other (one plan)", "side by side (two plans, two matrices)", "one after the other (one plan)", "side by side (two plans, two matrices)"):
        reps = 10
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            if mode.startswith("one"):
                pa.api.sdft_n(pa._p, n, xp, mp[0]); pa.api.isdft_n(pa._p, n, mp[0], yp)
            else:
                pa.api.sdft_n(pa._p, n, xp, mp[i & 1]); pb.api.isdft_n(pb._p, n, mp[(i + 1) & 1], yp)
        pa.synchronize(); pb.synchronize(); torch.cuda.synchronize()
        w = (time.perf_counter() - t0) / reps
        print(f"n={n:7d} {mode:42s}: {w * 1e3:7.3f} ms per pair = {2 * n * (m * 16 + 4) / w / 8e12:5.1%} of peak", flush=True)
    pa.close(); pb.close()
