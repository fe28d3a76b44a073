"""Short-call probe (development aid): wall time per call of the default drop-in path for the
north-star shape (n = 48000) and the reference's streaming test shape (hop = 100, m = 1000,
/root/reference/test/main.sh:3-6), synchronous and asynchronous, device pointers."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep


def north_star(n=48000, m=1024, reps=50, **opts):
    x = torch.from_numpy(sine_sweep(n)).cuda()
    out = torch.empty((n, m), dtype=torch.complex128, device="cuda")
    b = n * (m * 16 + 4)
    for mode in ("sync", "async"):
        p = SDFT(m)
        for k, v in opts.items():
            p.set_option(k, v)
        if mode == "async":
            p.set_option("async", 1)
        for _ in range(5):
            p.sdft(x, out)
        p.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            p.sdft(x, out)
        p.synchronize(); torch.cuda.synchronize()
        w = (time.perf_counter() - t0) / reps
        print(f"n={n} m={m} {mode:5s} opts={opts}: {w*1e6:8.1f} us/call  {n/w/1e6:7.1f} Msamples/s  {b/w/1e9:7.1f} GB/s "
              f"= {b/w/8e12*100:5.1f} % of peak  chunks={p.get_option('last_chunks')}", flush=True)
        p.close()


def hops(m=1000, hop=100, total=20000, combo="f32f64", **opts):
    td = np.float32 if combo[:3] == "f32" else np.float64
    cdt = torch.complex128 if combo[3:] == "f64" else torch.complex64
    x = torch.from_numpy(sine_sweep(total, dtype=td)).cuda()
    y = torch.empty(total, dtype=x.dtype, device="cuda")
    d = torch.empty((hop, m), dtype=cdt, device="cuda")
    for mode in ("sync", "async"):
        p = SDFT(m, "hann", 1.0, combo)
        for k, v in opts.items():
            p.set_option(k, v)
        if mode == "async":
            p.set_option("async", 1)
        for rep in range(2):
            p.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(0, total, hop):
                p.sdft(x[i:i + hop], d)
                p.isdft(d, y[i:i + hop])
            p.synchronize(); torch.cuda.synchronize()
            w = (time.perf_counter() - t0) / (total // hop)
        print(f"hop={hop} m={m} {combo} {mode:5s} opts={opts}: {w*1e6:7.1f} us/hop (sdft_n + isdft_n)  {hop/w/1e6:6.2f} Msamples/s", flush=True)
        p.close()
    # raw ctypes loop without the tensor-slicing overhead of the Python wrapper
    import ctypes as C
    p = SDFT(m, "hann", 1.0, combo)
    for k, v in opts.items():
        p.set_option(k, v)
    xs, ys, ds, isz = x.data_ptr(), y.data_ptr(), d.data_ptr(), x.element_size()
    for mode in ("sync", "async"):
        p.set_option("async", 1 if mode == "async" else 0)
        for rep in range(2):
            p.synchronize()
            t0 = time.perf_counter()
            for i in range(0, total, hop):
                p.api.sdft_n(p._p, hop, C.c_void_p(xs + i * isz), C.c_void_p(ds))
                p.api.isdft_n(p._p, hop, C.c_void_p(ds), C.c_void_p(ys + i * isz))
            p.synchronize()
            w = (time.perf_counter() - t0) / (total // hop)
        print(f"hop={hop} m={m} {combo} {mode:5s} raw C-ABI loop: {w*1e6:7.1f} us/hop", flush=True)
    p.close()


def hop_process(m=1000, hop=100, total=20000, combo="f32f64"):
    import ctypes as C
    td = np.float32 if combo[:3] == "f32" else np.float64
    x = torch.from_numpy(sine_sweep(total, dtype=td)).cuda()
    y = torch.empty(total, dtype=x.dtype, device="cuda")
    p = SDFT(m, "hann", 1.0, combo)
    xs, ys, isz = x.data_ptr(), y.data_ptr(), x.element_size()
    for mode in ("sync", "async"):
        p.set_option("async", 1 if mode == "async" else 0)
        for rep in range(2):
            p.synchronize()
            t0 = time.perf_counter()
            for i in range(0, total, hop):
                p.api.process_n(p._p, hop, C.c_void_p(xs + i * isz), C.c_void_p(ys + i * isz), 0, None, None)
            p.synchronize()
            w = (time.perf_counter() - t0) / (total // hop)
        print(f"hop={hop} m={m} {combo} {mode:5s} sdft_hip_process_n (one call per hop): {w*1e6:7.1f} us/hop", flush=True)
    p.close()


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "ns"):
        north_star()
        north_star(pointers=1)
    if which in ("all", "hop"):
        hop_process()
        hop_process(combo="f32f32")
        hops()
        hops(pointers=1)
        hops(combo="f32f32")
    if which == "prof":           # few iterations for rocprofv3 --kernel-trace
        hops(total=4000)
        north_star(reps=10)
