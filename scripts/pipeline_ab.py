"""Development probe (round 6): asynchronous analysis calls into two matrices in turn, option pipeline = 0 / 1 / 2, interleaved in one process, several
rounds, at the north star's length and longer; both matrices placed by the library (equal footing)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdft_amd import capi
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
m = 1024
for n in (24000, 48000, 90000, 131072, 1000000):
    x = torch.from_numpy(sine_sweep(n)).cuda()
    pms = [capi.PlacedMatrix((n, m), torch.complex128) for _ in range(2)]
    print(f"n={n}: matrices placed at {pms[0].info['window_gbs']:.0f} / {pms[1].info['window_gbs']:.0f} GB/s store-only", flush=True)
    ptr = [C.c_void_p(p.ptr) for p in pms]
    xp = C.c_void_p(x.data_ptr())
    b = n * (m * 16 + 4)
    reps = 200 if n < 200000 else 20
    for rnd in range(4):
        out = []
        for pipe in (0, 1, 2):
            p = SDFT(m, "hann", 1.0, "f32f64")
            p.set_option("async", 1); p.set_option("pipeline", pipe)
            for i in range(6): p.api.sdft_n(p._p, n, xp, ptr[i & 1])
            p.synchronize()
            t0 = time.perf_counter()
            for i in range(reps): p.api.sdft_n(p._p, n, xp, ptr[i & 1])
            p.synchronize()
            w = (time.perf_counter() - t0) / reps
            out.append(f"pipeline={pipe}: {w * 1e6:8.1f} us = {b / w / 8e12:.4f} (pipelined {p.get_option('pipelined_calls')}, chunks {p.get_option('last_chunks')}x{p.get_option('last_chunk_len')})")
            p.close()
        print(f"  round {rnd}  " + "   ".join(out), flush=True)
    for p in pms: p.free()
