"""Do the windows of an allocation that take the analysis' stores fast read fast too?  n = 1e6, m = 1024, f32f64: synthesis (only read, tuners settled) of a matrix placed
in the window with the best store-only rate, in the one with the worst, and in a separate allocation; and the batch share (64 x 48000) the same way."""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd import capi
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep, sweep_batch

lib = capi.load()
print(f"device: {torch.cuda.get_device_name(0)}")
for label, m, n, ch, chunk_len in (("n=1e6", 1024, 1000000, 1, 1960), ("64 x 48000", 1024, 48000, 64, 6000)):
    nbytes = ch * n * m * 16
    x = torch.from_numpy(sweep_batch(ch, n) if ch > 1 else sine_sweep(n)).cuda()
    free, _ = torch.cuda.mem_get_info()
    step = 4 << 30
    abytes = ((free - nbytes - (30 << 30)) // step) * step
    arena = torch.empty(abytes, dtype=torch.uint8, device="cuda")
    offs = list(range(0, abytes - nbytes + 1, step))
    rates = []
    for o in offs:
        ms = lib.sdft_hip_store_ceiling(arena.data_ptr() + o, (nbytes // 16384) * 16384, 4, 1024, 8, chunk_len, 2)
        rates.append(nbytes / (ms * 1e-3) / 1e9)
    best, worst = int(np.argmax(rates)), int(np.argmin(rates))
    shape = (ch, n, m) if ch > 1 else (n, m)
    mats = {f"best store window ({rates[best]:.0f} GB/s)": arena[offs[best]:offs[best] + nbytes].view(torch.complex128).view(shape),
            f"worst store window ({rates[worst]:.0f} GB/s)": arena[offs[worst]:offs[worst] + nbytes].view(torch.complex128).view(shape),
            "separate allocation": torch.empty(shape, dtype=torch.complex128, device="cuda")}
    for name, d in mats.items():
        p = SDFT(m, "hann", 1.0, "f32f64", channels=ch)
        p.set_option("async", 1)
        p.sdft(x, d)
        y = None
        for _ in range(16):
            y = p.isdft(d, y)
        p.synchronize()
        ts = []
        for r in range(3):
            t0 = time.perf_counter()
            for _ in range(4):
                p.isdft(d, y)
            p.synchronize()
            ts.append((time.perf_counter() - t0) / 4)
        ta = []
        for r in range(3):
            t0 = time.perf_counter()
            for _ in range(4):
                p.sdft(x, d)
            p.synchronize()
            ta.append((time.perf_counter() - t0) / 4)
        w, wa = float(np.median(ts)), float(np.median(ta))
        print(f"{label:10s} {name:34s}: synthesis {w * 1e3:7.3f} ms = {nbytes / w / 1e9:5.0f} GB/s (form {p.get_option('last_inverse_tuned')})   analysis {wa * 1e3:7.3f} ms = {nbytes / wa / 1e9:5.0f} GB/s", flush=True)
        p.close()
    del mats, arena, d
    torch.cuda.empty_cache()
