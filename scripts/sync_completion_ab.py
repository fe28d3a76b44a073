"""Synchronous completion of sdft_sdft_n (m = 1024, f32f64) by strategy, interleaved in one process so that a box's drift
hits every side alike: option "spin" = 0 (sleeping hipStreamSynchronize: round 4's choice beyond 60 us), 1 (quiet spin on the
host clock until bytes / 8 TB/s, then hipStreamQuery polls: default), 2 (polls from the start: round 3), the completion word
for every size (flag_max 2^40) and asynchronous calls as the floor.  Prints microseconds per call and the gap to async.
    python scripts/sync_completion_ab.py [rounds]"""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
m = 1024
modes = (("sleep (spin=0)", {"spin": 0}), ("quiet+poll (spin=1, default)", {"spin": 1}), ("poll from start (spin=2)", {"spin": 2}),
         ("completion word, any size", {"spin": 1, "flag_max": 1 << 40}), ("async", {"async": 1}))
print(f"device: {torch.cuda.get_device_name(0)}")
for n, reps in ((12000, 200), (48000, 100), (100000, 60), (1000000, 12)):
    x = torch.from_numpy(sine_sweep(n)).cuda()
    o = torch.empty((n, m), dtype=torch.complex128, device="cuda")
    xs, os_ = C.c_void_p(x.data_ptr()), C.c_void_p(o.data_ptr())
    plans = []
    for label, opts in modes:
        p = SDFT(m, "hann", 1.0, "f32f64")
        for k, v in opts.items():
            p.set_option(k, v)
        for _ in range(5):
            p.api.sdft_n(p._p, n, xs, os_)
        p.synchronize()
        plans.append((label, p))
    res = {label: [] for label, _ in plans}
    for r in range(rounds):
        for label, p in plans:
            p.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                p.api.sdft_n(p._p, n, xs, os_)
            p.synchronize()
            res[label].append((time.perf_counter() - t0) / reps * 1e6)
    base = float(np.median(res["async"]))
    for label, p in plans:
        v = res[label]
        med = float(np.median(v))
        print(f"n={n:8d} {label:30s} median {med:8.1f} us  min {min(v):8.1f}  max {max(v):8.1f}  gap to async {med - base:6.1f} us  "
              f"= {n * (m * 16 + 4) / (med * 1e-6) / 8e12:5.1%} of peak")
        p.close()
    del x, o
    torch.cuda.empty_cache()
