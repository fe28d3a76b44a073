"""The reference's streaming test shape (test/main.sh:3-6: dftsize 1000, hop 100, Hann; test/test.c:69-83: sdft_sdft_n +
sdft_isdft_n per hop) by the number of time parts of the hop's analysis launch (option hop_parts; 0 = by the launch's size):
kernel time by HIP events and wall clock per hop of the two reference calls, synchronous and asynchronous; interleaved.
    python scripts/hop_parts_ab.py [rounds]"""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
m, hop, total = 1000, 100, 20000
x = torch.from_numpy(sine_sweep(total)).cuda()
y = torch.empty(total, dtype=x.dtype, device="cuda")
d = torch.empty((hop, m), dtype=torch.complex128, device="cuda")
xs, ys, ds, isz = x.data_ptr(), y.data_ptr(), d.data_ptr(), x.element_size()
print(f"device: {torch.cuda.get_device_name(0)}")
plans = {parts: SDFT(m, "hann", 1.0, "f32f64") for parts in (1, 2, 4, 0, 8, 12)}
for parts, p in plans.items():
    p.set_option("hop_parts", parts)
res = {(parts, mode): [] for parts in plans for mode in ("sync", "async")}


def loop(p):
    for i in range(0, total, hop):
        p.api.sdft_n(p._p, hop, C.c_void_p(xs + i * isz), C.c_void_p(ds))
        p.api.isdft_n(p._p, hop, C.c_void_p(ds), C.c_void_p(ys + i * isz))


for r in range(rounds + 1):
    for parts, p in plans.items():
        for mode in ("sync", "async"):
            p.set_option("async", 1 if mode == "async" else 0)
            p.synchronize()
            t0 = time.perf_counter()
            loop(p)
            p.synchronize()
            if r:
                res[(parts, mode)].append((time.perf_counter() - t0) / (total // hop) * 1e6)
for parts, p in plans.items():
    p.set_option("async", 1); p.set_option("profile", 1)
    loop(p)
    pr = p.profile()
    fk = pr["forward"][0] / max(pr["forward"][1], 1) * 1e3
    ik = pr["inverse"][0] / max(pr["inverse"][1], 1) * 1e3
    cl = {}
    for mode in ("sync", "async"):
        p.set_option("profile", 0); p.set_option("async", 1 if mode == "async" else 0)
        cl[mode] = min(p.api.time_hops(p._p, total // hop, hop, C.c_void_p(xs), C.c_void_p(ds), C.c_void_p(ys)) for _ in range(3)) / (total // hop) * 1e6
    print(f"hop_parts={parts:2d} (launch: {p.get_option('last_hop_parts')} parts): analysis kernel {fk:5.1f} us, synthesis kernel {ik:5.1f} us; "
          f"two calls per hop: sync {np.median(res[(parts, 'sync')]):5.1f} us, async {np.median(res[(parts, 'async')]):5.1f} us from Python; "
          f"sync {cl['sync']:5.1f} us, async {cl['async']:5.1f} us from a C loop")
    p.close()
