"""n = 1e6, m = 1024, f32f64 (16.4 GB): what slows the synthesis that follows an analysis -- the state of the matrix it reads (just written) or the state of the chip
(an analysis has just run)?  Synthesis of matrix B after an analysis into matrix A, against synthesis of A after the analysis into A, against synthesis alone.
Both plans' tuners settled first; synchronous calls, ms."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

m, n = 1024, 1000000
print(f"device: {torch.cuda.get_device_name(0)}")
x = torch.from_numpy(sine_sweep(n)).cuda()
A = torch.empty((n, m), dtype=torch.complex128, device="cuda")
B = torch.empty((n, m), dtype=torch.complex128, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")
p = SDFT(m, "hann", 1.0, "f32f64")
p.sdft(x, B); p.sdft(x, A)
for form, opts in (("default (tuners)", {}), ("rows in step forced", {"inverse_step": 1}), ("tree sum, ordinary loads", {"inverse_step": -1, "inverse_tune": 0, "inverse_nt": 0}), ("tree sum, non-temporal loads", {"inverse_step": -1, "inverse_tune": 0, "inverse_nt": 1})):
    q = SDFT(m, "hann", 1.0, "f32f64")
    for k, v in opts.items():
        q.set_option(k, v)
    res = {}
    for label, pattern in (("synthesis alone (B)", "s"), ("analysis into A, synthesis of B", "ab"), ("analysis into A, synthesis of A", "aa"), ("analysis into A, 2 ms pause, synthesis of A", "apa")):
        ts = []
        for r in range(24):
            if pattern != "s":
                p.sdft(x, A)
            if pattern == "apa":
                t = time.perf_counter()
                while time.perf_counter() - t < 2e-3:
                    pass
            src = B if pattern in ("s", "ab") else A
            t0 = time.perf_counter(); q.isdft(src, y); t1 = time.perf_counter()
            if r >= 16:
                ts.append(t1 - t0)
        res[label] = np.median(ts)
    print(f"{form:30s}: " + "   ".join(f"{k} {v * 1e3:6.3f} ms" for k, v in res.items()) + f"   (form {q.get_option('last_inverse_tuned')} nt {q.get_option('last_inverse_nt')})", flush=True)
    q.close()
