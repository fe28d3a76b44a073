"""Round trips at n = 1e6 (16.4 GB): the rows-in-step synthesis with non-temporal loads, the first S MB read (the matrix' end: what the analysis wrote last) with
ordinary loads all the same.  Synchronous calls, ms."""
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
y = torch.empty(n, dtype=torch.float32, device="cuda")
for rep in range(2):
    for label, opts in (("non-temporal", {"inverse_nt": 1}), ("ordinary", {"inverse_nt": 0}), ("nt, first 300 MB ordinary", {"inverse_nt": 1, "inverse_nt_skip_mb": 300}), ("nt, first 600 MB ordinary", {"inverse_nt": 1, "inverse_nt_skip_mb": 600}),
                        ("nt, first 1500 MB ordinary", {"inverse_nt": 1, "inverse_nt_skip_mb": 1500}), ("nt, first 4000 MB ordinary", {"inverse_nt": 1, "inverse_nt_skip_mb": 4000})):
        p = SDFT(m, "hann", 1.0, "f32f64")
        p.set_option("inverse_step", 1)
        for k, v in opts.items():
            p.set_option(k, v)
        iv, ro = [], []
        for r in range(14):
            p.sdft(x, A)
            t0 = time.perf_counter(); p.isdft(A, y); t1 = time.perf_counter()
            if r >= 6:
                iv.append(t1 - t0)
        for r in range(10):
            t0 = time.perf_counter(); p.isdft(A, y); t1 = time.perf_counter()
            if r >= 4:
                ro.append(t1 - t0)
        print(f"rows in step, {label:28s}: after the analysis {np.median(iv) * 1e3:6.3f} ms   only read {np.median(ro) * 1e3:6.3f} ms", flush=True)
        p.close()
