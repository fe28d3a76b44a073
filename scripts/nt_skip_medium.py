"""Round trips at medium sizes: non-temporal loads for the synthesis, the first S MB read (the matrix' end) with ordinary loads.  Synchronous calls, us."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT

print(f"device: {torch.cuda.get_device_name(0)}")
for combo, m, n in (("f64f64", 1000, 44100), ("f32f64", 1024, 48000), ("f32f64", 1024, 131072), ("f64f64", 1024, 262144), ("f32f64", 1024, 262144)):
    td = torch.float64 if combo[:3] == "f64" else torch.float32
    x = torch.randn(n, dtype=td, device="cuda")
    A = torch.empty((n, m), dtype=torch.complex128, device="cuda")
    y = torch.empty(n, dtype=td, device="cuda")
    row = []
    for skip in (0, 64, 128, 256, 512, 1024, 100000):
        p = SDFT(m, "hann", 1.0, combo)
        p.set_option("inverse_nt", 1); p.set_option("inverse_nt_skip_mb", skip)
        iv, ro = [], []
        for r in range(36):
            p.sdft(x, A)
            t0 = time.perf_counter(); p.isdft(A, y); t1 = time.perf_counter()
            if r >= 20:
                iv.append(t1 - t0)
        for r in range(30):
            t0 = time.perf_counter(); p.isdft(A, y); t1 = time.perf_counter()
            if r >= 18:
                ro.append(t1 - t0)
        row.append(f"{skip if skip < 100000 else 'all'}: {np.median(iv) * 1e6:.0f} / {np.median(ro) * 1e6:.0f}")
        p.close()
    print(f"{combo} m={m} n={n} ({n * m * 16 / 1e6:.0f} MB)  first S MB ordinary -> isdft after the analysis / only read, us:   " + "   ".join(row), flush=True)
