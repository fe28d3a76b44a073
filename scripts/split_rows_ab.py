"""configs[2] (m = 4096, Blackman, FD float, n = 262144): analysis with half-row workgroups (option rows_split = 1, round 5) against the
two-slot kernel (option rows_split = 0), interleaved in one process; bits compared.  Also other row lengths and the relay alone.
    python scripts/split_rows_ab.py [rounds]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
print(f"device: {torch.cuda.get_device_name(0)}")
for m, window, n in ((4096, "blackman", 262144), (4096, "hann", 262144), (3072, "blackman", 262144), (2304, "hann", 262144)):
    x = torch.from_numpy(sine_sweep(n)).cuda()
    d = [torch.empty((n, m), dtype=torch.complex64, device="cuda") for _ in range(2)]
    plans = []
    for i, (label, opts) in enumerate((("half-row workgroups", {"rows_split": 1}), ("two-slot kernel", {"rows_split": 0}))):
        p = SDFT(m, window, 1.0, "f32f32")
        for k, v in opts.items():
            p.set_option(k, v)
        for _ in range(2):
            p.reset(); p.sdft(x, d[i])
        p.synchronize()
        plans.append((label, p, d[i]))
    same = bool(torch.equal(d[0], d[1]))
    res = {label: [] for label, _, _ in plans}
    for r in range(rounds):
        for label, p, out in plans:
            p.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                p.sdft(x, out)
            p.synchronize()
            res[label].append((time.perf_counter() - t0) / 5 * 1e3)
    b = n * (m * 8 + 4)
    for label, p, _ in plans:
        med = float(np.median(res[label]))
        print(f"m={m} {window:8s} n={n} {label:22s} median {med:7.3f} ms  min {min(res[label]):7.3f}  = {b / (med * 1e-3) / 8e12:5.1%} of peak"
              f"  split={p.get_option('last_rows_split')} chunks={p.get_option('last_chunks')}x{p.get_option('last_chunk_len')}  bits equal: {same}")
        p.close()
    del x, d
    torch.cuda.empty_cache()
