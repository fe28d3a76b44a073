"""Right after the analysis wrote a 706 MB matrix in 221 chunks of 200 rows (one round of the chip: every chunk advances at the same time), which rows does the Infinity
Cache still hold?  The synthesis of row subsets (sdft_isdft_nd, device row table): the last 72 rows of every chunk (254 MB: what was written last) against the first 72 rows
of every chunk, right after an analysis call and without one.  m = 1000, n = 44100, TD = FD = double."""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT

m, n = 1000, 44100
print(f"device: {torch.cuda.get_device_name(0)}")
x = torch.randn(n, dtype=torch.float64, device="cuda")
d = torch.empty((n, m), dtype=torch.complex128, device="cuda")
pa = SDFT(m, "hann", 1.0, "f64f64")
pa.sdft(x, d)
L = pa.get_option("last_chunk_len")
print(f"analysis: {pa.get_option('last_chunks')} chunks of {L} rows")
t = np.arange(n)
base = d.data_ptr()
sets = {"last 72 rows of every chunk": t[(t % L) >= L - 72], "first 72 rows of every chunk": t[(t % L) < 72], "middle rows of every chunk": t[((t % L) >= 64) & ((t % L) < 136)]}
for nt in (0, 1):
    ps = SDFT(m, "hann", 1.0, "f64f64"); ps.set_option("inverse_nt", nt); ps.set_option("inverse_tune", 0)
    for label, rows in sets.items():
        table = torch.from_numpy((base + rows.astype(np.int64) * m * 16)).cuda()
        y = torch.empty(len(rows), dtype=torch.float64, device="cuda")
        res = []
        for after in (True, False):
            ts = []
            for r in range(12):
                if after:
                    pa.sdft(x, d)
                torch.cuda.synchronize()
                t0 = time.perf_counter(); ps.api.isdft_nd(ps._p, len(rows), C.c_void_p(table.data_ptr()), C.c_void_p(y.data_ptr())); ps.synchronize(); t1 = time.perf_counter()
                if r >= 4:
                    ts.append(t1 - t0)
            res.append(np.median(ts))
        b = len(rows) * m * 16
        print(f"inverse_nt={nt} {label:30s} ({b / 1e6:.0f} MB): right after the analysis {res[0] * 1e6:6.1f} us ({b / res[0] / 1e9:5.0f} GB/s)   nothing written in between {res[1] * 1e6:6.1f} us ({b / res[1] / 1e9:5.0f} GB/s)", flush=True)
    ps.close()
